cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TA_[A-Z_0-9]+|GRBM_[A-Z_0-9]+)" | sort -u > gpurun_out/counters.txt; wc -l gpurun_out/counters.txt
mkdir -p gpurun_out/prof2
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof2/s$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof2/s$i.log 2>&1
  tail -1 gpurun_out/prof2/s$i.log | cut -c1-150
done
python3 - <<'PY'
import csv,glob
for f in sorted(glob.glob("gpurun_out/prof2/s*/*/*_counter_collection.csv")):
    agg={}
    for r in csv.DictReader(open(f)):
        if "forward" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v))
PY
