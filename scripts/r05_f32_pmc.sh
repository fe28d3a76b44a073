#!/bin/bash
# counter passes of the FD float analysis kernels alone (chunk-parallel carries: no relay beside them), m = 4096 Blackman:
# two-slot kernel and half-row workgroups, whole chip and a quarter of it (192 CUs held)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_f32pmc
mkdir -p $O
for split in 0 1; do
  for held in 0 192; do
    SDFT_SPLIT=$split rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES \
      --kernel-trace --output-format csv -d $O/s${split}_h${held}_a -o c -- python3 $R/scripts/f32_kernel_pmc_run.py $held > $O/s${split}_h${held}_a.log 2>&1
    SDFT_SPLIT=$split rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT \
      --kernel-trace --output-format csv -d $O/s${split}_h${held}_b -o c -- python3 $R/scripts/f32_kernel_pmc_run.py $held > $O/s${split}_h${held}_b.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/s*_h*_?")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "forward_rows_f32" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
    print(d.split("/")[-1], {k: round(v[0] / max(v[1], 1)) for k, v in sorted(acc.items())})
PY
