# counter passes of the relay kernel in two geometries (scripts/relay_variant.py), one --pmc set per pass; outputs under gpurun_out/r06n/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06n
mkdir -p $O
cd $R
for v in default groups2; do
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/${v}_a -o a -- python3 $R/scripts/relay_variant.py $v > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/${v}_b -o b -- python3 $R/scripts/relay_variant.py $v > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json, collections
for v in ("default", "groups2"):
    for s in ("a", "b"):
        f = glob.glob("$O/%s_%s/**/*counter_collection.csv" % (v, s), recursive=True)
        if not f:
            print(v, s, "no counters"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "carry_relay_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(v, s, json.dumps({k: int(sum(x) / len(x)) for k, x in sorted(acc.items())}), "launches", max((len(x) for x in acc.values()), default=0))
        t = glob.glob("$O/%s_%s/**/*kernel_trace.csv" % (v, s), recursive=True)
        if t:
            for name in ("carry_relay_kernel", "forward_rows_f32_kernel"):
                d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t[0])) if name in r["Kernel_Name"]]
                print(v, s, name, "durations us:", [round(x, 1) for x in d])
PY
