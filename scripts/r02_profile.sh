# Round-2 measurements on one MI355X: bench line, rocprofv3 kernel stats of the headline workload alone,
# HBM traffic counters (separate --pmc passes), all BASELINE shapes, probes.  Outputs -> gpurun_out/r02/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
python3 $R/bench.py --steps 20 2>/dev/null | tail -1 > $O/bench_n1e6_m1024.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 20 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_all -o bench_all -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-cpu-all-cores > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
python3 $R/scripts/config_report.py > $O/configs.md 2>/dev/null
python3 $R/scripts/chain_stats.py 2>/dev/null > $O/chain_stats.txt
$R/scripts/bin/dep > $O/latency_probe.txt 2>&1
$R/scripts/bin/ll >> $O/latency_probe.txt 2>&1
python3 $R/scripts/quick_perf.py process 2>/dev/null > $O/process_perf.txt
