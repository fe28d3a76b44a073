# Round-6 measurements on one MI355X (one lease: the numbers of one session belong together): bench line, rocprofv3 kernel stats of the headline
# workload alone and of every BASELINE shape, a kernel TRACE of config-2 calls (relay beside the forward launch), HBM traffic counters (separate
# --pmc passes, single and batch workload), the north star's A/B with and without placement, the hop loop.
# Outputs -> gpurun_out/r06/ (copied into profiles/ by scripts/r06_collect.py).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python3 $R/bench.py --steps 20 2>/dev/null | tail -1 > $O/bench_n1e6_m1024.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 20 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o full -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-cpu-all-cores > $O/bench_full_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg -o cfg -- python3 $R/scripts/config_report.py > $O/configs_under_rocprof.md 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c3 -o c3 -- python3 $R/scripts/config3_calls.py sync > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wb -o wb -- python3 $R/bench.py --workload batch --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fb -o fb -- python3 $R/bench.py --workload batch --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
python3 $R/scripts/config_report.py > $O/configs.md 2>/dev/null
python3 $R/scripts/ns_ab.py 2>/dev/null > $O/north_star_ab.txt
python3 $R/scripts/hop_host.py 2>/dev/null > $O/hop_host.txt
python3 $R/scripts/pipeline_ab.py 2>/dev/null > $O/pipelined_calls.txt
ls -la $O
