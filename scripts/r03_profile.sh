# Round-3 measurements on one MI355X: bench line, rocprofv3 kernel stats of the headline workload alone and of every
# BASELINE shape, HBM traffic counters (separate --pmc passes, single and batch workload), fused-kernel counters, probes.
# Outputs -> gpurun_out/r03/ (copied into profiles/ by scripts/r03_collect.py).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
python3 $R/bench.py --steps 20 2>/dev/null | tail -1 > $O/bench_n1e6_m1024.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 20 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg -o cfg -- python3 $R/scripts/config_report.py > $O/configs_under_rocprof.md 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wb -o wb -- python3 $R/bench.py --workload batch --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fb -o fb -- python3 $R/bench.py --workload batch --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_p1 -o p1 -- python3 $R/scripts/quick_perf.py process1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_p2 -o p2 -- python3 $R/scripts/quick_perf.py process1 > /dev/null 2>&1
python3 $R/scripts/config_report.py > $O/configs.md 2>/dev/null
python3 $R/scripts/relay_stats.py 2>/dev/null > $O/relay_stats.txt
python3 $R/scripts/relay_stamps.py 2>/dev/null > $O/relay_stamps.txt
$R/scripts/bin/relay_probe > $O/relay_probe.txt 2>&1
$R/scripts/bin/add_latency_probe > $O/add_latency_probe.txt 2>&1
python3 $R/scripts/nonlinear_perf.py 2>/dev/null > $O/nonlinear_perf.txt
python3 $R/scripts/float_fast_probe.py 2>/dev/null > $O/float_parallel_carries.txt
python3 $R/scripts/quick_perf.py process 2>/dev/null > $O/process_perf.txt
python3 $R/scripts/quick_perf.py self 2>/dev/null > $O/self_perf.txt
python3 $R/scripts/quick_perf.py relay 2>/dev/null > $O/relay_perf.txt
python3 $R/scripts/ns_ab.py 2>/dev/null > $O/north_star_ab.txt
python3 $R/scripts/hop_host.py 2>/dev/null > $O/hop_host.txt
ls -la $O
