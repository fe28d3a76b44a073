# Round-4 measurements on one MI355X: bench line, rocprofv3 kernel stats of the headline workload alone, of the bench line's
# `configs` block (configs[2], configs[3], single-sample calls) and of every BASELINE shape, a kernel TRACE of config-3 calls
# (relay beside the forward launch), HBM traffic counters (separate --pmc passes, single and batch workload), probes.
# Outputs -> gpurun_out/r04/ (copied into profiles/ by scripts/r04_collect.py).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python3 $R/bench.py --steps 20 2>/dev/null | tail -1 > $O/bench_n1e6_m1024.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 20 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg -o cfg -- python3 $R/scripts/config_report.py > $O/configs_under_rocprof.md 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c3 -o c3 -- python3 $R/scripts/config3_calls.py sync > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wb -o wb -- python3 $R/bench.py --workload batch --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fb -o fb -- python3 $R/bench.py --workload batch --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
python3 $R/scripts/config_report.py > $O/configs.md 2>/dev/null
python3 $R/scripts/f32_rows_ab.py > $O/f32_rows_ab.txt 2>/dev/null
python3 $R/scripts/held_cus_ab.py > $O/held_cus.txt 2>/dev/null
python3 $R/scripts/relay_alone.py > $O/relay_alone.txt 2>/dev/null
python3 $R/scripts/roundtrip_pattern.py > $O/roundtrip_pattern.txt 2>/dev/null
python3 $R/scripts/store_ceiling_rows.py > $O/store_ceiling_rows.txt 2>/dev/null
python3 $R/scripts/ns_ab.py 2>/dev/null > $O/north_star_ab.txt
ls -la $O
