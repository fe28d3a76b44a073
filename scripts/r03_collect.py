"""Copies what scripts/r03_profile.sh left under gpurun_out/r03/ into profiles/ (the tracked copies the documents cite) and
rebuilds profiles/hbm_traffic.json (one entry per benchmark workload) from the counter passes.  Touches no document."""
import csv, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r03")
P = os.path.join(R, "profiles")
for src, dst in (("bench_n1e6_m1024.json", "r03_bench_n1e6_m1024.json"), ("bench_under_rocprof.json", "r03_bench_n1e6_m1024_under_rocprof.json"),
                 ("trace/bench_kernel_stats.csv", "r03_bench_n1e6_m1024_kernel_stats.csv"),
                 ("trace_cfg/cfg_kernel_stats.csv", "r03_all_configs_kernel_stats.csv"),
                 ("configs.md", "r03_configs.md"), ("configs_under_rocprof.md", "r03_configs_under_rocprof.md"),
                 ("process_perf.txt", "r03_fused_process_perf.txt"), ("self_perf.txt", "r03_self_carried_chunks_perf.txt"),
                 ("relay_perf.txt", "r03_exact_carry_relay_perf.txt"), ("relay_stats.txt", "r03_relay_wave_cycles.txt"),
                 ("relay_stamps.txt", "r03_relay_critical_path_stamps.txt"), ("relay_probe.txt", "r03_relay_token_probe.txt"),
                 ("add_latency_probe.txt", "r03_dependent_add_latency.txt"), ("nonlinear_perf.txt", "r03_nonlinear_ops_perf.txt"),
                 ("float_parallel_carries.txt", "r03_float_parallel_carries.txt"),
                 ("north_star_ab.txt", "r03_north_star_n48000_ab.txt"), ("hop_host.txt", "r03_hop_host_pointers.txt")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
cols = ("Counter_Name", "Counter_Value", "Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Start_Timestamp", "End_Timestamp")


def counters(path, kernel, names):
    return [x for x in csv.DictReader(open(os.path.join(O, path))) if kernel in x["Kernel_Name"] and x["Counter_Name"] in names]


entries, keep = [], []
for label, wf, ff, n, channels in (("single", "pmc_w/w_counter_collection.csv", "pmc_f/f_counter_collection.csv", 1000000, 1),
                                   ("batch", "pmc_wb/wb_counter_collection.csv", "pmc_fb/fb_counter_collection.csv", 48000, 64)):
    w = counters(wf, "forward_rows_kernel", ("WRITE_SIZE",)); f = counters(ff, "forward_rows_kernel", ("FETCH_SIZE",))
    keep += w + f
    W = sum(float(x["Counter_Value"]) for x in w) / max(len(w), 1); F = sum(float(x["Counter_Value"]) for x in f) / max(len(f), 1)
    entries.append({"workload": label, "n": n, "m": 1024, "channels": channels, "bytes_per_launch": int(W * 1024 + 2 * F * 1024),
                    "WRITE_SIZE_KiB": W, "FETCH_SIZE_KiB_raw": F, "launches": len(w),
                    "algorithmic_bytes_per_launch": channels * n * (1024 * 16 + 4),
                    "note": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes over bench.py --no-extras (round 3, forward_rows_kernel "
                            "dispatches only, average per launch); counters are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide reads)",
                    "source": "profiles/r03_bench_pmc.csv"})
with open(os.path.join(P, "r03_bench_pmc.csv"), "w", newline="") as fh:
    wr = csv.writer(fh); wr.writerow(cols); wr.writerows([[x[k] for k in cols] for x in keep])
json.dump(entries, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
for e in entries:
    print(e["workload"], "launches", e["launches"], "traffic", e["bytes_per_launch"], "algorithmic", e["algorithmic_bytes_per_launch"], "ratio %.4f" % (e["bytes_per_launch"] / e["algorithmic_bytes_per_launch"]))
rows = []
for f in ("pmc_p1/p1_counter_collection.csv", "pmc_p2/p2_counter_collection.csv"):
    rows += [x for x in csv.DictReader(open(os.path.join(O, f))) if "process_rows_kernel" in x["Kernel_Name"]]
agg = {}
for x in rows:
    agg.setdefault(x["Counter_Name"], []).append(float(x["Counter_Value"]))
with open(os.path.join(P, "r03_fused_kernel_counters.json"), "w") as fh:
    json.dump({k: {"per_launch": sum(v) / len(v), "launches": len(v)} for k, v in agg.items()}, fh, indent=1)
print({k: round(sum(v) / len(v)) for k, v in agg.items()})
b = json.loads(open(os.path.join(P, "r03_bench_n1e6_m1024.json")).read())
print("value", b["value"], "frac", b["roofline"]["frac"], "north star", b["north_star_n48000"]["sync"]["frac_of_peak_wall"], b["north_star_n48000"]["async"]["frac_of_peak_wall"])
