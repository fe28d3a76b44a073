"""Copies what scripts/r06_profile.sh left under gpurun_out/r06/ into profiles/ (the tracked copies the documents cite), rebuilds
profiles/hbm_traffic.json (one entry per benchmark workload) from the counter passes and prints the figures the documents quote.  Touches no document."""
import csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r06")
P = os.path.join(R, "profiles")
for src, dst in (("bench_n1e6_m1024.json", "r06_bench_n1e6_m1024.json"), ("bench_under_rocprof.json", "r06_bench_n1e6_m1024_under_rocprof.json"),
                 ("trace/bench_kernel_stats.csv", "r06_bench_n1e6_m1024_kernel_stats.csv"),
                 ("trace_full/full_kernel_stats.csv", "r06_bench_all_side_measurements_kernel_stats.csv"),
                 ("trace_cfg/cfg_kernel_stats.csv", "r06_all_configs_kernel_stats.csv"),
                 ("configs.md", "r06_configs.md"), ("configs_under_rocprof.md", "r06_configs_under_rocprof.md"),
                 ("north_star_ab.txt", "r06_north_star_n48000_ab.txt"), ("hop_host.txt", "r06_hop_loop.txt"), ("pipelined_calls.txt", "r06_pipelined_calls.txt")):
    hits = glob.glob(os.path.join(O, src)) or glob.glob(os.path.join(O, os.path.dirname(src), "**", os.path.basename(src)), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(P, dst))
    else:
        print("missing:", src)
# a header for the pipelined calls' A/B
pc = os.path.join(P, "r06_pipelined_calls.txt")
if os.path.exists(pc):
    text = open(pc).read()
    if not text.startswith("Asynchronous"):
        open(pc, "w").write("Asynchronous sdft_sdft_n calls through the raw C-ABI into TWO matrices in turn, both placed by sdft_hip_malloc_matrix_in_arena (equal footing), option pipeline\n"
                            "= 0 (one stream) / 1 (default: calls below 2^29 bins) / 2 (any length), interleaved in one process, four rounds, 200 calls each (20 at n = 1e6) after 6\n"
                            "warm-up calls (scripts/pipeline_ab.py, the round's session); microseconds per call = fraction of the 8 TB/s HBM peak by the algorithmic bytes.\n"
                            "Pipelining hides a fixed 10-40 us per call (launch gap, prologue of the self-carried chunks, ragged end): +12 % at n = 24 000, +9 % at 48 000, +7 % at 90 000,\n"
                            "+12 % at 131 072, a tie at 1e6 -- where the default (1) therefore stays on one stream.\n" + text)
# the share of the placement probes in the GPU time of a whole bench run (all side measurements)
full = os.path.join(P, "r06_bench_all_side_measurements_kernel_stats.csv")
if os.path.exists(full):
    rows = list(csv.DictReader(open(full)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    probes = sum(float(r["TotalDurationNs"]) for r in rows if "store_rowgroup_kernel" in r["Name"] or "store_parts_kernel" in r["Name"] or "store_linear_kernel" in r["Name"])
    placing = sum(float(r["TotalDurationNs"]) for r in rows if "store_parts_kernel" in r["Name"])
    print("whole bench under rocprof: GPU time %.1f ms, store-only kernels (placement probes AND the store-only ceilings of the roofline block) %.1f ms = %.1f %%; the two-part placement probes alone %.2f ms"
          % (total / 1e6, probes / 1e6, 100.0 * probes / max(total, 1), placing / 1e6))
# kernel trace of config-2 calls: the relay beside the forward launch (start / end / duration in microseconds from the first row)
tr = glob.glob(os.path.join(O, "trace_c3", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))[-18:]
    t0 = int(rows[0]["Start_Timestamp"])
    with open(os.path.join(P, "r06_config2_kernel_trace.txt"), "w") as fh:
        fh.write("# rocprofv3 --kernel-trace of scripts/config3_calls.py (BASELINE configs[2]: m = 4096, Blackman, FD float, n = 262144; synchronous calls), per launch: kernel, start us, end us, duration us\n")
        for r in rows:
            fh.write("%-72s %10.1f %10.1f %9.1f\n" % (r["Kernel_Name"][:72], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
# the forward kernel's launches of `bench.py --no-extras` under rocprof, one by one: bench_kernel_stats.csv averages the launches into the placed matrix
# with those into the first allocation (and the warm-ups); this file keeps them apart, in the order bench.py makes them
tr = glob.glob(os.path.join(O, "trace", "**", "*kernel_trace.csv"), recursive=True)
und = os.path.join(P, "r06_bench_n1e6_m1024_under_rocprof.json")
if tr and os.path.exists(und):
    line = json.loads(open(und).read().strip().splitlines()[-1])
    W, K = line["warmup"], line["steps"]
    rows = sorted((r for r in csv.DictReader(open(tr[0])) if "forward_rows_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    regions = [("warm-up, placed matrix", W), ("TIMED REGION, placed matrix (the headline: roofline.avg_launch_ms)", K), ("warm-up, first allocation", W),
               ("timed region, first allocation (first_allocation)", K), ("one at a time, placed matrix", min(K, 20)), ("the rest (analysis + synthesis pairs, placed matrix)", len(d))]
    with open(os.path.join(P, "r06_bench_forward_launches.txt"), "w") as fh:
        fh.write("# rocprofv3 --kernel-trace of `bench.py --steps %d --no-extras --no-cpu-baseline` (the command of r06_bench_n1e6_m1024_kernel_stats.csv): every launch of\n"
                 "# forward_rows_kernel<double, 1, 1, true, 1, 0, true, float, false>, milliseconds, in launch order, by the region of bench.py it belongs to.\n"
                 "# The line this run printed: roofline.avg_launch_ms %.4f (HIP events), first_allocation.frac %.4f.\n" % (K, line["roofline"]["avg_launch_ms"], line["first_allocation"]["frac"]))
        i = 0
        for name, count in regions:
            part = d[i:i + count]; i += len(part)
            if part:
                fh.write("%-75s launches %3d  mean %.4f ms  min %.4f  max %.4f  = %.4f of 8 TB/s\n" % (name, len(part), sum(part) / len(part), min(part), max(part),
                                                                                                   line["roofline"]["algorithmic_bytes_per_launch"] / (sum(part) / len(part) * 1e-3) / 8e12))
                fh.write("    " + " ".join("%.3f" % v for v in part) + "\n")
cols = ("Counter_Name", "Counter_Value", "Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Start_Timestamp", "End_Timestamp")


def counters(path, kernel, names):
    hits = glob.glob(os.path.join(O, os.path.dirname(path), "**", os.path.basename(path)), recursive=True)
    if not hits:
        return []
    return [x for x in csv.DictReader(open(hits[0])) if kernel in x["Kernel_Name"] and x["Counter_Name"] in names]


entries, keep = [], []
for label, wf, ff, n, channels in (("single", "pmc_w/w_counter_collection.csv", "pmc_f/f_counter_collection.csv", 1000000, 1),
                                   ("batch", "pmc_wb/wb_counter_collection.csv", "pmc_fb/fb_counter_collection.csv", 48000, 64)):
    w = counters(wf, "forward_rows_kernel", ("WRITE_SIZE",)); f = counters(ff, "forward_rows_kernel", ("FETCH_SIZE",))
    if not w or not f:
        print("no counters for", label); continue
    keep += w + f
    W = sum(float(x["Counter_Value"]) for x in w) / max(len(w), 1); F = sum(float(x["Counter_Value"]) for x in f) / max(len(f), 1)
    entries.append({"workload": label, "n": n, "m": 1024, "channels": channels, "bytes_per_launch": int(W * 1024 + 2 * F * 1024),
                    "WRITE_SIZE_KiB": W, "FETCH_SIZE_KiB_raw": F, "launches": len(w),
                    "algorithmic_bytes_per_launch": channels * n * (1024 * 16 + 4),
                    "note": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes over bench.py --no-extras (round 6, forward_rows_kernel "
                            "dispatches only -- into the placed matrix and into the first allocation alike --, average per launch); counters are in KiB; "
                            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide reads)",
                    "source": "profiles/r06_bench_pmc.csv"})
if entries:
    with open(os.path.join(P, "r06_bench_pmc.csv"), "w", newline="") as fh:
        wr = csv.writer(fh); wr.writerow(cols); wr.writerows([[x[k] for k in cols] for x in keep])
    json.dump(entries, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
for e in entries:
    print(e["workload"], "launches", e["launches"], "traffic", e["bytes_per_launch"], "algorithmic", e["algorithmic_bytes_per_launch"], "ratio %.4f" % (e["bytes_per_launch"] / e["algorithmic_bytes_per_launch"]))
shutil.copy(os.path.join(P, "r06_bench_n1e6_m1024.json"), os.path.join(P, "r06.json"))        # the short name the documents' tables cite
b = json.loads(open(os.path.join(P, "r06_bench_n1e6_m1024.json")).read())
print("value", b["value"], "frac", b["roofline"]["frac"], "first allocation", b["first_allocation"]["value"], b["first_allocation"]["frac"],
      "north star", b["north_star_n48000"]["sync"]["frac_of_peak_wall"], b["north_star_n48000"]["async"]["frac_of_peak_wall"])
