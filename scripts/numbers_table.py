"""Prints the table of measured numbers the documents carry (README.md) from the round's session files, every number as a checked claim
<value · file> (tests/test_docs.py).  python scripts/numbers_table.py > /tmp/table.md"""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = "profiles/r06.json"          # (a copy of profiles/r06_bench_n1e6_m1024.json under a short name: the tables cite it forty times)
b = json.loads(open(os.path.join(R, F)).read().strip().splitlines()[-1])


def g(path):
    v = b
    for k in path.split("."):
        v = v[k]
    return v


def c(path, unit="", fmt=None):
    v = g(path)
    s = (fmt % v) if fmt else str(v)
    return "⟨%s%s · %s⟩" % (s, (" " + unit) if unit else "", F)


rows = [
    ("**configs[1]: n = 1e6, m = 1024, Hann, FD double, analysis** (`value`) — matrix placed by the library inside matrix + 64 GiB",
     "**%s Msamples/s, %s ms per step, %s of the 8 TB/s peak** (kernel %s ms; %s of the best store-only kernel on the same buffer; HBM traffic / algorithmic bytes 1.001)"
     % (c("value"), c("ms_per_step"), c("roofline.frac"), c("roofline.avg_launch_ms"), c("roofline.store_only_ceiling.frac_of_best_store_only"))),
    ("… the same K steps into the process' first plain allocation (`first_allocation`)", "%s Msamples/s, %s of peak (by lease: 0.71 … 0.87 — whether the allocation happens to straddle a change of kind)"
     % (c("first_allocation.value"), c("first_allocation.frac"))),
    ("… placement: arena, probes", "%s bytes; %s two-part probes + %s full-size probes, %s ms of GPU time; the window %s GB/s store-only against %s at the allocation's start"
     % (c("buffer_placement.arena_bytes"), c("buffer_placement.pair_probes"), c("buffer_placement.window_probes"), c("buffer_placement.probe_ms"), c("buffer_placement.window_gbs"), c("buffer_placement.start_gbs"))),
    ("analysis + synthesis pairs; synthesis alone", "%s Msamples/s per pair; synthesis %s Msamples/s after the write, %s on a matrix that is only read"
     % (c("analysis_plus_synthesis_msamples_s"), c("synthesis_msamples_s"), c("synthesis_matrix_only_read_msamples_s"))),
    ("**the north star's shape, n = 48 000** (786 MB), placed matrix, default options", "synchronous %s of peak (%s ms per call), asynchronous %s, kernel %s; into a plain allocation synchronous %s"
     % (c("north_star_n48000.sync.frac_of_peak_wall"), c("north_star_n48000.sync.ms_per_call_wall"), c("north_star_n48000.async.frac_of_peak_wall"), c("north_star_n48000.forward_kernel_frac_of_peak"), c("north_star_n48000.sync_first_allocation.frac_of_peak_wall"))),
    ("… asynchronous calls into two matrices in turn", "pipelined (default) %s, one stream %s" % (c("north_star_n48000.async_two_buffers.frac_of_peak_wall"), c("north_star_n48000.async_two_buffers_one_stream.frac_of_peak_wall"))),
    ("n = 1e6 into two matrices in turn", "pipeline = 2 %s, one stream %s, the default (one stream at this length) %s; synthesis of the two in turn %s"
     % (c("two_matrices_in_turn.pipelined.frac_of_peak_wall"), c("two_matrices_in_turn.one_stream.frac_of_peak_wall"), c("two_matrices_in_turn.library_default.frac_of_peak_wall"), c("two_matrices_in_turn.synthesis_frac_of_peak_wall"))),
    ("one GPU's share of configs[4]: 64 channels × 48 000 × 1024", "analysis %s Msamples/s = %s of peak (first allocation %s), synthesis %s Msamples/s"
     % (c("batch_share.analysis_msamples_s"), c("batch_share.analysis_frac_of_peak"), c("batch_share.first_allocation.analysis_frac_of_peak"), c("batch_share.synthesis_msamples_s"))),
    ("configs[2]: m = 4096, Blackman, FD float, n = 262 144, bit-identical", "analysis %s ms = %s of peak, synthesis %s, round trip %s"
     % (c("configs.config2.forward_ms_wall"), c("configs.config2.forward_frac_of_peak"), c("configs.config2.inverse_frac_of_peak"), c("configs.config2.round_trip_frac_of_peak"))),
    ("configs[3]: 64 channels × 48 000 × 2048 (100.7 GB)", "analysis %s of peak, synthesis %s" % (c("configs.config3.forward_frac_of_peak"), c("configs.config3.inverse_frac_of_peak"))),
    ("the reference's hop loop (m = 1000, hop = 100), device pointers", "two synchronous calls %s µs per hop (%s from a C loop), with `resident` = 1 %s µs (%s), asynchronous %s µs; the fused call %s / %s µs; the reference on one host core %s µs"
     % (c("hop100_m1000.us_per_hop_sync"), c("hop100_m1000.us_per_hop_sync_c_loop"), c("hop100_m1000.us_per_hop_resident_sync"), c("hop100_m1000.us_per_hop_resident_sync_c_loop"), c("hop100_m1000.us_per_hop_async"), c("hop100_m1000.us_per_hop_process_n_sync"), c("hop100_m1000.us_per_hop_process_n_async"), c("hop100_m1000.cpu_reference.us_per_hop"))),
    ("… on the reference driver's malloc'ed buffers", "%s µs per hop by default (pinned pieces of the plan), %s with the runtime's copy, %s registered in place; PCIe floor %s"
     % (c("hop100_m1000.us_per_hop_host_pointers"), c("hop100_m1000.us_per_hop_host_pointers_runtime_copy"), c("hop100_m1000.us_per_hop_host_pointers_registered"), c("hop100_m1000.host_pointers_pcie_floor_us"))),
    ("the reference's bench shape (m = 1000, 44 100 samples, TD = FD = double)", "`sdft` %s µs, `isdft` %s µs (the reference on one core: %s / %s µs)"
     % (c("reference_bench_shape.gpu_sdft_us"), c("reference_bench_shape.gpu_isdft_us"), c("reference_bench_shape.cpu_sdft_us"), c("reference_bench_shape.cpu_isdft_us"))),
    ("single samples (`sdft_sdft` / `sdft_isdft`, device row)", "%s / %s µs per call, with `resident` = 1 %s µs per sample for the pair (one host core of the reference: %s / %s)"
     % (c("configs.single_sample.sdft_us_per_call_device_row"), c("configs.single_sample.isdft_us_per_call_device_row"), c("configs.single_sample.sdft_plus_isdft_us_per_sample_resident"), c("configs.single_sample.cpu_sdft_us_per_sample"), c("configs.single_sample.cpu_isdft_us_per_sample"))),
    ("fused analysis → operation → synthesis (`sdft_hip_process_n`), n = 1e6", "%s Msamples/s (tree sum), %s in the reference's order" % (c("fused_process.tree_sum_msamples_s"), c("fused_process.reference_order_msamples_s"))),
    ("host samples in, host matrix out (PCIe-inclusive, never `value`)", "%s Msamples/s" % c("extras.host_pointer_pcie_inclusive_msamples_s")),
    ("the reference's `sdft_sdft_n` on one host core of the GPU box (`cpu_baseline`, kind reference, 256-core host)", "%s Msamples/s" % c("cpu_baseline.value")),
]
print("| what | measured (one session, `scripts/r06_profile.sh`) |\n|---|---|")
for a, v in rows:
    print("| %s | %s |" % (a, v))
