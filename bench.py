#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Sliding DFT engine (contract: see README/DESIGN).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: starts the N ranks itself, as a child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM: `sdft_sdft_n` through the C-ABI of libsdft_hip.so on device pointers.

  N = 1  -> BASELINE.json configs[1]: n = 1e6 samples, m = 1024, Hann, TD float / FD double,
            forward only, one channel.
  N > 1  -> BASELINE.json configs[4]: 64 independent channels per GPU (512 at N = 8), n = 48000
            each, m = 1024, Hann, FD double; channels are sharded over ranks, no data-path
            collective (RCCL is used for the barrier and the max/sum of scalars only).

Rank 0 prints ONE JSON line.

* `value` = samples ANALYSED (sdft_sdft_n) by all ranks / max-over-ranks wall time of the K timed
  steps -- what configs[1] ("forward sdft only") asks for.  The metric string also names
  synthesis: a second bracketed region times K analysis+synthesis pairs on every rank and reports
  `analysis_plus_synthesis_msamples_s` (whole job, max over ranks) next to `synthesis_msamples_s`.
* `roofline` prices the dominant kernel by its algorithmic bytes (m*sizeof(fdx) + sizeof(td) per
  sample) over the kernel's own duration, measured with HIP events recorded by the library on the
  stream the kernel runs on.
* `cpu_baseline` is the oracle (the genuine reference build when oracle/_ref is present, else our
  bit-identical port) timed on one host core on a bounded sample of the same workload.
* At N = 1 rank 0 also reports, outside the timed regions: the north star's own shape (n = 48000)
  on the DEFAULT drop-in path (no pointer hints; synchronous and asynchronous calls, into one matrix and into two in
  turn), the headline workload as asynchronous calls into two matrices in turn (`two_matrices_in_turn`: pipelined calls), the
  reference's streaming test shape (hop = 100, m = 1000), one GPU's share of configs[4]
  (64 channels x 48000, so the 1 -> 8 GPU curve has a like-for-like N = 1 point) with the
  many-core CPU baseline beside it, the fused analysis->synthesis path, and the PCIe-inclusive
  host-pointer rate.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "single", "batch"])
    ap.add_argument("--n", "--samples", dest="n", type=int, default=0, help="samples per channel (0 = workload default)")
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--channels-per-gpu", type=int, default=64)
    ap.add_argument("--window", default="hann")
    ap.add_argument("--combo", default="f32f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=200000)
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements outside the timed regions")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the many-core CPU baseline of the batch share")
    ap.add_argument("--no-placement", action="store_true", help="take the output matrix as the first allocation comes (the line's first_allocation figure becomes the headline)")
    return ap.parse_args()


def cpu_baseline(m, window, combo, n_cpu):
    """Oracle forward pass on one host core, output pre-touched, best of 5."""
    import numpy as np
    from oracle import oracle as O
    from sdft_amd.signals import sine_sweep
    if not O.have_port(combo):
        O.build(ref=True)
    td, fd, fdx = O.combo_types(combo)
    try:
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
        n_cpu = max(4096, min(n_cpu, int(avail * 0.25) // (m * np.dtype(fdx).itemsize)))
    except Exception:
        pass
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
        pinned = True
    except Exception:
        pinned = False
    x = sine_sweep(n_cpu, dtype=td)
    plan = O.best(m, window, 1.0, combo)
    out = np.zeros((n_cpu, m), dtype=fdx)           # pre-touch: first-touch faults would dominate
    best = float("inf")
    budget = time.perf_counter() + 25.0
    for _ in range(5):
        plan.reset()
        t0 = time.perf_counter()
        plan.sdft(x, out)
        best = min(best, time.perf_counter() - t0)
        if time.perf_counter() > budget:
            break
    try:
        os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
    except Exception:
        pass
    res = {
        "value": round(n_cpu / best / 1e6, 4),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": plan.kind,
        "sample": f"sdft_sdft_n forward, n={n_cpu} of the sine sweep, m={m}, {window}, {combo}, 1 thread"
                  f"{' pinned' if pinned else ''}, output pre-touched, best of 5; host has {os.cpu_count()} cores",
        "flags": "gcc -std=gnu99 -O2 -ffp-contract=off (the canonical oracle build, SURVEY.md 8c)",
    }
    # second figure: the same code with -O3 -march=native, compiled on this machine: the genuine header
    # when SDFT_REF_DIR points at a checkout of the reference, else the bit-identical restatement (the GPU box has none)
    if combo == "f32f64":
        try:
            path, kind = O.build_native(combo)
            nat = (O.Reference if kind == "reference" else O.Port)(m, window, 1.0, combo, lib_path=path)
            try:
                os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
            except Exception:
                pass
            nbest = float("inf")
            for _ in range(3):
                nat.reset()
                t0 = time.perf_counter()
                nat.sdft(x, out)
                nbest = min(nbest, time.perf_counter() - t0)
            try:
                os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
            except Exception:
                pass
            res["o3_march_native"] = {"value": round(n_cpu / nbest / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": kind,
                                      "flags": "gcc -std=gnu99 -O3 -march=native -ffp-contract=off, built on this host",
                                      "note": None if kind == "reference" else "reference sources are not on this machine (SDFT_REF_DIR unset): the restatement stands in"}
        except Exception as e:                                   # no compiler on the host: say so
            res["o3_march_native"] = {"value": None, "note": f"not built: {e}"[:200]}
    return res


def north_star_shape(torch, np, SDFT, sine_sweep, cdt, m, window, combo, esz, td, device, placed=True):
    """n = 48000 (the shape the north star quotes its >= 50 % target on) on the default drop-in path:
    device pointers, NO pointer hints, synchronous calls (what a C host that just calls sdft_sdft_n
    gets) and asynchronous ones; wall clock per call.  The matrices (786 MB) are placed like the headline's
    (sdft_hip_malloc_matrix_in_arena: the kind of memory matters at this size too); `sync_first_allocation` is the
    synchronous call into a plain allocation."""
    n48 = 48000
    x48 = torch.from_numpy(sine_sweep(n48, dtype=td)).cuda()
    o_plain = torch.empty((n48, m), dtype=cdt, device="cuda")
    o48, pl_a, hold_a = place_matrix(torch, (n48, m), cdt, placed=placed)
    o48b, pl_b, hold_b = place_matrix(torch, (n48, m), cdt, placed=placed)
    b48 = n48 * (m * esz + np.dtype(td).itemsize)
    res = {"buffer_placement": {"first": {k: pl_a.get(k) for k in ("placed", "window_gbs", "start_gbs", "boundary_offset", "pair_probes", "window_probes")},
                                "second": {k: pl_b.get(k) for k in ("placed", "window_gbs", "start_gbs", "boundary_offset")}}}
    # (async_two_buffers: a host that alternates between two matrices -- consecutive calls then overlap, option "pipeline";
    # calls into ONE matrix are ordered behind each other as on one stream)
    for mode in ("sync", "async", "async_two_buffers", "async_two_buffers_one_stream", "sync_first_allocation"):
        p = SDFT(m, window, 1.0, combo, device=device)
        if not mode.startswith("sync"):
            p.set_option("async", 1)
        if mode == "async_two_buffers_one_stream":
            p.set_option("pipeline", 0)
        xs48 = C.c_void_p(x48.data_ptr())
        first = o_plain if mode == "sync_first_allocation" else o48
        os48 = [C.c_void_p(first.data_ptr()), C.c_void_p((o48b if mode.startswith("async_two_buffers") else first).data_ptr())]
        for i in range(20):                                  # (the plan's first pipelined calls look for two concurrent streams: a millisecond, once)
            p.api.sdft_n(p._p, n48, xs48, os48[i & 1])       # the raw C-ABI call, as a C host makes it
        p.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            p.api.sdft_n(p._p, n48, xs48, os48[i & 1])
        p.synchronize(); torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / 200
        res[mode] = {"ms_per_call_wall": round(w * 1e3, 4), "msamples_s_wall": round(n48 / w / 1e6, 1),
                     "gbs_wall": round(b48 / w / 1e9, 1), "frac_of_peak_wall": round(b48 / w / 1e9 / HBM_PEAK_GBS, 4)}
        if mode.startswith("async_two_buffers"):
            res[mode]["pipelined_calls"] = int(p.get_option("pipelined_calls"))
            res[mode]["chunks"] = [int(p.get_option("last_chunks")), int(p.get_option("last_chunk_len"))]
            res[mode]["row_streams"] = {0: "none (one stream)", 1: "ordinary", 2: "by priority"}.get(int(p.get_option("pipeline_streams")) // 10, "?")
        if mode == "async":
            p.set_option("profile", 1)                   # kernel time in a separate pass
            for _ in range(20):
                p.sdft(x48, o48)
            pr = p.profile()
            k48 = pr["forward"][0] / max(pr["forward"][1], 1) * 1e-3
            res["forward_kernel_gbs"] = round(b48 / k48 / 1e9, 1)
            res["forward_kernel_frac_of_peak"] = round(b48 / k48 / 1e9 / HBM_PEAK_GBS, 4)
            res["prepass_us"] = round((pr["delta"][0] + pr["carry"][0]) / max(pr["forward"][1], 1) * 1e3, 1)
            res["launches_per_call"] = 1 if p.get_option("last_self") == 1 else 3
            res["self_carried_chunks"] = bool(p.get_option("last_self") == 1)
            res["chunks"] = [int(p.get_option("last_chunks")), int(p.get_option("last_chunk_len"))]
        p.close()
    del o48, o48b, o_plain
    for h in (hold_a, hold_b):
        if h is not None:
            h.free()
    res["path"] = "default: pointers classified by the library on every call, no options set"
    res["note"] = "786 MB matrix: part of the write is absorbed by the 256 MiB Infinity Cache"
    return res


def hop_streaming(torch, np, SDFT, sine_sweep, combo, td, cdt, device, m=1000, hop=100, total=20000):
    """/root/reference/test/test.c:69-83 with test/main.sh:3-6: hops of 100 samples, dftsize 1000,
    Hann: sdft_sdft_n + sdft_isdft_n per hop on device pointers through the raw C-ABI."""
    x = torch.from_numpy(sine_sweep(total, dtype=td)).cuda()
    y = torch.empty(total, dtype=x.dtype, device="cuda")
    d = torch.empty((hop, m), dtype=cdt, device="cuda")
    p = SDFT(m, "hann", 1.0, combo, device=device)
    xs, ys, ds, isz = x.data_ptr(), y.data_ptr(), d.data_ptr(), x.element_size()
    res = {"dftsize": m, "hop": hop, "calls": "sdft_sdft_n + sdft_isdft_n per hop, device pointers, default options"}
    for mode in ("sync", "async"):
        p.set_option("async", 1 if mode == "async" else 0)
        w = 0.0
        for rep in range(2):
            p.synchronize()
            t0 = time.perf_counter()
            for i in range(0, total, hop):
                p.api.sdft_n(p._p, hop, C.c_void_p(xs + i * isz), C.c_void_p(ds))
                p.api.isdft_n(p._p, hop, C.c_void_p(ds), C.c_void_p(ys + i * isz))
            p.synchronize()
            w = (time.perf_counter() - t0) / (total // hop)
        res[f"us_per_hop_{mode}"] = round(w * 1e6, 1)
    # option "resident" = 1 (off by default): the same two synchronous calls per hop served by one kernel that stays on the chip -- a doorbell
    # and a completion word per call instead of a launch (sdft_resident.hpp)
    p.set_option("async", 0)
    p.set_option("resident", 1)
    w = 0.0
    for rep in range(2):
        p.synchronize()
        t0 = time.perf_counter()
        for i in range(0, total, hop):
            p.api.sdft_n(p._p, hop, C.c_void_p(xs + i * isz), C.c_void_p(ds))
            p.api.isdft_n(p._p, hop, C.c_void_p(ds), C.c_void_p(ys + i * isz))
        w = (time.perf_counter() - t0) / (total // hop)
        p.synchronize()
    res["us_per_hop_resident_sync"] = round(w * 1e6, 1)
    # ... and both from a C loop (sdft_hip_time_hops: the calls as a C host makes them, without the interpreter's microsecond per call)
    try:
        hops = total // hop
        p.synchronize()
        tc = p.api.time_hops(p._p, hops, hop, C.c_void_p(xs), C.c_void_p(ds), C.c_void_p(ys))
        tc = p.api.time_hops(p._p, hops, hop, C.c_void_p(xs), C.c_void_p(ds), C.c_void_p(ys))
        res["us_per_hop_resident_sync_c_loop"] = round(tc / hops * 1e6, 1)
        p.set_option("resident", 0)
        tc = p.api.time_hops(p._p, hops, hop, C.c_void_p(xs), C.c_void_p(ds), C.c_void_p(ys))
        tc = p.api.time_hops(p._p, hops, hop, C.c_void_p(xs), C.c_void_p(ds), C.c_void_p(ys))
        res["us_per_hop_sync_c_loop"] = round(tc / hops * 1e6, 1)
        p.set_option("resident", 1)
    except Exception as e:
        res["c_loop_error"] = str(e)[:100]
    res["resident"] = {"calls": int(p.get_option("resident_calls")), "launches": int(p.get_option("resident_launches")), "missed": int(p.get_option("resident_missed"))}
    p.set_option("resident", 0)
    # the fused entry point: one call and one launch per hop (process_hop_kernel: folded form, tiles combined in the kernel)
    for mode in ("sync", "async"):
        p.set_option("async", 1 if mode == "async" else 0)
        w = 0.0
        for rep in range(2):
            p.synchronize()
            t0 = time.perf_counter()
            for i in range(0, total, hop):
                p.api.process_n(p._p, hop, C.c_void_p(xs + i * isz), C.c_void_p(ys + i * isz), 0, None, None)
            p.synchronize()
            w = (time.perf_counter() - t0) / (total // hop)
        res[f"us_per_hop_process_n_{mode}"] = round(w * 1e6, 1)
    p.set_option("async", 1); p.set_option("profile", 1)
    for i in range(0, total, hop):
        p.api.sdft_n(p._p, hop, C.c_void_p(xs + i * isz), C.c_void_p(ds))
        p.api.isdft_n(p._p, hop, C.c_void_p(ds), C.c_void_p(ys + i * isz))
    pr = p.profile()
    res["forward_kernel_us"] = round(pr["forward"][0] / max(pr["forward"][1], 1) * 1e3, 1)
    res["inverse_kernel_us"] = round(pr["inverse"][0] / max(pr["inverse"][1], 1) * 1e3, 1)
    p.close()
    # the literal drop-in: malloc'ed x, buffer, y exactly as test.c:62-64 has them (numpy arrays = pageable host memory):
    # default (copies through pinned pieces of the plan: nothing of the caller's is handed to the runtime to pin), with
    # option host_copy = 1 (the runtime's pageable path, for hosts that keep their buffers, like test.c) and with option
    # host_register = 1 (the library maps the caller's buffers once and the kernels work on them over PCIe; likewise)
    xh = sine_sweep(total, dtype=td)
    yh = np.zeros(total, dtype=td)
    dh = np.zeros((hop, m), dtype=np.complex128 if combo[3:] == "f64" else np.complex64)
    def hop_loop(p):
        w = 0.0
        for rep in range(2):
            t0 = time.perf_counter()
            for i in range(0, total, hop):
                p.api.sdft_n(p._p, hop, C.c_void_p(xh.ctypes.data + i * isz), C.c_void_p(dh.ctypes.data))
                p.api.isdft_n(p._p, hop, C.c_void_p(dh.ctypes.data), C.c_void_p(yh.ctypes.data + i * isz))
            w = (time.perf_counter() - t0) / (total // hop)
        return round(w * 1e6, 1)
    p = SDFT(m, "hann", 1.0, combo, device=device)
    res["us_per_hop_host_pointers"] = hop_loop(p)
    p.close()
    # The two legs that hand the caller's memory to the RUNTIME run in a process of their own: the runtime remembers the pins it makes by
    # address, and in a process that has copied and freed large host buffers (this one has) a remembered pin under a new buffer faults the
    # GPU -- the test suite's copy of this leg did, one run in four on some hosts.  A host that allocates its buffers once is safe.
    res.update(_host_pointer_legs_in_a_child(combo, m, hop, total, device))
    res["host_pointers_pcie_floor_us"] = round(2 * hop * m * dh.itemsize / 55e9 * 1e6, 1)      # the matrix out and back in at 55 GB/s
    return res


_HOST_LEGS = r"""
import sys, time, json, ctypes as C, numpy as np
sys.path.insert(0, sys.argv[1])
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
combo, m, hop, total, device = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
td = np.float32 if combo[:3] == "f32" else np.float64
isz = np.dtype(td).itemsize
xh = sine_sweep(total, dtype=td); yh = np.zeros(total, dtype=td)
dh = np.zeros((hop, m), dtype=np.complex128 if combo[3:] == "f64" else np.complex64)
out = {}
for label, reg, rt in (("us_per_hop_host_pointers_runtime_copy", 0, 1), ("us_per_hop_host_pointers_registered", 1, 0)):
    p = SDFT(m, "hann", 1.0, combo, device=device)
    p.set_option("host_register", reg); p.set_option("host_copy", rt)
    w = 0.0
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(0, total, hop):
            p.api.sdft_n(p._p, hop, C.c_void_p(xh.ctypes.data + i * isz), C.c_void_p(dh.ctypes.data))
            p.api.isdft_n(p._p, hop, C.c_void_p(dh.ctypes.data), C.c_void_p(yh.ctypes.data + i * isz))
        w = (time.perf_counter() - t0) / (total // hop)
    out[label] = round(w * 1e6, 1)
    p.close()
print(json.dumps(out))
"""


def _host_pointer_legs_in_a_child(combo, m, hop, total, device):
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", _HOST_LEGS, os.path.dirname(os.path.abspath(__file__)), combo, str(m), str(hop), str(total), str(device)],
                           capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        pass
    return {"us_per_hop_host_pointers_runtime_copy": None, "us_per_hop_host_pointers_registered": None}


def cpu_hop_baseline(np, sine_sweep, combo, td, m=1000, hop=100, total=2000):
    """The reference's own driver loop (test/test.c:69-83) on one host core: microseconds per hop."""
    from oracle import oracle as O
    if not O.have_port(combo):
        O.build(ref=True)
    td_, fd, fdx = O.combo_types(combo)
    plan = O.best(m, "hann", 1.0, combo)
    x = sine_sweep(total, dtype=td)
    buf = np.zeros((hop, m), dtype=fdx)
    best = float("inf")
    for rep in range(2):
        plan.reset()
        t0 = time.perf_counter()
        for i in range(0, total, hop):
            plan.sdft(x[i:i + hop], buf)
            plan.isdft(buf)
        best = min(best, (time.perf_counter() - t0) / (total // hop))
    return {"us_per_hop": round(best * 1e6, 1), "kind": plan.kind, "cores": 1, "sample": f"{total // hop} hops of {hop} samples, dftsize {m}, hann, {combo}"}


def reference_bench_shape(torch, np, SDFT, device, with_cpu=True, placed=True):
    """/root/reference/cpp/examples/bench.cpp:15-48 (rust/examples/bench.rs): dftsize 1000, 44100 zero samples, TD = FD =
    double, Hann, 10 runs, microseconds per sdft / isdft call -- on device pointers, and the reference on one host core."""
    m, n, runs = 1000, 44100, 10
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    d, rb_placement, rb_holder = place_matrix(torch, (n, m), torch.complex128, placed=placed)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    p = SDFT(m, "hann", 1.0, "f64f64", device=device)
    fw, iv = [], []
    for r in range(runs + TUNER_CALLS):                      # (the first calls of a shape settle the synthesis' form: untimed)
        t0 = time.perf_counter(); p.sdft(x, d); t1 = time.perf_counter(); p.isdft(d, y); t2 = time.perf_counter()
        if r >= TUNER_CALLS:
            fw.append(t1 - t0); iv.append(t2 - t1)
    p.close()
    del d
    if rb_holder is not None:
        rb_holder.free()
    res = {"shape": "dftsize 1000, 44100 samples of zeros, TD = FD = double, hann, 10 runs (cpp/examples/bench.cpp:15-48)",
           "gpu_sdft_us": round(float(np.median(fw)) * 1e6, 1), "gpu_isdft_us": round(float(np.median(iv)) * 1e6, 1),
           "matrix": "placed by sdft_hip_malloc_matrix_in_arena (%.0f GB/s store-only against %.0f at the allocation's start)" % (rb_placement.get("window_gbs", 0), rb_placement.get("start_gbs", 0))
                     if rb_placement["placed"] else rb_placement["policy"]}
    if with_cpu:
        from oracle import oracle as O
        if not O.have_port("f64f64"):
            O.build(ref=True)
        ref = O.best(m, "hann", 1.0, "f64f64")
        xh = np.zeros(n); dh = np.zeros((n, m), dtype=np.complex128); yh = np.zeros(n)
        t0 = time.perf_counter(); ref.sdft(xh, dh); t1 = time.perf_counter(); ref.isdft(dh, yh); t2 = time.perf_counter()
        res.update({"cpu_sdft_us": round((t1 - t0) * 1e6, 1), "cpu_isdft_us": round((t2 - t1) * 1e6, 1), "cpu_kind": ref.kind, "cpu_cores": 1,
                    "cpu_runs": 1})
    return res


def baseline_configs(torch, np, SDFT, sine_sweep, device, with_cpu=True, placement=True):
    """The other single-GPU BASELINE.json configs at full size and the single-sample entry points, so that
    the driver's record carries them: configs[2] (round trip, m = 4096, Blackman, FD float, latency 1, n = 262144) and
    configs[3] (64 channels x 48000, m = 2048, Hann, TD float / FD double).  Outside the contract's timed region.  Every
    fraction = algorithmic bytes (dftsize * sizeof(fdx) + sizeof(td) per sample and direction) / time / 8 TB/s; `*_ms_wall`
    is the wall clock of one synchronous call through the C-ABI, `*_kernel_ms` the stage's HIP events (asynchronous calls)."""
    res = {}

    def timed(fn, sync, reps):
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        return (time.perf_counter() - t0) / reps

    # ---- configs[2] ----
    m, n, combo, window = 4096, 262144, "f32f32", "blackman"
    free, _ = torch.cuda.mem_get_info()
    if free < n * m * 8 * 1.1:
        res["config2"] = {"skipped": f"needs {n * m * 8 / 1e9:.1f} GB of free HBM, {free / 1e9:.1f} GB free"}
    else:
        x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
        d_first = torch.empty((n, m), dtype=torch.complex64, device="cuda")        # a plain allocation, timed beside the placed matrix
        d, c2_placement, c2_holder = place_matrix(torch, (n, m), torch.complex64, placed=placement)
        if not c2_placement["placed"]:
            del d; d = d_first
        y = torch.empty(n, dtype=torch.float32, device="cuda")
        p = SDFT(m, window, 1.0, combo, device=device)
        # (the synthesis finds its form and its kind of load on the first calls of a shape -- up to six candidates, two timed calls each, one
        # tuner for syntheses that follow an analysis and one for those that do not: logic::FormTuner; none of that inside a timed region)
        for _ in range(TUNER_CALLS):
            p.sdft(x, d); p.isdft(d, y)
        for _ in range(TUNER_CALLS):
            p.isdft(d, y)
        fwd = timed(lambda: p.sdft(x, d), p.synchronize, 5)
        inv = timed(lambda: p.isdft(d, y), p.synchronize, 5)
        p.sdft(x, d_first)
        fwd_first = timed(lambda: p.sdft(x, d_first), p.synchronize, 5)
        p.set_option("async", 1)
        pair = timed(lambda: (p.sdft(x, d), p.isdft(d, y)), p.synchronize, 5)
        p.set_option("profile", 1)
        for _ in range(5):
            p.sdft(x, d); p.isdft(d, y)
        pr = p.profile()
        calls = max(pr["forward"][1], 1)
        b = n * (m * 8 + 4)
        res["config2"] = {
            "workload": f"BASELINE configs[2]: round trip, n={n}, m={m}, {window}, TD float / FD float, latency 1 (exact carries: bit-identical to the reference)",
            "forward_ms_wall": round(fwd * 1e3, 4), "inverse_ms_wall": round(inv * 1e3, 4), "round_trip_ms_wall_async_pair": round(pair * 1e3, 4),
            "forward_ms_wall_first_allocation": round(fwd_first * 1e3, 4), "forward_frac_of_peak_first_allocation": round(n * (m * 8 + 4) / fwd_first / 1e9 / HBM_PEAK_GBS, 4),
            "buffer_placement": {k: c2_placement.get(k) for k in ("placed", "policy", "arena_bytes", "window_offset", "boundary_offset", "window_gbs", "start_gbs", "pair_probes", "window_probes", "probe_ms")},
            "forward_gbs": round(b / fwd / 1e9, 1), "forward_frac_of_peak": round(b / fwd / 1e9 / HBM_PEAK_GBS, 4),
            "inverse_gbs": round(b / inv / 1e9, 1), "inverse_frac_of_peak": round(b / inv / 1e9 / HBM_PEAK_GBS, 4),
            "round_trip_gbs": round(2 * b / pair / 1e9, 1), "round_trip_frac_of_peak": round(2 * b / pair / 1e9 / HBM_PEAK_GBS, 4),
            "carry_kernel_ms": round((pr["delta"][0] + pr["carry"][0]) / calls, 4),
            "forward_kernel_ms": round(pr["forward"][0] / calls, 4), "inverse_kernel_ms": round(pr["inverse"][0] / max(pr["inverse"][1], 1), 4),
            "carry_form": {0: "serial pass", 1: "chain", 2: "ring", 3: "relay"}.get(p.get_option("last_chain"), "?") + (" + flow mode" if p.get_option("last_flow") == 1 else ""),
            "algorithmic_bytes_per_direction": b, "msamples_s_round_trip": round(n / pair / 1e6, 2),
            "note": "carry and forward stages overlap (the relay runs beside the forward launch): their kernel times do not add up to the wall time",
        }
        p.close(); del x, d, d_first, y
        if c2_holder is not None:
            c2_holder.free()
        torch.cuda.empty_cache()

    # ---- configs[3] ----
    chs, n, m, combo, window = 64, 48000, 2048, "f32f64", "hann"
    need = chs * n * m * 16
    free, _ = torch.cuda.mem_get_info()
    if free < need * 1.03:
        res["config3"] = {"skipped": f"needs {need / 1e9:.1f} GB of free HBM, {free / 1e9:.1f} GB free"}
    else:
        x = torch.from_numpy(np.stack([sine_sweep(n, channel=c, channels=chs, dtype=np.float32) for c in range(chs)])).cuda()
        d, c3_placement, _ = place_matrix(torch, (chs, n, m), torch.complex128, placed=False)      # (100.7 GB: more than two stretches of memory whatever its place)
        c3_placement["policy"] = "first allocation (a matrix of 100.7 GB spans several stretches of memory wherever it lies)"
        p = SDFT(m, window, 1.0, combo, channels=chs, device=device)
        y = None
        for _ in range(2):
            p.sdft(x, d); y = p.isdft(d, y)
        for _ in range(TUNER_CALLS):
            y = p.isdft(d, y)
        fwd = timed(lambda: p.sdft(x, d), p.synchronize, 3)
        inv = timed(lambda: p.isdft(d, y), p.synchronize, 3)
        b = chs * n * (m * 16 + 4)
        res["config3"] = {
            "workload": f"BASELINE configs[3]: {chs} independent channels (one batched plan), n={n} each, m={m}, {window}, TD float / FD double",
            "forward_ms_wall": round(fwd * 1e3, 3), "inverse_ms_wall": round(inv * 1e3, 3),
            "forward_gbs": round(b / fwd / 1e9, 1), "forward_frac_of_peak": round(b / fwd / 1e9 / HBM_PEAK_GBS, 4),
            "inverse_gbs": round(b / inv / 1e9, 1), "inverse_frac_of_peak": round(b / inv / 1e9 / HBM_PEAK_GBS, 4),
            "analysis_msamples_s": round(chs * n / fwd / 1e6, 2), "synthesis_msamples_s": round(chs * n / inv / 1e6, 2),
            "algorithmic_bytes_per_direction": b, "buffer_placement": c3_placement,
        }
        p.close(); del x, d, y
        torch.cuda.empty_cache()

    # ---- single-sample entry points (sdft.h:562, :635): one launch and one completion per call ----
    m, combo = 1024, "f32f64"
    p = SDFT(m, "hann", 1.0, combo, device=device)
    row = torch.empty(m, dtype=torch.complex128, device="cuda")
    xs = sine_sweep(4000, dtype=np.float32)
    rp = C.c_void_p(row.data_ptr())
    for i in range(200):
        p.api.sdft(p._p, float(xs[i]), rp)
    t0 = time.perf_counter()
    for i in range(200, 2200):
        p.api.sdft(p._p, float(xs[i]), rp)
    t1 = time.perf_counter()
    for i in range(2000):
        p.api.isdft(p._p, rp)
    t2 = time.perf_counter()
    hrow = np.zeros(m, dtype=np.complex128)
    hp = C.c_void_p(hrow.ctypes.data)
    for i in range(100):
        p.api.sdft(p._p, float(xs[i]), hp)
    t3 = time.perf_counter()
    for i in range(1000):
        p.api.sdft(p._p, float(xs[i]), hp)
    t4 = time.perf_counter()
    # ... and with option "resident" = 1: one pair per sample (the reference's per-sample loop), no launch per call
    p.set_option("resident", 1)
    for i in range(200):
        p.api.sdft(p._p, float(xs[i]), rp); p.api.isdft(p._p, rp)
    t5 = time.perf_counter()
    for i in range(200, 2200):
        p.api.sdft(p._p, float(xs[i]), rp); p.api.isdft(p._p, rp)
    t6 = time.perf_counter()
    res_calls = int(p.get_option("resident_calls"))
    p.set_option("resident", 0)
    p.close()
    one = {"shape": f"sdft_sdft / sdft_isdft, one sample per call, m={m}, hann, {combo}, synchronous (the sample comes back by value)",
           "sdft_us_per_call_device_row": round((t1 - t0) / 2000 * 1e6, 2), "isdft_us_per_call_device_row": round((t2 - t1) / 2000 * 1e6, 2),
           "sdft_us_per_call_host_row": round((t4 - t3) / 1000 * 1e6, 2),
           "sdft_plus_isdft_us_per_sample_resident": round((t6 - t5) / 2000 * 1e6, 2), "resident_calls": res_calls}
    if with_cpu:
        from oracle import oracle as O
        ref = O.best(m, "hann", 1.0, combo)
        buf = np.zeros((2000, m), dtype=np.complex128)
        ref.sdft(xs[:2000], buf)
        ta = time.perf_counter(); ref.sdft(xs[:2000], buf); tb = time.perf_counter(); ref.isdft(buf); tc = time.perf_counter()
        one.update({"cpu_sdft_us_per_sample": round((tb - ta) / 2000 * 1e6, 2), "cpu_isdft_us_per_sample": round((tc - tb) / 2000 * 1e6, 2),
                    "cpu_kind": ref.kind, "cpu_cores": 1,
                    "note": "per sample, the drop-in costs a launch and a completion: on a par with one host core; with option resident = 1 a doorbell instead (sdft_plus_isdft_us_per_sample_resident "
                            "for the pair); hosts that have the samples call sdft_sdft_n"})
    res["single_sample"] = one
    return res


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (torch.distributed.run, one rank per
    GPU, rendezvous on 127.0.0.1), relay its output -- rank 0's one JSON line included -- and hand back its return code.
    Nothing in this process has touched the GPU (torch.cuda.device_count() does not initialise it on this image), and this
    process never replaces itself with another program."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    backend = os.environ.get("SDFT_BENCH_BACKEND", "nccl")
    if have < args.gpus and backend == "nccl":
        print(f"[bench] --gpus {args.gpus} but this node shows {have} GPU(s): refusing to print a line that is not a {args.gpus}-GPU "
              f"measurement (SDFT_BENCH_BACKEND=gloo runs the ranks on the GPUs there are, for functional tests only)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd[1:9])} bench.py ...", file=sys.stderr)
    r = subprocess.run(cmd, env=env, cwd=ROOT)
    return r.returncode


TUNER_CALLS = 14        # untimed calls of a shape before its synthesis is timed (FormTuner: up to 6 candidates x 2 samples)


def place_matrix(torch, shape, cdt, placed=True, arena_extra=64 << 30):
    """The output matrix.  Which memory backs a large buffer decides how fast it can be WRITTEN (rounds 5 and 6: device memory comes in kinds
    that alternate every 16-32 GiB of an allocation, and the analysis' store stream takes a matrix that lies half in one kind and half in
    another at 6.8-7.1 TB/s, one that lies in a single kind at 5.6-5.85: profiles/r06_stretch_map.txt) -- a property of the allocation, not of
    the kernel.  placed: the matrix comes from the LIBRARY's call for exactly this, `sdft_hip_malloc_matrix_in_arena(bytes, bytes + 64 GiB)` --
    what include/sdft/sdft_hip.h gives a C host -- which finds where the kind of memory changes with a few 2 GiB probes and returns the window
    centred there (untimed, before the warm-up; the line carries the call's own record: `buffer_placement`).  Not placed (--no-placement, shapes
    the call does not cover, or a rank whose arena does not fit): the first allocation as it comes.  Returns (tensor, info, holder)."""
    import math
    from sdft_amd import capi
    nbytes = math.prod(shape) * torch.empty(0, dtype=cdt).element_size()
    why = "--no-placement" if not placed else None
    if placed and not ((64 << 20) <= nbytes < (96 << 30)):
        why = "no placement for this size"
    if why is None:
        free, _ = torch.cuda.mem_get_info()
        if free < nbytes + arena_extra + (8 << 30):
            why = f"the arena does not fit ({free / 1e9:.0f} GB free)"
    if why is None:
        try:
            pm = capi.PlacedMatrix(shape, cdt, arena_extra)
            info = {"policy": "sdft_hip_malloc_matrix_in_arena(bytes, bytes + 64 GiB): the window centred on the first change of the kind of memory "
                              "inside one allocation, found by two-part store probes; untimed, before the warm-up (--no-placement: the first allocation)",
                    "placed": True, "matrix_bytes": nbytes}
            info.update(pm.info)
            return pm.tensor, info, pm
        except Exception as e:                                   # (out of memory on this rank: say so, take the first allocation)
            why = f"sdft_hip_malloc_matrix_in_arena failed: {str(e)[:120]}"
    t = torch.empty(shape, dtype=cdt, device="cuda")
    return t, {"policy": f"first allocation ({why})", "placed": False, "matrix_bytes": nbytes}, None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if world != args.gpus:
        # a record that says n_gpus = WORLD_SIZE while the command line asked for something else helps nobody
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # one rank per GPU; SDFT_BENCH_BACKEND=gloo lets several ranks share a GPU for functional tests
    backend = os.environ.get("SDFT_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world if distributed else 1

    from sdft_amd import shard
    from sdft_amd.sdft import SDFT
    from sdft_amd.signals import sine_sweep

    m, window, combo = args.m, args.window, args.combo
    td = np.float32 if combo[:3] == "f32" else np.float64
    esz = 16 if combo[3:] == "f64" else 8
    cdt = torch.complex128 if esz == 16 else torch.complex64
    workload = args.workload
    if workload == "auto":
        workload = "single" if n_gpus == 1 else "batch"
    if workload == "single":
        n = args.n or 1_000_000
        channels_total = n_gpus
        first, count = rank, 1
        name = (f"BASELINE configs[1]: forward sdft_sdft_n, n={n}, m={m}, {window}, TD {combo[:3]} / FD {combo[3:]}, "
                f"1 channel per GPU")
    else:
        n = args.n or 48000
        channels_total = shard.weak_scaling_channels(args.channels_per_gpu, n_gpus)
        first, count = shard.channel_block(channels_total, n_gpus, rank)
        name = (f"BASELINE configs[4]: forward sdft_sdft_n, {channels_total} channels sharded {args.channels_per_gpu}/GPU, "
                f"n={n} each, m={m}, {window}, TD {combo[:3]} / FD {combo[3:]}")

    # synthetic input, resident in HBM before the timed region
    if count == 1:
        xh = sine_sweep(n, channel=first, channels=max(channels_total, 1), dtype=td)
    else:
        xh = np.stack([sine_sweep(n, channel=c, channels=channels_total, dtype=td) for c in range(first, first + count)])
    x = torch.from_numpy(xh).cuda()
    shape = (n, m) if count == 1 else (count, n, m)
    # The matrix a plain host has: the FIRST allocation of the process, as hipMalloc hands it out (the reference's contract is a caller-allocated
    # matrix, sdft.h:605).  Timed below with the same --steps as the headline and reported beside it (`first_allocation`).
    out_first = torch.empty(shape, dtype=cdt, device="cuda")
    # ... and the matrix a host gets that asks the library for the memory (include/sdft/sdft_hip.h): the headline's
    out, placement, out_holder = place_matrix(torch, shape, cdt, placed=not args.no_placement)
    if not placement["placed"]:
        del out
        out = out_first                                       # (no second matrix: the headline IS the first allocation)

    stream = torch.cuda.Stream()
    plan = SDFT(m, window, 1.0, combo, channels=count, device=local_rank)
    plan.set_stream(stream.cuda_stream)
    plan.set_option("async", 1)
    plan.set_option("profile", 1)

    def sync():
        plan.synchronize()
        torch.cuda.synchronize()

    def timed_region(matrix):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides (the contract's region)."""
        plan.set_option("profile", 1)
        for w in range(args.warmup):
            plan.sdft(x, matrix)
            if w == 0 and args.warmup > 1:
                sync(); plan.profile()       # the first call allocates the workspace: not representative
        sync()
        warm = plan.profile()                # warm-up pass: per-stage events (delta, carries, forward)
        plan.set_option("profile", 2)        # timed region: only the event pair around the dominant kernel
        shard.barrier(local_rank)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            plan.sdft(x, matrix)
        sync()
        shard.barrier(local_rank)
        sync()
        return time.perf_counter() - t0, plan.profile(), warm

    elapsed, prof, warm = timed_region(out)
    prepass_ms = (warm["delta"][0] + warm["carry"][0]) / max(warm["forward"][1], 1)

    units = float(count * n * args.steps)
    rate, secs = shard.job_throughput(units, elapsed, local_rank)
    local_elapsed = elapsed
    bytes_per_launch = count * n * (m * esz + np.dtype(td).itemsize)
    # the same region into the process' first allocation (every rank; the same K and W)
    # (every rank runs the region, placed or not: its barriers are collective)
    first_elapsed, first_prof, _ = timed_region(out_first)
    first_rate, first_secs = shard.job_throughput(units, first_elapsed, local_rank)
    ff_ms, ff_calls = first_prof["forward"]
    first_kernel_gbs = bytes_per_launch / (ff_ms / max(ff_calls, 1) * 1e-3) / 1e9 if ff_ms > 0 else 0.0
    first_allocation = {
        "value": round(first_rate / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(first_secs / args.steps * 1e3, 4),
        "frac": round(first_kernel_gbs / HBM_PEAK_GBS, 4), "frac_is": "the kernel's algorithmic bytes / its HIP-event time / 8 TB/s, as roofline.frac",
        "is": "the same K timed steps into the process' FIRST allocation of the matrix' size (a plain hipMalloc, what a host of the reference has: "
              "sdft.h:605 'already allocated'); `value` is the matrix the library placed (buffer_placement)" if out is not out_first else
              "this IS the headline: the matrix was not placed",
    }
    if out is not out_first:
        del out_first                                         # (the side measurements need the room)
        torch.cuda.empty_cache()

    # the same step one at a time (SURVEY.md 8d asks for a median): wall clock around one synchronised call each, and the
    # kernel's own HIP events launch by launch; outside the contract's timed region above
    step_ms, kern_ms = [], []
    for _ in range(min(args.steps, 20)):
        sync()
        ts = time.perf_counter()
        plan.sdft(x, out)
        sync()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        pf = plan.profile()["forward"]
        if pf[1]:
            kern_ms.append(pf[0] / pf[1])

    # second bracketed region, every rank: analysis + synthesis pairs (the metric string names both)
    # (long synthesis calls find the fastest of their bit-identical forms and kinds of load on the first calls of a shape -- the ones that follow an
    # analysis call separately from the ones that do not: logic::FormTuner; the pairs below are of the first kind, so pairs are what settles it)
    y = plan.isdft(out)
    for _ in range(TUNER_CALLS):
        plan.sdft(x, out)
        plan.isdft(out, y)
        sync()
    sync(); plan.profile()
    shard.barrier(local_rank)
    sync()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        plan.sdft(x, out)
        plan.isdft(out, y)
    sync()
    shard.barrier(local_rank)
    sync()
    elapsed_rt = time.perf_counter() - t1
    prof_rt = plan.profile()
    rate_rt, secs_rt = shard.job_throughput(units, elapsed_rt, local_rank)
    i_ms, i_calls = prof_rt["inverse"]
    i_avg = i_ms / max(i_calls, 1)
    syn_rate = shard.sum_over_ranks(count * n / (i_avg * 1e-3) if i_avg > 0 else 0.0, local_rank)
    # (outside both bracketed regions: the synthesis of a matrix that is only read -- call after call, no analysis in between; this rank's calls)
    for _ in range(TUNER_CALLS):
        plan.isdft(out, y)
        sync()
    plan.profile()
    for _ in range(5):
        plan.isdft(out, y)
    sync()
    ro_ms, ro_calls = plan.profile()["inverse"]
    syn_read_only = count * n / (ro_ms / max(ro_calls, 1) * 1e-3) if ro_ms > 0 else 0.0

    # roofline of the dominant kernel (this rank's launches; every rank runs the same shape)
    f_ms, f_calls = prof["forward"]
    f_avg_ms = f_ms / max(f_calls, 1)
    achieved = bytes_per_launch / (f_avg_ms * 1e-3) / 1e9 if f_avg_ms > 0 else 0.0
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            t = json.load(fh)
        for entry in (t if isinstance(t, list) else [t]):
            if entry.get("workload") == workload and entry.get("n") == n and entry.get("m") == m and entry.get("channels") == count:
                traffic = entry.get("bytes_per_launch")
    except Exception:
        pass

    # N > 1: what makes the record checkable without logs -- how many ranks the collectives saw, on how many distinct GPUs,
    # and the spread of the per-rank step time and roofline fraction (every rank runs the same shape on its own channels)
    census = shard.run_census(local_elapsed / args.steps * 1e3, achieved / HBM_PEAK_GBS, local_rank, placement_gbs=float(placement.get("window_gbs", 0.0)),
                              placed=bool(placement["placed"]), first_frac=first_kernel_gbs / HBM_PEAK_GBS) if distributed else None
    kernel_names = {1: "forward_kernel", 2: "forward_rows_kernel", 3: "forward_hop_kernel"}
    result = {
        "metric": "Msamples/s analysis+synthesis, m=1024 Hann fp64; achieved HBM GB/s vs peak",
        "value": round(rate / 1e6, 3),
        "unit": "Msamples/s",
        "value_is": "analysis only (sdft_sdft_n), as configs[1] states; the analysis+synthesis pair rate is the next field",
        "analysis_plus_synthesis_msamples_s": round(rate_rt / 1e6, 3),
        "synthesis_msamples_s": round(syn_rate / 1e6, 3),
        "synthesis_is": "the synthesis calls of the analysis + synthesis pairs (each reads the matrix the analysis has just written), by the plan's events; "
                        "synthesis_matrix_only_read_msamples_s: call after call on a matrix nobody writes (this rank, outside the timed regions)",
        "synthesis_matrix_only_read_msamples_s": round(syn_read_only / 1e6, 3),
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(secs / args.steps * 1e3, 4),
        "ms_per_step_median_one_at_a_time": round(float(np.median(step_ms)), 4) if step_ms else None,
        "ms_per_step_analysis_plus_synthesis": round(secs_rt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if esz == 16 else "f32",
        "data": "synthetic",
        "first_allocation": first_allocation,
        "buffer_placement": placement,
        "config": {
            "workload": name,
            "channels_total": channels_total,
            "samples_per_channel": n,
            "dftsize": m,
            "window": window,
            "types": combo,
            "step": "analysis (sdft_sdft_n) on device-resident buffers",
            "time_chunks": plan.get_option("last_chunks"),
            "chunk_len": plan.get_option("last_chunk_len"),
            "carry_mode": "exact" if plan.get_option("carry") else "fast",
        },
        "ranks": None if census is None else {
            "collective_backend": census["backend"], "ranks_in_collectives": census["ranks"], "world_size": census["world_size"],
            "distinct_local_devices": census["distinct_local_devices"],
            "ms_per_step_min_over_ranks": round(census["seconds_min"], 4), "ms_per_step_max_over_ranks": round(census["seconds_max"], 4),
            "roofline_frac_min_over_ranks": round(census["roofline_frac_min"], 4), "roofline_frac_max_over_ranks": round(census["roofline_frac_max"], 4),
            "per_gpu_msamples_s": round(rate / 1e6 / max(n_gpus, 1), 3),
            "ranks_with_a_placed_matrix": census["ranks_placed"],
            "placement_gbs_min_over_ranks": round(census["placement_gbs_min"], 1), "placement_gbs_max_over_ranks": round(census["placement_gbs_max"], 1),
            "first_allocation_frac_min_over_ranks": round(census["first_allocation_frac_min"], 4),
            "first_allocation_frac_max_over_ranks": round(census["first_allocation_frac_max"], 4),
            "placement_note": "buffer_placement is rank 0's; a rank whose arena did not fit took its first allocation (placement_gbs 0 in the minimum)",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_names.get(plan.get_option("last_kernel"), "?"),
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": ("profiles/hbm_traffic.json (PMC passes of this workload taken with rocprofv3 --pmc in the builder's session and "
                               "replayed here; this run measured no counters)") if traffic is not None else None,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_ms": round(f_avg_ms, 4),
            "median_launch_ms": round(float(np.median(kern_ms)), 4) if kern_ms else None,
            "launches": f_calls,
            "prepass_ms_per_step": round(prepass_ms, 4),
            "synthesis_read_gbs": round(bytes_per_launch / (i_avg * 1e-3) / 1e9, 1) if i_avg > 0 else None,
        },
    }

    # side measurements outside the timed regions (rank 0, single GPU)
    if rank == 0 and not distributed and not args.no_extras:
        extras = {}
        from sdft_amd import capi
        lib = capi.load()
        # achievable-write ceiling: kernels that do nothing but the store stream,
        # (a) plain linear fill, (b) the forward kernel's own shape (one workgroup per time chunk
        # writing whole rows in lockstep, barrier every 8 rows), over the same output buffer
        if esz == 16 and m % 64 == 0 and m <= 1024:
            nbytes = count * n * m * esz
            sync()
            lin = lib.sdft_hip_store_ceiling(out.data_ptr(), nbytes, 0, m, 64, 1, 5)
            grp = lib.sdft_hip_store_ceiling(out.data_ptr(), nbytes, 2, m, 8, max(int(plan.get_option("last_chunk_len")), 1), 5)
            # (c) the same with every XCD writing a contiguous eighth of the matrix -- the placement the analysis launches take
            # since round 5 (ForwardArgs::xcd_map): the workgroups running at the same time are spread over the whole matrix
            spr = lib.sdft_hip_store_ceiling(out.data_ptr(), nbytes, 4, m, 8, max(int(plan.get_option("last_chunk_len")), 1), 5)
            torch.cuda.synchronize()
            best_ms = min(v for v in (lin, grp, spr) if v > 0)
            ld = lib.sdft_hip_load_ceiling(out.data_ptr(), nbytes, 5)
            torch.cuda.synchronize()
            if ld > 0 and i_avg > 0:
                result["roofline"]["load_only_ceiling"] = {
                    "linear_load_gbs": round(nbytes / (ld * 1e-3) / 1e9, 1),
                    "synthesis_frac_of_load_only": round((bytes_per_launch / (i_avg * 1e-3)) / (nbytes / (ld * 1e-3)), 4),
                }
            result["roofline"]["store_only_ceiling"] = {
                "linear_fill_gbs": round(nbytes / (lin * 1e-3) / 1e9, 1),
                "row_lockstep_gbs": round(nbytes / (grp * 1e-3) / 1e9, 1),
                "row_lockstep_xcd_contiguous_gbs": round(nbytes / (spr * 1e-3) / 1e9, 1) if spr > 0 else None,
                "frac_of_best_store_only": round(achieved / (nbytes / (best_ms * 1e-3) / 1e9), 4),
            }
        # fused analysis -> synthesis (sdft_hip_process_n): same input, no matrix traffic; VALU-bound, so
        # it is priced against the fp64 vector peak next to the two-pass figure it replaces
        fp = {}
        for label, fe in (("tree_sum", 0), ("reference_order", 1)):
            plan.set_option("fused_exact", fe)
            yf = plan.process(x)
            sync()
            tf = time.perf_counter()
            for _ in range(5):
                plan.process(x, out=yf)
            sync()
            wf = (time.perf_counter() - tf) / 5
            fp[label + "_msamples_s"] = round(count * n / wf / 1e6, 1)
            fp[label + "_ms"] = round(wf * 1e3, 3)
            if fe:
                # float samples: the reference's float is proven from the tree sum for most samples; the rest are walked in order
                fp["reference_order_walked_frac"] = round(plan.get_option("ordered_walks") / (6.0 * count * n), 5)
        plan.set_option("fused_exact", -1)
        # round 3: gains that change every 512 samples (one launch: the folded coefficients of all gain vectors come from one
        # small launch in front), and a spectral gate (not linear: the windowed rows in LDS, tree sum)
        if count == 1:
            rows_g = torch.rand(((n + 511) // 512, m), dtype=torch.float64 if esz == 16 else torch.float32, device="cuda") + 0.5
            # ... and the same gate handed in as code (sdft_hip_op_expr: compiled into the fused kernel at run time; the
            # compilation is outside the timed calls and reported beside them)
            gate_code = "if (re * re + im * im < p[0] * p[0]) { re = 0; im = 0; }"
            for label, kw in (("gain_rows_hop512", dict(op="gain_rows", gain=rows_g, hop=512)), ("gate", dict(op="gate", threshold=1e-3, floor=0.0)),
                              ("gate_as_expression", dict(op="expr", expr=gate_code, expr_params=[1e-3]))):
                tc = time.perf_counter()
                yf = plan.process(x, **kw)
                sync()
                if label == "gate_as_expression":
                    fp["expression_first_call_s"] = round(time.perf_counter() - tc, 2)
                tf = time.perf_counter()
                for _ in range(5):
                    plan.process(x, out=yf, **kw)
                sync()
                wf = (time.perf_counter() - tf) / 5
                fp[label + "_msamples_s"] = round(count * n / wf / 1e6, 1)
                fp[label + "_ms"] = round(wf * 1e3, 3)
            del rows_g
        # the same call on HOST buffers (what a host of the reference has: malloc'ed samples in, samples out): 4 bytes per
        # sample each way over PCIe instead of the 16 KiB per sample of the matrix -- PCIe-inclusive, never `value`
        xh_f = xh if count > 1 else np.ascontiguousarray(xh)
        yh_f = plan.process(xh_f)
        th = time.perf_counter()
        for _ in range(3):
            plan.process(xh_f, out=yh_f)
        wh = (time.perf_counter() - th) / 3
        fp["host_pointers_msamples_s"] = round(count * n / wh / 1e6, 1)
        fp["host_pointers_ms"] = round(wh * 1e3, 3)
        # folded form (process_rows_kernel, FD double, round 3): the demodulated bin is carried through the chunk, X' = (X + d) * conj(tw):
        # 1 addition, 2 multiplications and 2 fused multiply-adds, then 1 fused multiply-add for the coefficient
        # = 6 vector instructions = 9 flops at 2 per FMA (round 2: 9 instructions, 15 flops)
        flops, instr = 9, 6
        fp["two_pass_msamples_s"] = round(rate_rt / 1e6, 1)
        fp["speedup_vs_two_pass"] = round(fp["tree_sum_msamples_s"] / max(rate_rt / 1e6, 1e-9), 2)
        fp["fp64_vector_tflops"] = round(count * n * m * flops / (fp["tree_sum_ms"] * 1e-3) / 1e12, 2)
        fp["fp64_vector_peak_tflops"] = 78.6
        fp["fp64_instruction_slots_frac"] = round(count * n * m * instr * 2 / (fp["tree_sum_ms"] * 1e-3) / 78.6e12, 3)
        fp["fp64_instructions_per_bin_sample"] = instr
        fp["note"] = ("tree-sum flavour = folded form: window, operation and synthesis folded into per-bin coefficients, the demodulated "
                      "bin carried through the chunk: 6 fp64 vector instructions (9 flops, an FMA counted as two) per bin-sample (round 2: 9 and 15); "
                      "fp64_instruction_slots_frac prices every instruction as an FMA slot of the 78.6 TFLOP/s peak -- with a third fewer "
                      "instructions to issue the fraction says less than the time does")
        result["fused_process"] = fp
        if workload == "single":
            result["north_star_n48000"] = north_star_shape(torch, np, SDFT, sine_sweep, cdt, m, window, combo, esz, td, local_rank, placed=not args.no_placement)
        if workload == "single" and count == 1:
            # the headline workload as a host that alternates between two matrices runs it: asynchronous calls on the plan's own
            # stream, pipelined; outside the contract's timed region, which stays one matrix on the caller's stream
            out2_holder = None
            try:
                # (both matrices placed the same way: each the window of an arena of its own, bytes + 64 GiB -- round 5 compared a placed
                # matrix with the best of four separate allocations)
                out2, place2, out2_holder = place_matrix(torch, tuple(out.shape), cdt, placed=placement["placed"])
                tm = {"workload": name, "placement": {"first": placement.get("window_gbs"), "second": place2.get("window_gbs"),
                                                        "is": "store-only GB/s of the two matrices, each placed by sdft_hip_malloc_matrix_in_arena" if place2["placed"] else place2["policy"]}}
                ptr = [C.c_void_p(out.data_ptr()), C.c_void_p(out2.data_ptr())]
                xptr = C.c_void_p(x.data_ptr())
                bq = n * (m * esz + np.dtype(td).itemsize)
                for label, pipe in (("pipelined", 2), ("one_stream", 0), ("library_default", None)):
                    pp = SDFT(m, window, 1.0, combo, device=local_rank)
                    pp.set_option("async", 1)
                    if pipe is not None:
                        pp.set_option("pipeline", pipe)
                    for i in range(4):
                        pp.api.sdft_n(pp._p, n, xptr, ptr[i & 1])
                    pp.synchronize(); torch.cuda.synchronize()
                    tq = time.perf_counter()
                    for i in range(10):
                        pp.api.sdft_n(pp._p, n, xptr, ptr[i & 1])
                    pp.synchronize(); torch.cuda.synchronize()
                    wq = (time.perf_counter() - tq) / 10
                    tm[label] = {"ms_per_call_wall": round(wq * 1e3, 4), "msamples_s": round(n / wq / 1e6, 1), "frac_of_peak_wall": round(bq / wq / 1e9 / HBM_PEAK_GBS, 4),
                                 "pipelined_calls": int(pp.get_option("pipelined_calls")),
                                 "chunks": [int(pp.get_option("last_chunks")), int(pp.get_option("last_chunk_len"))]}
                    if pipe is None:
                        # ... and the synthesis of the two matrices in turn (stateless: the calls go to the two row streams in turn)
                        y2 = [torch.empty(n, dtype=x.dtype, device="cuda") for _ in range(2)]
                        yptr = [C.c_void_p(y2[0].data_ptr()), C.c_void_p(y2[1].data_ptr())]
                        for i in range(2 * TUNER_CALLS):
                            pp.api.isdft_n(pp._p, n, ptr[i & 1], yptr[i & 1])
                            if i >= 2:
                                pp.synchronize()
                        pp.synchronize(); torch.cuda.synchronize()
                        tq = time.perf_counter()
                        for i in range(10):
                            pp.api.isdft_n(pp._p, n, ptr[i & 1], yptr[i & 1])
                        pp.synchronize(); torch.cuda.synchronize()
                        wq = (time.perf_counter() - tq) / 10
                        tm.update({"synthesis_ms_per_call_wall": round(wq * 1e3, 4), "synthesis_msamples_s": round(n / wq / 1e6, 1),
                                   "synthesis_frac_of_peak_wall": round(bq / wq / 1e9 / HBM_PEAK_GBS, 4),
                                   "pipelined_synthesis_calls": int(pp.get_option("pipelined_inverse_calls"))})
                        del y2
                    pp.close()
                tm["note"] = ("asynchronous sdft_sdft_n calls on the plan's own stream into two matrices in turn: option pipeline = 2 (the rows of consecutive calls on two "
                              "streams whatever the length), = 0 (one stream), and the library's default (1), which pipelines only calls of less than 2^29 bins (a tie at this length: profiles/r06_pipelined_calls.txt)")
                result["two_matrices_in_turn"] = tm
                del out2
            except Exception as e:                              # (a second 16 GB matrix: not on every box)
                result["two_matrices_in_turn"] = {"error": str(e)[:200]}
            if out2_holder is not None:
                out2_holder.free()
        result["hop100_m1000"] = hop_streaming(torch, np, SDFT, sine_sweep, combo, td, cdt, local_rank)
        if not args.no_cpu_baseline:
            result["hop100_m1000"]["cpu_reference"] = cpu_hop_baseline(np, sine_sweep, combo, td)
        result["reference_bench_shape"] = reference_bench_shape(torch, np, SDFT, local_rank, with_cpu=not args.no_cpu_baseline, placed=not args.no_placement)

        # PCIe-inclusive host-pointer path (never `value`)
        npci = min(n, 65536)
        hx = xh[..., :npci].copy() if count == 1 else np.ascontiguousarray(xh[:, :npci])
        hp = SDFT(m, window, 1.0, combo, channels=count, device=local_rank)
        hout = np.empty((npci, m) if count == 1 else (count, npci, m), dtype=np.complex128 if esz == 16 else np.complex64)
        hp.sdft(hx, hout)
        tpci = []
        for _ in range(3):
            hp.reset()
            tp = time.perf_counter()
            hp.sdft(hx, hout)
            tpci.append(time.perf_counter() - tp)
        extras["host_pointer_pcie_inclusive_msamples_s"] = round(count * npci / min(tpci) / 1e6, 3)
        extras["host_pointer_pcie_inclusive_is"] = f"one synchronous sdft_sdft_n call, host samples in, host matrix out ({count * npci * m * esz / 1e9:.2f} GB over PCIe), best of 3 (median {count * npci / sorted(tpci)[1] / 1e6:.3f})"
        hp.close()

        if workload == "single":
            # one GPU's share of configs[4] (64 channels x 48000): the like-for-like N = 1 point of the
            # 1 -> 8 GPU curve the driver builds from the N > 1 runs of this script
            plan.close(); del out, y
            if out_holder is not None:
                out_holder.free(); out_holder = None
            torch.cuda.empty_cache()
            chs, nb_ = args.channels_per_gpu, 48000
            free, _ = torch.cuda.mem_get_info()
            if free > chs * nb_ * m * esz * 1.05:
                xb = torch.from_numpy(np.stack([sine_sweep(nb_, channel=c, channels=chs, dtype=td) for c in range(chs)])).cuda()
                ob_first = torch.empty((chs, nb_, m), dtype=cdt, device="cuda")        # (what N > 1 runs report as first_allocation)
                ob, share_placement, ob_holder = place_matrix(torch, (chs, nb_, m), cdt, placed=not args.no_placement)
                if not share_placement["placed"]:
                    del ob; ob = ob_first
                pb = SDFT(m, window, 1.0, combo, channels=chs, device=local_rank)
                pb.set_option("async", 1)
                yb = None
                for _ in range(2):
                    pb.sdft(xb, ob); yb = pb.isdft(ob, yb)
                for _ in range(TUNER_CALLS):
                    yb = pb.isdft(ob, yb); pb.synchronize()
                pb.synchronize(); torch.cuda.synchronize()
                reps = 5
                tb = time.perf_counter()
                for _ in range(reps):
                    pb.sdft(xb, ob)
                pb.synchronize(); torch.cuda.synchronize()
                wa = (time.perf_counter() - tb) / reps
                tb = time.perf_counter()
                for _ in range(reps):
                    pb.isdft(ob, yb)
                pb.synchronize(); torch.cuda.synchronize()
                ws = (time.perf_counter() - tb) / reps
                bb = chs * nb_ * (m * esz + np.dtype(td).itemsize)
                share = {
                    "workload": f"one GPU's share of BASELINE configs[4]: {chs} channels x n={nb_}, m={m}, {window}, {combo}",
                    "analysis_msamples_s": round(chs * nb_ / wa / 1e6, 2), "analysis_gbs": round(bb / wa / 1e9, 1),
                    "analysis_frac_of_peak": round(bb / wa / 1e9 / HBM_PEAK_GBS, 4),
                    "synthesis_msamples_s": round(chs * nb_ / ws / 1e6, 2), "synthesis_gbs": round(bb / ws / 1e9, 1),
                    "analysis_plus_synthesis_msamples_s": round(chs * nb_ / (wa + ws) / 1e6, 2),
                    "ms_per_call_analysis": round(wa * 1e3, 3),
                    "buffer_placement": share_placement,
                }
                if ob is not ob_first:
                    tb = time.perf_counter()
                    for _ in range(reps):
                        pb.sdft(xb, ob_first)
                    pb.synchronize(); torch.cuda.synchronize()
                    wf = (time.perf_counter() - tb) / reps
                    share["first_allocation"] = {"analysis_msamples_s": round(chs * nb_ / wf / 1e6, 2), "analysis_frac_of_peak": round(bb / wf / 1e9 / HBM_PEAK_GBS, 4)}
                pb.close(); del ob, ob_first, xb, yb
                if ob_holder is not None:
                    ob_holder.free()
                torch.cuda.empty_cache()
                if not args.no_cpu_all_cores:
                    import subprocess
                    procs = min(os.cpu_count() or 1, 64)
                    r = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--procs", str(procs), "--n", "16384", "--m", str(m),
                                        "--window", window, "--combo", combo], capture_output=True, text=True, cwd=ROOT, timeout=600)
                    try:
                        share["cpu_baseline_all_cores"] = json.loads(r.stdout.strip().splitlines()[-1])
                    except Exception:
                        share["cpu_baseline_all_cores"] = {"error": r.stderr[-300:]}
                result["batch_share"] = share
            plan = None
            result["configs"] = baseline_configs(torch, np, SDFT, sine_sweep, local_rank, with_cpu=not args.no_cpu_baseline, placement=not args.no_placement)
        result["extras"] = extras

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(m, window, combo, args.cpu_samples)
    elif rank == 0:
        result["cpu_baseline"] = None

    # an N-GPU line must BE an N-GPU measurement: every rank in the collectives, and under RCCL every rank on a GPU of its own
    bad = None
    if census is not None:
        if census["ranks"] != n_gpus:
            bad = f"{census['ranks']} ranks took part in the collectives, --gpus says {n_gpus}"
        elif backend == "nccl" and census["distinct_local_devices"] != n_gpus:
            bad = f"the {n_gpus} ranks sit on {census['distinct_local_devices']} distinct GPU(s)"
    if rank == 0:
        if bad:
            print(f"[bench] NOT a valid {n_gpus}-GPU line: {bad}", file=sys.stderr)
            result["invalid"] = bad
        print(json.dumps(result), flush=True)
    if plan is not None:
        plan.close()
    if distributed:
        shard.barrier(local_rank)        # rank 0 may still have been busy with its side measurements
        dist.destroy_process_group()
    if bad:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
