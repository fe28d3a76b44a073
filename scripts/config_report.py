"""Runs every BASELINE.json config shape on one GPU through the C-ABI (device pointers) and prints a
markdown table: per-stage device times (HIP events recorded by the library), Msamples/s and
achieved bandwidth against the algorithmic bytes.  Development/report aid; bench.py is the
contract benchmark.

    python scripts/config_report.py > profiles/rNN_configs.md
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

ROWS = []


def run(label, n, m, window, combo, channels=1, reps=5, roundtrip=True, **opts):
    td = np.float32 if combo[:3] == "f32" else np.float64
    esz = 16 if combo[3:] == "f64" else 8
    cdt = torch.complex128 if esz == 16 else torch.complex64
    free, _ = torch.cuda.mem_get_info()
    if channels * n * m * esz * 1.05 > free:
        ROWS.append(f"| {label} | skipped: needs {channels * n * m * esz / 1e9:.0f} GB |" + " |" * 8)
        return
    xh = sine_sweep(n, dtype=td) if channels == 1 else np.stack([sine_sweep(n, channel=c, channels=channels, dtype=td) for c in range(channels)])
    x = torch.from_numpy(xh).cuda()
    # the matrix: placed by the library's own call where its arena fits (sdft_hip_malloc_matrix_in_arena, as bench.py's headline), else a plain allocation
    shape = (n, m) if channels == 1 else (channels, n, m)
    nbytes = channels * n * m * esz
    holder = None
    if (64 << 20) <= nbytes < (96 << 30) and free > nbytes + (64 << 30) + (8 << 30):
        try:
            holder = capi.PlacedMatrix(shape, cdt)
            out = holder.tensor
            label += f" [placed: {holder.info['window_gbs']:.0f} GB/s store-only, {holder.info['start_gbs']:.0f} at the allocation's start]"
        except Exception:
            holder = None
    if holder is None:
        out = torch.empty(shape, dtype=cdt, device="cuda")
        label += " [plain allocation]"
    p = SDFT(m, window, 1.0, combo, channels)
    for k, v in opts.items():
        p.set_option(k, v)
    p.set_option("profile", 1); p.set_option("async", 1)
    y = None
    for _ in range(2):
        p.sdft(x, out); y = p.isdft(out, y)
    p.synchronize(); p.profile()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.sdft(x, out)
    p.synchronize(); wall_f = (time.perf_counter() - t0) / reps
    for _ in range(16):                                       # (the plan tries its synthesis forms on the host's own calls: past that --
        p.isdft(out, y)                                       # one call at a time: a trial reports when its events have completed)
        p.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.isdft(out, y)
    p.synchronize(); wall_i = (time.perf_counter() - t0) / reps
    pr = p.profile()
    geometry = (f"{p.get_option('last_chunks')}×{p.get_option('last_chunk_len')}"
                f" (kernel {p.get_option('last_kernel')}, {'exact' if p.get_option('carry') else 'fast'} carry"
                f"{', chain form' if p.get_option('last_chain') else ''}, {p.get_option('last_segments')} seg)")
    # fused analysis -> synthesis on the same input (no matrix traffic)
    p.set_option("profile", 0)
    yf = p.process(x)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.process(x, out=yf)
    p.synchronize(); wall_p = (time.perf_counter() - t0) / reps
    path = {1: "fused", 2: "hop pair", 3: "two-pass"}[p.get_option("last_process_path")]
    f = pr["forward"][0] / max(pr["forward"][1], 1)          # one event pair per call, all segments
    c = pr["carry"][0] / max(pr["carry"][1], 1)
    i = pr["inverse"][0] / max(pr["inverse"][1], 1)
    byts = channels * n * (m * esz + x.element_size())
    ROWS.append(f"| {label} | {channels}×{n}×{m} {window} {combo} | {geometry}"
                f" | {wall_f * 1e3:.3f} | {channels * n / wall_f / 1e6:.1f} | {byts / wall_f / 1e12:.2f} | {f:.3f} | {c:.3f}"
                f" | {wall_i * 1e3:.3f} | {channels * n / wall_i / 1e6:.1f} | {byts / wall_i / 1e12:.2f}"
                f" | {wall_p * 1e3:.3f} ({path}) | {channels * n / wall_p / 1e6:.1f} | {(wall_f + wall_i) / wall_p:.2f}× |")
    p.close()
    del out, x
    if holder is not None:
        holder.free()
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("configs[0] shape (n=48000)", 48000, 1024, "hann", "f32f64")
    run("configs[1] n=1e6 forward", 1_000_000, 1024, "hann", "f32f64")
    run("configs[2] m=4096 Blackman FD float round trip", 262144, 4096, "blackman", "f32f32")
    run("configs[3] 64 ch × m=2048 (FD double, 100.7 GB)", 48000, 2048, "hann", "f32f64", channels=64, reps=3)
    run("configs[3] 64 ch × m=2048 (FD float, 50.3 GB)", 48000, 2048, "hann", "f32f32", channels=64, reps=3)
    run("configs[4] one GPU's share: 64 ch × m=1024", 48000, 1024, "hann", "f32f64", channels=64, reps=3)
    run("reference test shape m=1000", 352800, 1000, "hann", "f32f64")
    run("configs[1] with bit-exact carries (carry=1)", 1_000_000, 1024, "hann", "f32f64", carry=1)
    run("configs[2] with the serial pass instead of the chain form", 262144, 4096, "blackman", "f32f32", chain=0)
    run("reference bench shape (cpp/examples/bench.cpp): m=1000, 44100 samples, TD = FD = double", 44100, 1000, "hann", "f64f64")
    run("north-star size with double samples: n=48000, m=1024, f64f64", 48000, 1024, "hann", "f64f64")
    run("TD = FD = double, n=1e6, m=1024", 1_000_000, 1024, "hann", "f64f64")
    run("FD float, n=1e6, m=1024", 1_000_000, 1024, "hann", "f32f32")
    run("FD float, m=1024", 262144, 1024, "hann", "f32f32")
    run("FD float, m=1024, float_carry_parallel=1 (not the float reference's bits)", 262144, 1024, "hann", "f32f32", float_carry_parallel=1)
    run("configs[2] with float_carry_parallel=1 (not the float reference's bits)", 262144, 4096, "blackman", "f32f32", float_carry_parallel=1)
    print(f"# BASELINE config shapes on 1× MI355X ({torch.cuda.get_device_name(0)}), device-resident buffers\n")
    print("forward = sdft_sdft_n (delta + carries + forward kernel, wall per call incl. launches); inverse = sdft_isdft_n (after the 16 calls on which the plan tries its forms).")
    print("TB/s = algorithmic bytes (N·sizeof(fdx) + sizeof(td) per sample) / wall time.")
    print("process = sdft_hip_process_n (identity), the fused analysis→synthesis call; its last column is the speed-up over forward + inverse.\n")
    print("| config | shape | time chunks | fwd ms | fwd Msamples/s | fwd TB/s | fwd-kernel ms | carry ms (on main stream) | inv ms | inv Msamples/s | inv TB/s | process ms | process Msamples/s | vs two calls |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    print("\n".join(ROWS))
