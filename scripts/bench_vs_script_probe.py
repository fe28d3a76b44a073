"""Why does the headline workload run slower inside bench.py than in a script on the same box?  The same measurement (n = 1e6,
m = 1024, asynchronous analysis calls) under bench.py's circumstances, one at a time, each in a child process.
    python scripts/bench_vs_script_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, os
sys.path.insert(0, sys.argv[1])
mode = sys.argv[2]
import numpy as np, torch
if "dist" in mode:
    import torch.distributed as dist
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
from sdft_amd import capi
n, m = 1_000_000, 1024
torch.cuda.set_device(0)
x = torch.from_numpy(sine_sweep(n)).cuda()
out = torch.empty((n, m), dtype=torch.complex128, device="cuda")
p = SDFT(m, "hann", 1.0, "f32f64", device=0)
if "stream" in mode:
    s = torch.cuda.Stream(); p.set_stream(s.cuda_stream)
p.set_option("async", 1)
if "profile" in mode:
    p.set_option("profile", 2)
for _ in range(3): p.sdft(x, out)
p.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): p.sdft(x, out)
p.synchronize(); torch.cuda.synchronize()
w = (time.perf_counter() - t0) / 10
lib = capi.load()
grp = lib.sdft_hip_store_ceiling(out.data_ptr(), n * m * 16, 2, m, 8, 1960, 4)
spr = lib.sdft_hip_store_ceiling(out.data_ptr(), n * m * 16, 4, m, 8, 1960, 4)
print(f"{mode:28s} {w * 1e3:7.3f} ms = {n * (m * 16 + 4) / w / 8e12:5.1%} of peak; store-only {n * m * 16 / grp / 1e6:6.0f} / {n * m * 16 / spr / 1e6:6.0f} GB/s; out at 0x{out.data_ptr():x}")
'''
for mode in ("plain", "dist", "stream", "profile", "dist+stream+profile", "plain"):
    q = subprocess.run([sys.executable, "-c", CHILD, ROOT, mode], capture_output=True, text=True, cwd=ROOT)
    lines = [l for l in q.stdout.splitlines() if "of peak" in l]
    print(lines[-1] if lines else q.stderr[-300:])
