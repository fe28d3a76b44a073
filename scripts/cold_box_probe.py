"""Is a fresh lease slower for its first seconds?  The headline measurement (n = 1e6, m = 1024, asynchronous analysis calls, 10 per
sample) in child processes: at once, again, after 20 s of idling, after 20 s of stores, and again.
    python scripts/cold_box_probe.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, sys.argv[1])
label, busy = sys.argv[2], float(sys.argv[3])
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
from sdft_amd import capi
n, m = 1_000_000, 1024
x = torch.from_numpy(sine_sweep(n)).cuda()
out = torch.empty((n, m), dtype=torch.complex128, device="cuda")
lib = capi.load()
t_busy = time.perf_counter()
while time.perf_counter() - t_busy < busy:
    lib.sdft_hip_store_ceiling(out.data_ptr(), n * m * 16, 4, m, 8, 1960, 20)
p = SDFT(m, "hann", 1.0, "f32f64")
p.set_option("async", 1)
res = []
for rep in range(4):
    for _ in range(2 if rep == 0 else 0): p.sdft(x, out)
    p.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): p.sdft(x, out)
    p.synchronize(); torch.cuda.synchronize()
    res.append(n * (m * 16 + 4) / ((time.perf_counter() - t0) / 10) / 8e12)
grp = lib.sdft_hip_store_ceiling(out.data_ptr(), n * m * 16, 2, m, 8, 1960, 4)
spr = lib.sdft_hip_store_ceiling(out.data_ptr(), n * m * 16, 4, m, 8, 1960, 4)
p.set_option("xcd_map", 0)
p.synchronize(); t0 = time.perf_counter()
for _ in range(10): p.sdft(x, out)
p.synchronize(); nomap = n * (m * 16 + 4) / ((time.perf_counter() - t0) / 10) / 8e12
print(f"{label:44s} " + " ".join(f"{r:5.1%}" for r in res) + f"   xcd_map=0: {nomap:5.1%}   store-only {n * m * 16 / grp / 1e6:5.0f} / {n * m * 16 / spr / 1e6:5.0f} GB/s (b -> chunk b / XCD-contiguous)")
'''
t_start = time.time()
for label, idle, busy in (("first process on the lease", 0, 0), ("again at once", 0, 0), ("after 20 s of idling", 20, 0), ("after 20 s of store kernels", 0, 20), ("again at once", 0, 0)):
    time.sleep(idle)
    q = subprocess.run([sys.executable, "-c", CHILD, ROOT, f"[{time.time() - t_start:5.0f} s] {label}", str(busy)], capture_output=True, text=True, cwd=ROOT)
    lines = [l for l in q.stdout.splitlines() if "%" in l]
    print(lines[-1] if lines else q.stderr[-300:], flush=True)
