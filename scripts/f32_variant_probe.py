"""The bin-pair analysis kernel (m = 4096, Blackman, FD float, n = 262144) of several builds of the library side by side, each in a
child process (SDFT_HIP_LIBRARY), in turn on one lease: alone on the chip with chunk-parallel carries (not bit-identical:
measurement only), beside 128 and 192 held CUs, and the product's call (exact carries, flow mode beside the relay).
    python scripts/f32_variant_probe.py label=path [label=path ...] [rounds=2]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
lib = capi.load()
m, window, n = 4096, "blackman", 262144
x = torch.from_numpy(sine_sweep(n)).cuda()
d = torch.empty((n, m), dtype=torch.complex64, device="cuda")
b = n * (m * 8 + 4)
out = {}
def timed(p, reps):
    p.synchronize()                              # (not the device: a hold kernel may be running)
    t0 = time.perf_counter()
    for _ in range(reps): p.sdft(x, d)
    p.synchronize()
    return (time.perf_counter() - t0) / reps
p = SDFT(m, window, 1.0, "f32f32")
p.set_option("async", 1)
for _ in range(2): p.sdft(x, d)
out["flow_ms"] = min(timed(p, 5) for _ in range(3)) * 1e3
p.close()
p = SDFT(m, window, 1.0, "f32f32")
p.set_option("async", 1); p.set_option("float_carry_parallel", 1)
for _ in range(2): p.sdft(x, d)
out["alone_ms"] = min(timed(p, 5) for _ in range(3)) * 1e3
for held in (128, 192):
    lib.sdft_hip_hold_cus(held, 400.0)
    time.sleep(0.01)
    w = timed(p, 3)
    lib.sdft_hip_hold_cus(0, 0.0)
    time.sleep(0.01)
    out[f"held{held}_gbs_per_cu"] = b / w / 1e9 / (256 - held)
p.close()
print(json.dumps(out))
'''
variants, rounds = [], 2
for a in sys.argv[1:]:
    k, v = a.split("=", 1)
    if k == "rounds": rounds = int(v)
    else: variants.append((k, v if os.path.isabs(v) else os.path.join(ROOT, v)))
for r in range(rounds):
    for label, path in variants:
        env = dict(os.environ, SDFT_HIP_LIBRARY=path)
        q = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, cwd=ROOT, env=env)
        try:
            o = json.loads(q.stdout.strip().splitlines()[-1])
            print(f"{label:28s} flow {o['flow_ms']:.3f} ms   alone {o['alone_ms']:.3f} ms   per free CU at 128 / 192 held: {o['held128_gbs_per_cu']:.1f} / {o['held192_gbs_per_cu']:.1f} GB/s", flush=True)
        except Exception:
            print(label, "failed:", q.stdout[-500:], q.stderr[-1500:], flush=True)
