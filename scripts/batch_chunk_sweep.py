"""One GPU's share of configs[4] (64 channels x 48000 samples, m = 1024, FD double): analysis by chunk length, carry form
(self-carried chunks / pre-pass) and workgroup placement; asynchronous calls, interleaved rounds.
    python scripts/batch_chunk_sweep.py [rounds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sweep_batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ch, n, m = 64, 48000, 1024
x = torch.from_numpy(sweep_batch(ch, n)).cuda()
d = torch.empty((ch, n, m), dtype=torch.complex128, device="cuda")
variants = [("default", {}), ("xcd_map=0", {"xcd_map": 0}), ("self_carry=0", {"self_carry": 0}), ("self_carry=0 xcd_map=0", {"self_carry": 0, "xcd_map": 0})]
for c in (1504, 3000, 12000, 24000, 48000):
    variants.append((f"chunk={c}", {"chunk": c}))
    variants.append((f"chunk={c} self_carry=0", {"chunk": c, "self_carry": 0}))
plans = []
for label, opts in variants:
    p = SDFT(m, "hann", 1.0, "f32f64", channels=ch)
    p.set_option("async", 1)
    for k, v in opts.items():
        p.set_option(k, v)
    for _ in range(2):
        p.sdft(x, d)
    p.synchronize()
    plans.append((label, p))
res = {label: [] for label, _ in plans}
for r in range(rounds):
    for label, p in plans:
        p.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            p.sdft(x, d)
        p.synchronize()
        res[label].append((time.perf_counter() - t0) / 3)
b = ch * n * (m * 16 + 4)
print(f"device: {torch.cuda.get_device_name(0)}")
for label, p in plans:
    w = float(np.median(res[label]))
    print(f"{label:30s} {w * 1e3:8.3f} ms = {b / w / 1e9:6.0f} GB/s = {b / w / 8e12:5.1%} of peak  chunks {p.get_option('last_chunks')} x {p.get_option('last_chunk_len')} self {p.get_option('last_self')}")
    p.close()
