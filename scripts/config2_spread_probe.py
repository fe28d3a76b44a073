"""configs[2] analysis (m = 4096, Blackman, FD float, n = 262144): does the bin-pair kernel gain from workgroups spread over
the matrix (xcd_map) and from the buffer?  Exact carries (flow mode beside the relay: time-major order) against
chunk-parallel carries (not bit-identical: measurement only) with xcd_map 0 / 1, on several buffers of one process.
    python scripts/config2_spread_probe.py [buffers]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m, n = 4096, 262144
print(f"device: {torch.cuda.get_device_name(0)}")
x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
bufs = [torch.empty((n, m), dtype=torch.complex64, device="cuda") for _ in range(nbuf)]
V = (("exact carries (flow mode)", {}), ("parallel carries, xcd_map=0", {"float_carry_parallel": 1, "xcd_map": 0}),
     ("parallel carries, xcd_map=1", {"float_carry_parallel": 1, "xcd_map": 1}))
plans = []
for vl, opts in V:
    p = SDFT(m, "blackman", 1.0, "f32f32")
    p.set_option("async", 1)
    for k, v in opts.items():
        p.set_option(k, v)
    plans.append((vl, p))
b = n * (m * 8 + 4)
for bi, d in enumerate(bufs):
    for vl, p in plans:
        for _ in range(2):
            p.sdft(x, d)
        p.synchronize()
    for vl, p in plans:
        ws = []
        for r in range(3):
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                p.sdft(x, d)
            p.synchronize()
            ws.append((time.perf_counter() - t0) / 6)
        w = float(np.median(ws))
        print(f"buffer {bi} ({d.data_ptr():#x})  {vl:30s} {w * 1e3:7.3f} ms = {b / w / 1e9:6.0f} GB/s = {b / w / 8e12:5.1%}   chunks {p.get_option('last_chunks')} x {p.get_option('last_chunk_len')}")
