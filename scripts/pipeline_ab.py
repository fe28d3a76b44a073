"""Asynchronous analysis calls back to back (m = 1024, f32f64): consecutive calls' row kernels on two streams behind a chain
of state kernels (option pipeline = 1, default) against one stream (pipeline = 0): time per call by call length, and the bits."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
outs = [torch.empty((262144, m), dtype=torch.complex128, device="cuda") for _ in range(2)]
for n in (12000, 24000, 36000, 48000, 52000, 66000, 100000, 131072, 262144):
    xs = [torch.from_numpy(sine_sweep(n) * (1.0 + 0.1 * i)).cuda() for i in range(4)]
    res = {}
    for pipe in (0, 1):
        p = SDFT(m, "hann", 1.0, "f32f64")
        p.set_option("async", 1)
        p.set_option("pipeline", pipe)
        for i in range(4):
            p.sdft(xs[i], outs[i & 1][:n])
        p.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            p.sdft(xs[i & 3], outs[i & 1][:n])
        p.synchronize()
        w = (time.perf_counter() - t0) / 40
        # bits: four calls into separate buffers on a fresh stream state
        p.reset()
        got = []
        for i in range(4):
            o = torch.empty((n, m), dtype=torch.complex128, device="cuda")
            p.sdft(xs[i], o)
            got.append(o)
        p.synchronize()
        res[pipe] = (w, [g.clone() for g in got], p.get_option("pipelined_calls"), p.state())
        p.close()
    err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[1][1], res[0][1]))
    serr = float(np.abs(res[1][3][0] - res[0][3][0]).max() / np.abs(res[0][3][0]).max())
    pc = lambda w: f"{w * 1e6:7.1f} us = {n * (m * 16 + 4) / w / 8e12:5.1%}"
    print(f"n={n:6d}  one stream {pc(res[0][0])}   pipelined {pc(res[1][0])}"
          f"   (pipelined calls {res[1][2]}, max deviation {err:.1e}, state {serr:.1e}, cursor {res[1][3][3]} / {res[0][3][3]})", flush=True)
