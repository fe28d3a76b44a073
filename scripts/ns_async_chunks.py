"""Pipelined calls (two matrices in turn, raw C-ABI): time per call by chunk length at a few call lengths."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m = 1024
big = [torch.empty((262144, m), dtype=torch.complex128, device="cuda") for _ in range(2)]
for n in (24000, 48000, 66000, 100000, 262144):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    line = []
    for chunk in ((0, 64, 96, 128, 160, 192, 224, 256, 288, 320, 384, 512, 0) if n < 200000 else (0, 256, 352, 440, 512, 640, 880, 1024, 0)):
        p = SDFT(m, "hann", 1.0, "f32f64")
        p.set_option("async", 1)
        if chunk:
            p.set_option("chunk", chunk)
        xp = C.c_void_p(x.data_ptr())
        op = [C.c_void_p(big[0].data_ptr()), C.c_void_p(big[1].data_ptr())]
        for i in range(6):
            p.api.sdft_n(p._p, n, xp, op[i & 1])
        p.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(50):
            p.api.sdft_n(p._p, n, xp, op[i & 1])
        p.synchronize(); torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / 50
        line.append(f"{p.get_option('last_chunks')}x{p.get_option('last_chunk_len')}{'*' if not chunk else ''}: {w * 1e6:.1f} us {n * (m * 16 + 4) / w / 8e12:.1%} p{p.get_option('last_pipelined')}")
        p.close()
    print(f"n={n}: " + " | ".join(line), flush=True)
