"""Config 3 (m = 4096, Blackman, FD float, n = 262144) analysis by relay geometry: waves per relay, relays per workgroup."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep


def run(m, window, n, **opts):
    x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
    p = SDFT(m, window, 1.0, "f32f32")
    for k, v in opts.items():
        p.set_option(k, v)
    d = p.sdft(x)
    for _ in range(2):
        p.sdft(x, d)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        p.sdft(x, d)
    p.synchronize()
    wall = (time.perf_counter() - t0) / 8
    p.set_option("async", 1); p.set_option("profile", 1)
    for _ in range(8):
        p.sdft(x, d)
    pr = p.profile()
    b = n * (m * 8 + 4)
    print(f"m={m} n={n} {opts}: wall {wall * 1e3:6.3f} ms = {b / wall / 8e12:5.1%}  carry {(pr['delta'][0] + pr['carry'][0]) / 8:6.3f}  forward {pr['forward'][0] / 8:6.3f}  chunks {p.get_option('last_chunks')} x {p.get_option('last_chunk_len')}")
    p.close()


if __name__ == "__main__":
    for m, window, n in ((4096, "blackman", 262144), (2048, "hann", 262144), (1024, "hann", 262144)):
        run(m, window, n)
        for waves, groups in ((6, 2), (4, 2), (4, 1), (6, 1)):
            run(m, window, n, relay_waves=waves)
        run(m, window, n, chunk=256)
        run(m, window, n, chunk=512)
