"""The relay kernel alone (no forward launch beside it: flow mode off, one segment) against beside the forward launch."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

for m, window, n in ((1024, "hann", 262144), (4096, "blackman", 262144), (256, "hann", 262144), (512, "hann", 262144), (2048, "hann", 262144), (1024, "hann", 48000)):
    x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
    for label, opts in (("flow mode (beside the forward launch)", {}), ("alone (relay_flow = 0, one segment)", {"relay_flow": 0, "segments": 1}),
                        ):
        p = SDFT(m, window, 1.0, "f32f32")
        for k, v in opts.items():
            p.set_option(k, v)
        d = p.sdft(x)
        p.set_option("async", 1); p.set_option("profile", 1)
        for _ in range(3):
            p.sdft(x, d)
        p.profile()
        for _ in range(6):
            p.sdft(x, d)
        pr = p.profile()
        c = pr["carry"][0] / 6
        p.set_option("profile", 0); p.set_option("async", 0)
        p.sdft(x, d)
        t0 = time.perf_counter()
        for _ in range(6):
            p.sdft(x, d)
        wall = (time.perf_counter() - t0) / 6
        print(f"m={m} n={n} wall {wall * 1e3:6.3f} ms {label:56s}: carry stage {c:6.3f} ms = {c * 1e6 / n:5.2f} ns per step, forward {pr['forward'][0] / 6:6.3f} ms")
        p.close()
