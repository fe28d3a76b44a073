"""A handful of config-3 analysis calls (m = 4096, Blackman, FD float, n = 262144) for a kernel trace: python3 scripts/config3_calls.py [sync|async]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
n, m = 262144, 4096
x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
p = SDFT(m, "blackman", 1.0, "f32f32")
d = p.sdft(x)
if mode == "async":
    p.set_option("async", 1)
for _ in range(6):
    p.sdft(x, d)
p.synchronize()
p.close()
