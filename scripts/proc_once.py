"""Runs the fused call (folded form) a few times at the headline shape: the target of counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

n, m = 1000000, 1024
x = torch.from_numpy(sine_sweep(n)).cuda()
y = torch.empty_like(x)
p = SDFT(m, "hann", 1.0, "f32f64")
p.set_option("async", 1)
for _ in range(6):
    p.process(x, "identity", out=y)
p.synchronize()
p.close()
