"""Streams 200 hops of 100 samples through the fused call and through the two reference calls (device pointers,
asynchronous): the target of kernel-trace passes for the hop kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m, hop, total = 1000, 100, 20000
x = torch.from_numpy(sine_sweep(total)).cuda()
y = torch.empty_like(x)
d = torch.empty((hop, m), dtype=torch.complex128, device="cuda")
p = SDFT(m, "hann", 1.0, "f32f64")
p.set_option("async", 1)
for i in range(0, total, hop):
    p.process(x[i:i + hop], out=y[i:i + hop])
p.synchronize()
for i in range(0, total, hop):
    p.sdft(x[i:i + hop], d); p.isdft(d, y[i:i + hop])
p.synchronize()
p.close()
