import sys, os, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
r = bench.hop_streaming(torch, np, SDFT, sine_sweep, "f32f64", np.float32, torch.complex128, 0)
print(r)
print(bench.cpu_hop_baseline(np, sine_sweep, "f32f64", np.float32))
# kernel times of the mapped host path
m, hop, total = 1000, 100, 20000
xh = sine_sweep(total); yh = np.zeros(total, np.float32); dh = np.zeros((hop, m), np.complex128)
p = SDFT(m, "hann", 1.0, "f32f64"); p.set_option("profile", 2); p.set_option("host_register", 1)
for rep in range(2):
    for i in range(0, total, hop):
        p.api.sdft_n(p._p, hop, C.c_void_p(xh.ctypes.data + i * 4), C.c_void_p(dh.ctypes.data))
        p.api.isdft_n(p._p, hop, C.c_void_p(dh.ctypes.data), C.c_void_p(yh.ctypes.data + i * 4))
    pr = p.profile()
print("mapped host buffers, kernel us: forward", round(pr["forward"][0] / pr["forward"][1] * 1e3, 1), "inverse", round(pr["inverse"][0] / pr["inverse"][1] * 1e3, 1),
      "hits", p.get_option("host_register_hits"), "misses", p.get_option("host_register_misses"))
p.close()
