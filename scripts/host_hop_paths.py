"""The reference driver's loop on malloc'ed buffers (test/test.c:62-83: m = 1000, hops of 100): time of the two calls of a
hop, each by itself, by the way host memory travels (option host_copy 0 / 1, host_register)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m, hop, total = 1000, 100, 20000
xh = sine_sweep(total)
yh = np.zeros(total, dtype=np.float32)
dh = np.zeros((hop, m), dtype=np.complex128)
for label, reg, rt, direct, threads in (("pinned slots, kernels on them (default: 2 copy threads)", 0, 0, 1, 2), ("... one thread (round 4)", 0, 0, 1, 0), ("... 3 copy threads", 0, 0, 1, 3),
                                        ("pinned slots, DMA", 0, 0, 0, 2), ("runtime's pageable path", 0, 1, 0, 0),
                                        ("registered in place", 1, 0, 0, 0), ("pinned slots, kernels on them (again)", 0, 0, 1, 2), ("... one thread (again)", 0, 0, 1, 0)):
    p = SDFT(m, "hann", 1.0, "f32f64")
    p.set_option("copy_threads", threads)
    p.set_option("host_register", reg)
    p.set_option("host_copy", rt)
    p.set_option("host_direct", direct)
    ta = ts = 0.0
    for rep in range(2):
        ta = ts = 0.0
        for i in range(0, total, hop):
            t0 = time.perf_counter()
            p.api.sdft_n(p._p, hop, C.c_void_p(xh.ctypes.data + i * 4), C.c_void_p(dh.ctypes.data))
            t1 = time.perf_counter()
            p.api.isdft_n(p._p, hop, C.c_void_p(dh.ctypes.data), C.c_void_p(yh.ctypes.data + i * 4))
            t2 = time.perf_counter()
            ta += t1 - t0; ts += t2 - t1
    k = total // hop
    print(f"{label:58s} sdft_n {ta / k * 1e6:7.1f} us   isdft_n {ts / k * 1e6:7.1f} us   per hop {(ta + ts) / k * 1e6:7.1f} us")
    if p.get_option("host_copies_staged"):
        c = p.get_option("host_copies_staged")
        print(f"{'':58s} per staged copy: host memcpy {p.get_option('host_copy_memcpy_us') / c:6.1f} us, device {p.get_option('host_copy_device_us') / c:6.1f} us")
    p.close()
