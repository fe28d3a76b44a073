import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
m = 1024
xs = sine_sweep(4000, dtype=np.float32)
for resident in (0, 1):
    p = SDFT(m, "hann", 1.0, "f32f64"); p.set_option("resident", resident)
    row = torch.empty(m, dtype=torch.complex128, device="cuda"); rp = C.c_void_p(row.data_ptr())
    for i in range(200): p.api.sdft(p._p, float(xs[i]), rp); p.api.isdft(p._p, rp)
    t0 = time.perf_counter()
    for i in range(200, 2200): p.api.sdft(p._p, float(xs[i]), rp)
    t1 = time.perf_counter()
    for i in range(2000): p.api.isdft(p._p, rp)
    t2 = time.perf_counter()
    for i in range(200, 2200): p.api.sdft(p._p, float(xs[i]), rp); p.api.isdft(p._p, rp)
    t3 = time.perf_counter()
    print(f"resident={resident}: sdft {(t1-t0)/2000*1e6:.2f} us  isdft {(t2-t1)/2000*1e6:.2f} us  pair {(t3-t2)/2000*1e6:.2f} us  resident_calls {p.get_option('resident_calls')}")
    p.close()
