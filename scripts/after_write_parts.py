"""Right after the analysis wrote a 706 MB matrix: how fast does the synthesis read its tail (the last rows written: in the Infinity Cache?) against its head, with
ordinary and with non-temporal loads?  m = 1000, n = 44100, TD = FD = double, analysis workgroups in time order (xcd_map = 0) or placed by XCD; synchronous calls."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT

m, n, part = 1000, 44100, 11000
print(f"device: {torch.cuda.get_device_name(0)}")
x = torch.randn(n, dtype=torch.float64, device="cuda")
d = torch.empty((n, m), dtype=torch.complex128, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass


for pause in (0, 50, 200, 1000):
    pa = SDFT(m, "hann", 1.0, "f64f64")
    ps = SDFT(m, "hann", 1.0, "f64f64"); ps.set_option("inverse_tune", 0)
    ts = []
    for r in range(14):
        pa.sdft(x, d)
        spin(pause)
        t0 = time.perf_counter(); ps.isdft(d, y); t1 = time.perf_counter()
        if r >= 4:
            ts.append(t1 - t0)
    print(f"whole matrix, {pause:4d} us between the analysis' return and the synthesis call: isdft {np.median(ts) * 1e6:6.1f} us", flush=True)
    pa.close(); ps.close()
ts = []
ps = SDFT(m, "hann", 1.0, "f64f64"); ps.set_option("inverse_tune", 0)
for r in range(14):
    t0 = time.perf_counter(); ps.isdft(d, y); t1 = time.perf_counter()
    if r >= 4:
        ts.append(t1 - t0)
print(f"whole matrix, nothing written in between: isdft {np.median(ts) * 1e6:6.1f} us", flush=True)
ps.close()
for xm in (0,):
    for nt in (1,):
        pa = SDFT(m, "hann", 1.0, "f64f64"); pa.set_option("xcd_map", xm)
        ps = SDFT(m, "hann", 1.0, "f64f64"); ps.set_option("inverse_nt", nt); ps.set_option("inverse_tune", 0)
        res = {}
        for label, lo in (("head", 0), ("second quarter", part), ("third quarter", 2 * part), ("tail", n - part)):
            ts = []
            for r in range(12):
                pa.sdft(x, d)
                torch.cuda.synchronize()
                t0 = time.perf_counter(); ps.isdft(d[lo:lo + part], y[lo:lo + part]); t1 = time.perf_counter()
                if r >= 4:
                    ts.append(t1 - t0)
            res[label] = np.median(ts)
        ts = []
        for r in range(12):
            t0 = time.perf_counter(); ps.isdft(d[:part], y[:part]); t1 = time.perf_counter()
            if r >= 4:
                ts.append(t1 - t0)
        b = part * m * 16
        print(f"xcd_map={xm} inverse_nt={nt}: " + "  ".join(f"{k} {v * 1e6:6.1f} us ({b / v / 1e9:5.0f} GB/s)" for k, v in res.items()) + f"   head again, nothing written {np.median(ts) * 1e6:6.1f} us ({b / np.median(ts) / 1e9:5.0f} GB/s)", flush=True)
        pa.close(); ps.close()
