"""Matrices beyond 4 GiB by the host's pattern -- (A) the matrix is only read: synthesis call after synthesis call; (B) round trips: every
synthesis follows the analysis that wrote the matrix -- and by the kind of load: inverse_nt = 0 ordinary, 1 non-temporal, -1 the default
(by size, and the plan's tuners -- one per pattern -- try the other kind on the host's own calls).  Synchronous calls, ms per call.
    python scripts/roundtrip_nt_big.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep, sweep_batch


def run(combo, m, n, ch, label, **opts):
    td = torch.float64 if combo[:3] == "f64" else torch.float32
    cdt = torch.complex128 if combo[3:] == "f64" else torch.complex64
    x = torch.from_numpy(sweep_batch(ch, n) if ch > 1 else sine_sweep(n)).to(td).cuda()
    d = torch.empty((ch, n, m) if ch > 1 else (n, m), dtype=cdt, device="cuda")
    y = torch.empty((ch, n) if ch > 1 else n, dtype=td, device="cuda")
    p = SDFT(m, "hann", 1.0, combo, channels=ch)
    for k, v in opts.items():
        p.set_option(k, v)
    p.sdft(x, d)
    ro = []
    for r in range(26):                                     # (A) only read
        t0 = time.perf_counter(); p.isdft(d, y); t1 = time.perf_counter()
        if r >= 16:
            ro.append(t1 - t0)
    a_state = (p.get_option("last_inverse_tuned"), f"{p.get_option('last_inverse_nt')}/{p.get_option('last_inverse_skip')}")
    fw, iv = [], []
    for r in range(26):                                     # (B) round trips
        t0 = time.perf_counter(); p.sdft(x, d); t1 = time.perf_counter(); p.isdft(d, y); t2 = time.perf_counter()
        if r >= 16:
            fw.append(t1 - t0); iv.append(t2 - t1)
    b_state = (p.get_option("last_inverse_tuned"), f"{p.get_option('last_inverse_nt')}/{p.get_option('last_inverse_skip')}")
    esz = 16 if combo[3:] == "f64" else 8
    b = ch * n * m * esz
    print(f"{combo} {ch} x {n} x {m} {label:24s}: only read {np.median(ro) * 1e3:7.3f} ms ({b / np.median(ro) / 1e9:5.0f} GB/s, form {a_state[0]} nt {a_state[1]})"
          f"   round trip: sdft {np.median(fw) * 1e3:7.3f} + isdft {np.median(iv) * 1e3:7.3f} ms ({b / np.median(iv) / 1e9:5.0f} GB/s, form {b_state[0]} nt {b_state[1]})", flush=True)
    p.close()


if __name__ == "__main__":
    print(f"device: {torch.cuda.get_device_name(0)}")
    shapes = (("f32f64", 1024, 1000000, 1), ("f64f64", 1024, 1000000, 1), ("f64f64", 1024, 500000, 1), ("f32f64", 1024, 400000, 1), ("f32f64", 1024, 48000, 64), ("f32f64", 2048, 48000, 64),
              ("f64f64", 1000, 44100, 1), ("f32f64", 1024, 131072, 1), ("f64f64", 1024, 262144, 1), ("f32f64", 1024, 262144, 1))
    if len(sys.argv) > 1:
        shapes = tuple(shapes[int(i)] for i in sys.argv[1].split(","))
    for combo, m, n, ch in shapes:
        for rep in range(2):
            run(combo, m, n, ch, "ordinary loads", inverse_nt=0)
            run(combo, m, n, ch, "non-temporal loads", inverse_nt=1, inverse_nt_skip_mb=0)
            run(combo, m, n, ch, "default")
