"""A/B of the FD float analysis kernels inside one process: the bin-pair kernel (option rows_f32 = 1, default) against the
generic row-group kernel (rows_f32 = 0): bits must be equal; wall time per call, carry and forward stage times."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

HBM = 8000.0


def run(m, window, n, combo="f32f32", reps=8, **opts):
    x = torch.from_numpy(sine_sweep(n, dtype=np.float32 if combo[:3] == "f32" else np.float64)).cuda()
    outs, line = [], []
    for flavour in (0, 1):
        p = SDFT(m, window, 1.0, combo)
        for k, v in opts.items():
            p.set_option(k, v)
        p.set_option("rows_f32", flavour)
        d = p.sdft(x)
        for _ in range(2):
            p.reset(); p.sdft(x, d)
        p.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            p.sdft(x, d)
        p.synchronize()
        wall = (time.perf_counter() - t0) / reps
        p.set_option("async", 1); p.set_option("profile", 1)
        for _ in range(reps):
            p.sdft(x, d)
        pr = p.profile()
        p.set_option("profile", 0); p.set_option("async", 0)
        p.reset()
        outs.append(p.sdft(x).clone())
        b = n * (m * 8 + 4)
        line.append(f"rows_f32={flavour} used={p.get_option('last_rows_f32')} wall {wall * 1e3:7.3f} ms = {b / wall / 1e9:6.0f} GB/s ({b / wall / 1e9 / HBM:5.1%})"
                    f"  carry {(pr['delta'][0] + pr['carry'][0]) / max(pr['forward'][1], 1):6.3f} ms  fwd-kernel {pr['forward'][0] / max(pr['forward'][1], 1):6.3f} ms")
        p.close()
    same = torch.equal(outs[0].view(torch.float32), outs[1].view(torch.float32))
    print(f"m={m} {window} n={n} {combo} {opts}: bits {'EQUAL' if same else 'DIFFER'}")
    for l in line:
        print("   ", l)
    return same


if __name__ == "__main__":
    ok = True
    ok &= run(4096, "blackman", 262144)
    ok &= run(1024, "hann", 262144)
    ok &= run(2048, "hamming", 131072)
    ok &= run(128, "boxcar", 100000)
    ok &= run(256, "blackman", 50000)
    ok &= run(1024, "blackman", 200000, "f64f32")
    ok &= run(4096, "hann", 131072, float_carry_parallel=1)
    ok &= run(1024, "hann", 262144, float_carry_parallel=1)
    print("ALL EQUAL" if ok else "MISMATCH")
