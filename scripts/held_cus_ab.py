"""What the analysis kernels keep of their speed when another kernel holds part of the chip (sdft_hip_hold_cus): the FD
float kernels without a relay beside them (option float_carry_parallel = 1), the FD double kernel for comparison."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

lib = capi.load()


def run(m, window, n, combo, held, **opts):
    x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
    p = SDFT(m, window, 1.0, combo)
    for k, v in opts.items():
        p.set_option(k, v)
    d = p.sdft(x)
    p.set_option("async", 1)
    for _ in range(3):
        p.sdft(x, d)
    p.synchronize()
    if held:
        lib.sdft_hip_hold_cus(held, 60.0)
        time.sleep(0.005)
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        p.sdft(x, d)
    p.synchronize()
    w = (time.perf_counter() - t0) / reps
    lib.sdft_hip_hold_cus(0, 0.0)
    esz = 16 if combo[3:] == "f64" else 8
    b = n * (m * esz + 4)
    p.close()
    return b / w / 1e9


if __name__ == "__main__":
    for label, m, window, n, combo, opts in (
            ("f32 half-row workgroups, m=4096 blackman", 4096, "blackman", 131072, "f32f32", {"float_carry_parallel": 1, "rows_f32": 1, "rows_split": 1}),
            ("f32 bin-pair kernel, m=4096 blackman", 4096, "blackman", 131072, "f32f32", {"float_carry_parallel": 1, "rows_f32": 1, "rows_split": 0}),
            ("f32 generic kernel,  m=4096 blackman", 4096, "blackman", 131072, "f32f32", {"float_carry_parallel": 1, "rows_f32": 0}),
            ("f32 bin-pair kernel, m=1024 hann", 1024, "hann", 262144, "f32f32", {"float_carry_parallel": 1, "rows_f32": 1}),
            ("f32 generic kernel,  m=1024 hann", 1024, "hann", 262144, "f32f32", {"float_carry_parallel": 1, "rows_f32": 0}),
            ("f64 kernel (pre-pass carries), m=1024 hann", 1024, "hann", 262144, "f32f64", {"self_carry": 0})):
        rates = [run(m, window, n, combo, held, **opts) for held in (0, 64, 128, 192)]
        print(f"{label:46s} CUs held 0 / 64 / 128 / 192: " + " / ".join(f"{r:5.0f}" for r in rates) + " GB/s"
              + f"   per free CU at 128 held: {rates[2] / 128:5.1f} GB/s")
