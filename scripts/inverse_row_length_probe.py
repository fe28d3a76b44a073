"""Development probe (round 6): does the exact-order synthesis (tiles of rows x 256-byte pieces) read slower where the row length is a power of two?
sdft_isdft_n on device pointers, TD double / FD double and FD float, by dftsize around the configs' sizes; the matrix is whatever an analysis left there."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

cases = [("f64f64", 1000000, (1000, 1024, 1040)), ("f32f64", 1000000, (1000, 1024, 1040)), ("f32f32", 262144, (4000, 4096, 4160)), ("f32f32", 1000000, (1000, 1024, 1040))]
for combo, n, sizes in cases:
    td = torch.float64 if combo.startswith("f64") else torch.float32
    cd = torch.complex64 if combo.endswith("f32") else torch.complex128
    x = torch.from_numpy(sine_sweep(n).astype(np.float64 if td == torch.float64 else np.float32)).cuda()
    for rnd in range(2):
        for m in sizes:
            p = SDFT(m, "hann", 1.0, combo)
            if combo == "f32f32": p.set_option("float_carry_parallel", 1)      # (only to fill the matrix quickly)
            out = torch.empty((n, m), dtype=cd, device="cuda")
            y = torch.empty((n,), dtype=td, device="cuda")
            p.sdft(x, out)
            for _ in range(10): p.isdft(out, y)
            p.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): p.isdft(out, y)
            p.synchronize(); dt = (time.perf_counter() - t0) / 10
            b = n * (m * out.element_size() + y.element_size())
            print(f"{combo} n={n} m={m:5d} round {rnd}: {dt*1e3:7.3f} ms  {b/dt/1e9:7.0f} GB/s  tuned {p.get_option('last_inverse_tuned')} nt {p.get_option('last_inverse_nt')}", flush=True)
            p.close(); del out, y
            torch.cuda.empty_cache()
