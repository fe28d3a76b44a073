"""Fused call with operations that are not linear (gate, power: forward_rows_kernel<SYN = 1>) and with the reference's
summation order (SYN = 2): wall time per call.    python scripts/nonlinear_perf.py [n] [m] [label substring: only these variants, f32f64 hann only]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
only = sys.argv[3] if len(sys.argv) > 3 else None
for combo, win in (("f32f64", "hann"), ("f32f64", "blackman"), ("f32f32", "hann"), ("f64f64", "hamming")):
    x = torch.from_numpy(sine_sweep(n, dtype=np.float32 if combo[:3] == "f32" else np.float64)).cuda()
    y = torch.empty_like(x)
    for label, op, kw, opts in (("identity (folded)", "identity", {}, {}), ("identity, windowed rows (fold=0)", "identity", {}, {"fold": 0}),
                                ("gate", "gate", {"threshold": 0.01, "floor": 0.0}, {}), ("power", "power", {"exponent": 0.7, "scale": 1.0}, {}),
                                ("gate as an expression (run-time compiled)", "expr", {"expr": "if (re * re + im * im < p[0] * p[0]) { re *= p[1]; im *= p[1]; }", "expr_params": [0.01, 0.0]}, {}),
                                ("identity, reference order", "identity", {}, {"fused_exact": 2})):
        if combo == "f32f32" and n > 300000: continue
        if only and (only not in label or (combo, win) != ("f32f64", "hann")): continue
        with SDFT(m, win, 1.0, combo) as p:
            for k, v in opts.items(): p.set_option(k, v)
            p.set_option("async", 1)
            p.process(x, op, out=y, **kw); p.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): p.process(x, op, out=y, **kw)
            p.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f"{combo} {win} n={n} m={m} {label}: {dt*1e3:.3f} ms ({n/dt/1e6:.0f} Msamples/s) path={p.get_option('last_process_path')} exact_order={p.get_option('last_fused_exact')} chunks={p.get_option('last_chunks')}", flush=True)
