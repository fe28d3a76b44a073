"""FD float with chunk-parallel carries (option float_carry_parallel = 1): deviation from the float reference and wall time.
    python scripts/float_fast_probe.py            (GPU box; oracle libraries built)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep, noise
from oracle import oracle as O

def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
def rel_rows(a, b):
    # worst row: the deviation of a row against that row's own largest bin
    return float(np.max(np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), 1e-30)))

for m, win, n, sig in ((1024, "hann", 262144, "sweep"), (1024, "hann", 262144, "noise"), (4096, "blackman", 65536, "sweep"), (1000, "hamming", 100000, "noise")):
    x = sine_sweep(n) if sig == "sweep" else noise(n, seed=5)
    ref = O.best(m, win, 1.0, "f32f32")
    want = ref.sdft(x)
    ref64 = O.best(m, win, 1.0, "f32f64")
    truth = ref64.sdft(x)
    print(f"m={m} {win} n={n} {sig}: float reference against the double reference: {rel(want, truth):.2e} (max-normalised) {rel_rows(want, truth):.2e} (worst row)", flush=True)
    for carry in (1, 0):
        with SDFT(m, win, 1.0, "f32f32") as p:
            p.set_option("float_carry_parallel", 1 - carry)
            got = p.sdft(x)
            xd = torch.from_numpy(x).cuda(); out = torch.empty((n, m), dtype=torch.complex64, device="cuda")
            p.set_option("async", 1)
            p.sdft(xd, out); p.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): p.sdft(xd, out)
            p.synchronize(); dt = (time.perf_counter() - t0) / 10
            y = p.isdft(got)
            p.set_option("async", 0)
            xs = x[:50000]; p.reset(); ref.reset(); ref64.reset()
            d = ref.sdft(xs); yy = p.process(xs, "identity"); yw = ref.isdft(d); yt = ref64.isdft(ref64.sdft(xs))
            print(f"   fused call, 50000 samples: against the float reference's round trip {rel(yy, yw):.2e}, against the double one {rel(yy, yt):.2e} (float reference: {rel(yw, yt):.2e})")
            print(f"   carry={carry}: against the float reference {rel(got, want):.2e} / {rel_rows(got, want):.2e}, against the double reference {rel(got, truth):.2e}; "
                  f"chunks={p.get_option('last_chunks')} {dt*1e3:.3f} ms per call ({n*m*8/dt/1e12:.2f} TB/s); isdft against the reference's {rel(y, ref.isdft(want)):.2e}", flush=True)
