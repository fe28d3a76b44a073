"""Development probe: A/B of the north star's shape (n = 48000, N = 1024, f32f64) inside one process -- leases differ.  Round 6: into a
matrix the library placed (sdft_hip_malloc_matrix_in_arena) and into a plain allocation; chunkings of one, 1.5 and two rounds of the chip."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
n, m = 48000, 1024
x = torch.from_numpy(sine_sweep(n)).cuda()
plain = torch.empty((n, m), dtype=torch.complex128, device="cuda")
placed = capi.PlacedMatrix((n, m), torch.complex128)
print("placed matrix:", placed.info, flush=True)
variants = [("default", {}), ("chunk=192 (250 chunks)", {"chunk": 192}), ("chunk=96 (500)", {"chunk": 96}), ("chunk=376 (128)", {"chunk": 376}),
            ("chunk=128 (375)", {"chunk": 128}), ("chunk=64 (750)", {"chunk": 64}), ("pre-pass", {"self_carry": 0})]
b = n * (m * 16 + 4)
for rnd in range(2):
    for where, out in (("placed", placed.tensor), ("plain ", plain)):
        for name, opts in variants:
            p = SDFT(m, "hann", 1.0, "f32f64")
            for k, v in opts.items(): p.set_option(k, v)
            res = {}
            for mode in ("async", "sync"):
                p.set_option("async", 1 if mode == "async" else 0)
                for _ in range(5): p.sdft(x, out)
                p.synchronize()
                t0 = time.perf_counter()
                for _ in range(200): p.sdft(x, out)
                p.synchronize(); res[mode] = (time.perf_counter() - t0) / 200
            print(f"round {rnd} {where} {name:26s} chunks={p.get_option('last_chunks')} len={p.get_option('last_chunk_len')} self={p.get_option('last_self')}: "
                  f"async {res['async']*1e6:.1f} us ({b/res['async']/8e12:.3f}) sync {res['sync']*1e6:.1f} us ({b/res['sync']/8e12:.3f})", flush=True)
            p.close()
placed.free()
