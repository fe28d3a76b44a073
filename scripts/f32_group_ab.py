"""Development probe (round 6): the FD float bin-pair kernel with 4 (product) or 2 rows per lockstep group -- a group's 128 KiB of stores may exceed what a CU
can have in flight, so that arithmetic and stores overlap only in part.  configs[2] (m = 4096, Blackman, n = 262144): the exact call (relay beside the forward
launch) and the kernel alone (chunk-parallel carries: timing only); m = 2048 (one slot) too.  Interleaved, three rounds; bits compared."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
for m, n in ((4096, 262144), (2048, 262144)):
    x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
    pm = capi.PlacedMatrix((n, m), torch.complex64)
    d = pm.tensor
    b = n * (m * 8 + 4)
    ref = {}
    for rnd in range(3):
        for alone in (0, 1):
            for G in ((4, 2, 8) if m == 2048 else (4, 2)):
                p = SDFT(m, "blackman", 1.0, "f32f32")
                p.set_option("rows_f32_group", G)
                if alone:
                    p.set_option("float_carry_parallel", 1)
                p.sdft(x, d); p.synchronize()
                if rnd == 0:
                    s = d[::4099].cpu().numpy().copy()
                    ref.setdefault(alone, s)
                    same = np.array_equal(s, ref[alone])
                p.set_option("async", 1)
                t0 = time.perf_counter()
                for _ in range(8):
                    p.sdft(x, d)
                p.synchronize(); wall = (time.perf_counter() - t0) / 8
                print(f"m={m} round {rnd} {'kernel alone (parallel carries)' if alone else 'exact call (relay + forward)  '} rows per group {G}: {wall * 1e3:6.3f} ms = {b / wall / 8e12:.4f}"
                      + (f"   same bits as 4 rows per group: {same}" if rnd == 0 else ""), flush=True)
                p.close()
    del d; pm.free()
