"""Development probe (round 6): configs[2] (m = 4096, Blackman, FD float, n = 262144, bit-identical) by the geometry of the exact-carry relay -- the
default (one relay of 8 waves per workgroup, blocks of 128 steps: 128 CUs), and the relay on 64 CUs: two relays per workgroup with 6 waves each (blocks
of 128: 12 waves per CU, round 4's form) or 8 waves each (blocks of 64 steps, 16 waves per CU = 4 per SIMD: the form the round-5 review asked for).
Interleaved in one process, three rounds; wall per synchronous call, the stages' own times (HIP events, asynchronous calls), bits against the default."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
n, m = 262144, 4096
x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
pm = capi.PlacedMatrix((n, m), torch.complex64)
d = pm.tensor
variants = [("default: 1 relay x 8 waves, L = 128 (128 CUs)", {}),
            ("1 relay x 8 waves, L = 64 (128 CUs)", {"chain_block": 64}),
            ("2 relays x 6 waves, L = 128 (64 CUs, 12 waves per CU)", {"relay_groups": 2, "relay_waves": 6}),
            ("2 relays x 8 waves, L = 64 (64 CUs, 16 waves per CU)", {"chain_block": 64, "relay_groups": 2, "relay_waves": 8}),
            ("2 relays x 6 waves, L = 64 (64 CUs)", {"chain_block": 64, "relay_groups": 2, "relay_waves": 6}),
            ("2 relays x 8 waves, L = 32 (64 CUs)", {"chain_block": 32, "relay_groups": 2, "relay_waves": 8}),
            ("2 relays x 4 waves, L = 64 (64 CUs)", {"chain_block": 64, "relay_groups": 2, "relay_waves": 4})]
b = n * (m * 8 + 4)
ref_bits = None
for rnd in range(3):
    for name, opts in variants:
        p = SDFT(m, "blackman", 1.0, "f32f32")
        for k, v in opts.items():
            p.set_option(k, v)
        p.sdft(x, d); p.synchronize()
        if rnd == 0:
            sample = d[::4099].cpu().numpy().copy()
            st = p.state()
            if ref_bits is None:
                ref_bits = (sample, st[0].copy())
            same = np.array_equal(sample, ref_bits[0]) and np.array_equal(st[0], ref_bits[1])
        t0 = time.perf_counter()
        for _ in range(5):
            p.sdft(x, d)
        p.synchronize(); wall = (time.perf_counter() - t0) / 5
        p.set_option("async", 1); p.set_option("profile", 1)
        for _ in range(5):
            p.sdft(x, d)
        pr = p.profile()
        calls = max(pr["forward"][1], 1)
        print(f"round {rnd}  {name:58s} wall {wall * 1e3:6.3f} ms = {b / wall / 8e12:.4f}   relay {(pr['delta'][0] + pr['carry'][0]) / calls:6.3f} ms   forward launch {pr['forward'][0] / calls:6.3f} ms"
              f"   chain form {p.get_option('last_chain')} flow {p.get_option('last_flow')}" + (f"   bits as the default: {same}" if rnd == 0 else ""), flush=True)
        p.close()
pm.free()
