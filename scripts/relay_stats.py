"""Development probe: where the waves of carry_relay_kernel spend their cycles (option chain_debug = 64)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
for combo, n, m, win, opts in (("f32f32", 262144, 1024, "hann", {}), ("f32f32", 262144, 4096, "blackman", {}), ("f32f64", 262144, 1024, "hann", {"carry": 1})):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    for waves in (4, 6, 8):
        for seg in (1, 8):
            p = SDFT(m, win, 1.0, combo)
            for k, v in dict(chain=2, segments=seg, chain_debug=64, relay_waves=waves, **opts).items(): p.set_option(k, v)
            out = p.sdft(x); torch.cuda.synchronize()
            p.set_option("profile", 1)
            out = p.sdft(x); torch.cuda.synchronize()
            pr = p.profile()
            fn = getattr(p.api.lib, "sdft_hip_chain_stats_" + combo); fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
            st = np.zeros(32, dtype=np.uint64); fn(p._p, st.ctypes.data)
            st = st.reshape(8, 4)
            print(f"{combo} n={n} m={m} waves={waves} segments={seg} last_chain={p.get_option('last_chain')} len={p.get_option('last_chunk_len')}: carry {pr['carry'][0]:.3f} ms forward {pr['forward'][0]:.3f} ms (last segment's launch, workgroup 0:)")
            for w in range(waves):
                t = max(int(st[w, 3]), 1)
                print(f"  wave {w}: turns {t:5d}  products {int(st[w,0])/t:7.0f}  token wait {int(st[w,1])/t:7.0f}  chain+publish {(int(st[w,2]) & 0xffffffff)/t:7.0f}  rest {(int(st[w,2]) >> 32)/t:7.0f} cycles per turn")
            p.close()
