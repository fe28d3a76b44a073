"""Where the relay's waves spend their cycles per turn (chain_debug = 64, the measurement build), flow mode against alone."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
combo, n = "f32f32", 262144
for m, win in ((1024, "hann"), (4096, "blackman")):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    for label, opts in (("flow", {}), ("alone", {"relay_flow": 0, "segments": 1})):
        p = SDFT(m, win, 1.0, combo)
        for k, v in dict(chain_debug=64, **opts).items(): p.set_option(k, v)
        out = p.sdft(x); torch.cuda.synchronize()
        p.set_option("profile", 1)
        out = p.sdft(x); torch.cuda.synchronize()
        pr = p.profile()
        fn = getattr(p.api.lib, "sdft_hip_chain_stats_" + combo); fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
        st = np.zeros(32, dtype=np.uint64); fn(p._p, st.ctypes.data)
        st = st.reshape(8, 4)
        print(f"m={m} {label}: flow={p.get_option('last_flow')} carry {pr['carry'][0]:.3f} ms forward {pr['forward'][0]:.3f} ms (workgroup 0, cycles per turn of a wave = 8 blocks of the chain)")
        for w in range(8):
            t = max(int(st[w, 3]), 1)
            pr_, tw, ch, rs = int(st[w, 0]) / t, int(st[w, 1]) / t, (int(st[w, 2]) & 0xffffffff) / t, (int(st[w, 2]) >> 32) / t
            print(f"  wave {w}: turns {t:5d}  products {pr_:7.0f}  token wait {tw:7.0f}  chain+publish {ch:7.0f}  rest {rs:7.0f}  sum {pr_ + tw + ch + rs:7.0f}")
        p.close()
