"""Development probe: where the waves of carry_chain_kernel spend their cycles (option chain_debug = 16)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
for combo, n, m, opts in (("f32f32", 262144, 1024, {}), ("f32f64", 262144, 1024, {"carry": 1})):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    for dbg in (16, 17, 18):
        p = SDFT(m, "hann", 1.0, combo)
        for k, v in dict(chain=2, segments=1, chain_debug=dbg, **opts).items(): p.set_option(k, v)
        out = p.sdft(x); torch.cuda.synchronize()
        out = p.sdft(x); torch.cuda.synchronize()
        fn = getattr(p.api.lib, "sdft_hip_chain_stats_" + combo); fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
        st = np.zeros(32, dtype=np.uint64); fn(p._p, st.ctypes.data)
        st = st.reshape(8, 4)
        print(f"{combo} n={n} m={m} debug={dbg} (bit 0: consumer idles, bit 1: producers idle): rounds={st[0,3]}")
        for w in range(8):
            if st[w, 3]: print(f"  wave {w}: work {st[w,0]/max(st[w,3],1):8.0f}  tail-wait {st[w,1]/max(st[w,3],1):8.0f}  barrier {st[w,2]/max(st[w,3],1):8.0f} cycles per round")
        p.close()
