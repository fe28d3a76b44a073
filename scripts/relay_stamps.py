"""Development probe: the critical path of carry_relay_kernel turn by turn (option chain_debug = 128): cycles from a wave
seeing the token to the end of its additions, to the token's store, and from there to the next wave seeing it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
for combo, n, m, win, opts in (("f32f32", 262144, 1024, "hann", {}), ("f32f64", 262144, 1024, "hann", {"carry": 1})):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    for waves in (4, 6, 8):
        p = SDFT(m, win, 1.0, combo)
        for k, v in dict(chain=2, segments=1, chain_debug=128, relay_waves=waves, **opts).items(): p.set_option(k, v)
        out = p.sdft(x); torch.cuda.synchronize()
        out = p.sdft(x); torch.cuda.synchronize()
        fn = getattr(p.api.lib, "sdft_hip_relay_stamps_" + combo); fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        st = np.zeros(3 * 1024, dtype=np.uint64); rc = fn(p._p, st.ctypes.data, st.size)
        st = st.reshape(1024, 3).astype(np.int64)[8:1000]
        chain = st[:, 1] - st[:, 0]; store = st[:, 2] - st[:, 1]; hand = st[1:, 0] - st[:-1, 2]; period = st[1:, 2] - st[:-1, 2]
        print(f"{combo} m={m} waves={waves} rc={rc}: per block: token seen -> additions done {np.median(chain):.0f} (p90 {np.percentile(chain,90):.0f}), -> token stored {np.median(store):.0f}, "
              f"-> next wave sees it {np.median(hand):.0f} (p90 {np.percentile(hand,90):.0f}); period {np.median(period):.0f} (mean {period.mean():.0f}) cycles")
        p.close()
