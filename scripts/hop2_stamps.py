"""Development probe (library built with SDFT_HIP_EXTRA_FLAGS=-DSDFT_HOP_STAMPS): phases of forward_hop2_kernel's LAST
workgroup (the last time part: the longest way) in realtime ticks (100 MHz): start | staged (barrier) | state run through the
samples before the part | loop done, for the recurrence wave; start | staged | loop done for the window wave.
    python scripts/hop2_stamps.py [hop_parts ...]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
m, hop, total = 1000, 100, 20000
x = torch.from_numpy(sine_sweep(total)).cuda()
d = torch.empty((hop, m), dtype=torch.complex128, device="cuda")
for parts in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    p = SDFT(m, "hann", 1.0, "f32f64")
    p.set_option("hop_parts", parts)
    fn = getattr(p.api.lib, "sdft_hip_chain_stats_f32f64"); fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
    acc = np.zeros(5); cnt = 0
    for i in range(0, total, hop):
        p.sdft(x[i:i + hop], d)
        st = np.zeros(32, dtype=np.uint64)
        if fn(p._p, st.ctypes.data) == 0 and i >= 10 * hop and st[2] > st[0]:
            s = st.astype(np.int64)
            acc += np.array([s[1] - s[0], s[3] - s[1], s[2] - s[3], s[5] - s[4], s[6] - s[5]]) / 100.0; cnt += 1
    print("forward_hop2_kernel, %d parts, last workgroup, us: recurrence wave prologue %.2f, state through the earlier samples %.2f, loop %.2f | window wave prologue %.2f loop %.2f (%d launches)"
          % (p.get_option("last_hop_parts"), *(acc / max(cnt, 1)), cnt))
    p.close()
