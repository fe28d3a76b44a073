"""Development probe: where a process_hop_kernel launch spends its time.  Needs a library built with
SDFT_HIP_EXTRA_FLAGS=-DSDFT_HOP_STAMPS (python -m sdft_amd.build); the stamps are s_memrealtime (100 MHz) of the
workgroup that drew the last ticket: start | state + input loaded, differences staged | samples done | ticket drawn | end."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

m, hop, total = 1000, 100, 20000
x = torch.from_numpy(sine_sweep(total)).cuda()
y = torch.empty_like(x)
p = SDFT(m, "hann", 1.0, "f32f64")
fn = getattr(p.api.lib, "sdft_hip_chain_stats_f32f64"); fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
acc = np.zeros(4)
cnt = 0
for i in range(0, total, hop):
    p.process(x[i:i + hop], out=y[i:i + hop])
    st = np.zeros(32, dtype=np.uint64)
    if fn(p._p, st.ctypes.data) == 0 and st[4] > st[0] and i >= 10 * hop:
        acc += np.diff(st[:5].astype(np.int64)) / 100.0
        cnt += 1
print("process_hop_kernel, last workgroup, us: prologue %.2f | samples %.2f | state stores + ticket %.2f | combine %.2f  (total %.2f, %d launches)"
      % (*(acc / max(cnt, 1)), acc.sum() / max(cnt, 1), cnt))
p.close()
