"""Development probe (libraries built with SDFT_HIP_EXTRA_FLAGS=-DSDFT_SELF_STAMPS python -m sdft_amd.build; the key is a test hook: the plan
moves to libsdft_hip_hooks.so): shader-cycle stamps of the last chunk's workgroup
of a self-carried call -- kernel entry, fold started, fold done, FFT done, state ready, first group done, end."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
names = ["entry", "fold starts", "fold done", "FFT done", "state ready", "first group done", "end"]
for n in (4096, 12000, 48000, 131072):
    x = torch.from_numpy(sine_sweep(n)).cuda()
    pm = capi.PlacedMatrix((n, 1024), torch.complex128) if n * 1024 * 16 >= (64 << 20) else None      # (placed like bench.py's matrices)
    out = pm.tensor if pm else torch.empty((n, 1024), dtype=torch.complex128, device="cuda")
    st = torch.zeros(8, dtype=torch.int64, device="cuda")
    with SDFT(1024, "hann", 1.0, "f32f64") as p:
        p.set_option("self_stamps", st.data_ptr())
        for _ in range(3): p.sdft(x, out)
        p.synchronize()
        v = st.cpu().numpy().astype(np.int64)
        d = [int(v[i] - v[0]) for i in range(7)]
        print(f"n={n} {'placed' if pm else 'plain'} chunks={p.get_option('last_chunks')} len={p.get_option('last_chunk_len')}: " + ", ".join(f"{nm} {dd}" for nm, dd in zip(names, d)), flush=True)
    del out
    if pm: pm.free()
