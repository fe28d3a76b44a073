"""The store-only probe on windows of 50.3 GB (one GPU's share of configs[4]: 64 x 48000 rows of 16 KiB, chunks of 6000 rows) every 4 GiB inside one allocation."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from sdft_amd import capi

lib = capi.load()
win = 16384 * 64 * 48000
free, _ = torch.cuda.mem_get_info()
abytes = ((free - (24 << 30)) // (4 << 30)) * (4 << 30)
arena = torch.empty(abytes, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
print(f"arena of {abytes / 1e9:.0f} GB")
for rep in range(2):
    out = []
    for o in range(0, abytes - win + 1, 4 << 30):
        ms = lib.sdft_hip_store_ceiling(base + o, win, 4, 1024, 8, 6000, 2)
        out.append(f"{o >> 30}:{win / (ms * 1e-3) / 1e9:.0f}")
    print("window offset GiB -> GB/s:  " + "  ".join(out), flush=True)
