"""Which memory takes the analysis' store stream fast?  One large allocation (an arena), the store-only probe (row-lockstep, 16 KiB rows, two rounds of the
chip: what bench.py's placement uses) on windows of 16.4 GB at different offsets inside it, twice; then separate allocations of 16.4 GB, with their addresses.
    python scripts/arena_probe.py [arena_GB=200]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from sdft_amd import capi

lib = capi.load()
lib.sdft_hip_store_ceiling.restype = C.c_double
lib.sdft_hip_store_ceiling.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_int]
arena_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
win = 16384 * 1000000
print(f"device: {torch.cuda.get_device_name(0)}")


def rate(ptr, nbytes):
    ms = lib.sdft_hip_store_ceiling(ptr, (nbytes // 16384) * 16384, 4, 1024, 8, 1960, 2)
    return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0


arena = torch.empty(int(arena_gb * 1e9), dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
print(f"arena of {arena_gb:.0f} GB at {base:#x}")
step = 4 * (1 << 30)
for rep in range(2):
    offs = list(range(0, arena.numel() - win, step))
    print("window offset GiB -> GB/s:  " + "  ".join(f"{o / (1 << 30):.0f}:{rate(base + o, win):.0f}" for o in offs), flush=True)
# finer: windows of 2 GiB (their own rate, not the 16.4 GB one)
small = 2 * (1 << 30)
print("2 GiB windows, offset GiB -> GB/s:  " + "  ".join(f"{o / (1 << 30):.0f}:{rate(base + o, small):.0f}" for o in range(0, min(arena.numel() - small, 64 * (1 << 30)), small)), flush=True)
del arena
torch.cuda.empty_cache()
bufs = []
for i in range(12):
    free, _ = torch.cuda.mem_get_info()
    if free < win * 1.1:
        break
    t = torch.empty(win, dtype=torch.uint8, device="cuda")
    bufs.append(t)
    print(f"allocation {i:2d} at {t.data_ptr():#x}: {rate(t.data_ptr(), win):.0f} GB/s", flush=True)
