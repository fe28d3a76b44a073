"""Store-only probes with ordinary against non-temporal stores (row-lockstep, workgroups placed by XCD: patterns 4 and 7) on several allocations of
16.4 GB and on matrices of 786 MB and 4.3 GB.    python scripts/nt_store_probe.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from sdft_amd import capi

lib = capi.load()
lib.sdft_hip_store_ceiling.restype = C.c_double
lib.sdft_hip_store_ceiling.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_int]
print(f"device: {torch.cuda.get_device_name(0)}")


def rate(ptr, nbytes, pattern, chunk_len):
    ms = lib.sdft_hip_store_ceiling(ptr, (nbytes // 16384) * 16384, pattern, 1024, 8, chunk_len, 3)
    return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0


for label, rows, chunk_len, count in (("16.4 GB", 1000000, 1960, 8), ("4.3 GB", 262144, 512, 3), ("786 MB", 48000, 192, 3)):
    nbytes = rows * 16384
    bufs = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(count)]
    for i, t in enumerate(bufs):
        r = [rate(t.data_ptr(), nbytes, pat, chunk_len) for pat in (4, 7, 2, 3, 4, 7)]
        print(f"{label} allocation {i}: placed by XCD ordinary / non-temporal {r[0]:.0f} / {r[1]:.0f}   workgroup b -> chunk b ordinary / non-temporal {r[2]:.0f} / {r[3]:.0f}   again placed {r[4]:.0f} / {r[5]:.0f} GB/s", flush=True)
    del bufs
    torch.cuda.empty_cache()
