"""Development probe (round 6): can the analysis' store stream be made faster on a matrix that lies in ONE kind of device memory (a plain hipMalloc's
bad case, 5.6-5.85 TB/s) by its own geometry -- chunk length (the distance between the rows written at the same time), workgroup placement, row phase?
One arena; a 16.4 GB window inside its first stretch (one kind) and the window the library centres on the change of kind, the same patterns on both."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdft_amd import capi
lib = capi.load()
n, m = 1000000, 1024
nbytes = n * m * 16
pm = capi.PlacedMatrix((n, m), torch.complex128)
print("placement:", pm.info, flush=True)
base = pm.ptr - pm.info["window_offset"]
windows = (("one kind (the allocation's start)", base), ("two kinds (the library's window)", pm.ptr))
def rate(ptr, pattern, chunk_len, lanes=8):
    ms = lib.sdft_hip_store_ceiling(C.c_void_p(ptr), nbytes, pattern, 1024, lanes, chunk_len, 3)
    return nbytes / (ms * 1e-3) / 1e9
names = {2: "b -> chunk b", 4: "XCD-contiguous", 5: "b -> chunk b, staggered", 6: "XCD-contiguous, staggered", 102: "2 regions", 104: "4 regions", 116: "16 regions", 164: "64 regions", 3: "non-temporal", 0: "linear fill"}
for label, ptr in windows:
    print(f"--- {label} ---", flush=True)
    for pattern in (4, 2, 6, 5, 102, 104, 116, 164, 3, 0):
        print(f"  {names[pattern]:28s} chunks of 1960: {rate(ptr, pattern, 1960):6.0f} GB/s", flush=True)
    for cl in (1960, 1953, 1961, 1984, 2000, 2048, 1920, 1999, 2040, 2056, 3907, 3912, 4096, 977, 980, 1024, 488, 512, 7813, 15625):
        print(f"  XCD-contiguous, chunks of {cl:6d} rows ({(n + cl - 1) // cl:5d} chunks): {rate(ptr, 4, cl):6.0f} GB/s   b -> chunk b: {rate(ptr, 2, cl):6.0f}", flush=True)
    for lanes in (1, 2, 4, 16, 32, 64):
        print(f"  XCD-contiguous, chunks of 1960, a barrier every {lanes:2d} rows: {rate(ptr, 4, 1960, lanes):6.0f} GB/s", flush=True)
pm.free()
