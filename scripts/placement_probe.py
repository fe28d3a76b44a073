"""Does the achievable store rate depend on WHICH memory a matrix lives in?  Twelve 16.4 GB buffers allocated one after the other in
one process (all kept), each probed with the row-lockstep store kernel, workgroup b -> chunk b and every XCD a contiguous eighth;
then the same buffers again in reverse order.    python scripts/placement_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from sdft_amd import capi

lib = capi.load()
n, m = 1_000_000, 1024
nbytes = n * m * 16
bufs = []
for i in range(12):
    free, _ = torch.cuda.mem_get_info()
    if free < nbytes * 1.1:
        break
    bufs.append(torch.empty((n, m), dtype=torch.complex128, device="cuda"))


def probe(b):
    ms2 = lib.sdft_hip_store_ceiling(b.data_ptr(), nbytes, 2, m, 8, 1960, 6)
    ms4 = lib.sdft_hip_store_ceiling(b.data_ptr(), nbytes, 4, m, 8, 1960, 6)
    ld = lib.sdft_hip_load_rows_ceiling(b.data_ptr(), nbytes, m, 1960, 0, 6)
    return nbytes / ms2 / 1e6, nbytes / ms4 / 1e6, nbytes / ld / 1e6


for rnd in range(2):
    order = list(range(len(bufs))) if rnd == 0 else list(reversed(range(len(bufs))))
    for i in order:
        a, b, c = probe(bufs[i])
        print(f"pass {rnd} buffer {i:2d} at 0x{bufs[i].data_ptr():x}: store-only {a:5.0f} / {b:5.0f} GB/s   load-only (rows in step) {c:5.0f} GB/s", flush=True)
