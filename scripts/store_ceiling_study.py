"""Round-5 store-ceiling study (review item 9): does the row-lockstep store stream of the headline workload (n = 1e6 rows of
16 KiB, FD double) reach more than its 5.8-6.15 TB/s with (a) every XCD writing a contiguous eighth of the matrix instead of
consecutive workgroups -> consecutive 32 MB regions, (b) workgroups started at different row phases, (c) another number of rows
per barrier, (d) one or four rounds of the chip instead of two?  Store-only kernels (sdft_hip_store_ceiling), interleaved rounds.
    python scripts/store_ceiling_study.py [rounds]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd import capi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = capi.load()
n, m = 1_000_000, 1024
out = torch.empty((n, m), dtype=torch.complex128, device="cuda")
nbytes = n * m * 16
print(f"device: {torch.cuda.get_device_name(0)}; {nbytes / 1e9:.2f} GB matrix, rows of {m * 16} B")
cases = []
for chunk_len in (1960, 3907, 980):
    for pattern, label in ((2, "workgroup b -> chunk b"), (4, "XCD-contiguous chunks"), (5, "staggered row phase"), (6, "XCD-contiguous + staggered")):
        cases.append((f"{label:30s} chunks of {chunk_len:4d} rows, barrier every 8", pattern, 8, chunk_len))
for sync in (0, 2, 4, 16, 32):
    cases.append((f"{'workgroup b -> chunk b':30s} chunks of 1960 rows, barrier every {sync}", 2, sync, 1960))
for R in (2, 3, 4, 16, 32, 64, 128, 256):
    cases.append((f"{'%d regions in turn' % R:30s} chunks of 1960 rows, barrier every 8", 100 + R, 8, 1960))
cases.append((f"{'linear fill (grid-stride)':30s}", 0, 64, 1))
cases.append((f"{'non-temporal stores':30s} chunks of 1960 rows, barrier every 8", 3, 8, 1960))
res = {c[0]: [] for c in cases}
for r in range(rounds):
    for label, pattern, sync, chunk_len in cases:
        torch.cuda.synchronize()
        ms = lib.sdft_hip_store_ceiling(out.data_ptr(), nbytes, pattern, m, sync, chunk_len, 4)
        res[label].append(nbytes / (ms * 1e-3) / 1e9)
for label, *_ in cases:
    v = res[label]
    print(f"{label:75s} median {np.median(v):7.0f} GB/s  (min {min(v):7.0f}, max {max(v):7.0f})  = {np.median(v) / 8000:5.1%} of 8 TB/s")
# the batch workload's geometry: 64 channels x 48000 rows = 50 GB, 512 chunks of 6000 rows
del out
torch.cuda.empty_cache()
free, _ = torch.cuda.mem_get_info()
if free > 52e9:
    rows = 64 * 48000
    big = torch.empty((rows, m), dtype=torch.complex128, device="cuda")
    bb = rows * m * 16
    bcases = [("workgroup b -> chunk b", 2), ("XCD-contiguous (8 regions)", 4), ("2 regions in turn", 102), ("64 regions in turn", 164)]
    bres = {c[0]: [] for c in bcases}
    for r in range(rounds):
        for label, pattern in bcases:
            torch.cuda.synchronize()
            ms = lib.sdft_hip_store_ceiling(big.data_ptr(), bb, pattern, m, 8, 6000, 2)
            bres[label].append(bb / (ms * 1e-3) / 1e9)
    for label, _ in bcases:
        v = bres[label]
        print(f"50 GB, 512 chunks of 6000 rows: {label:30s} median {np.median(v):7.0f} GB/s  = {np.median(v) / 8000:5.1%} of 8 TB/s")
