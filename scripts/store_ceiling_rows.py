"""Store-only ceiling by row geometry: does a 16000-byte row (m = 1000, the reference's test size) cost the store stream
what the analysis loses against m = 1024?  Pattern 2 = one workgroup per time chunk writing whole rows in lockstep."""
import sys

import torch

sys.path.insert(0, ".")
from sdft_amd import capi

lib = capi.load()
buf = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
for slots, rows, chunk in ((1024, 352800, 192), (1000, 352800, 192), (1000, 352800, 696), (1008, 352800, 192), (992, 352800, 192), (960, 352800, 192),
                           (1024, 48000, 192), (1000, 48000, 192)):
    nbytes = rows * slots * 16
    res = []
    for pattern, lanes in ((0, 64), (2, 8)):
        ms = lib.sdft_hip_store_ceiling(buf.data_ptr(), nbytes, pattern, slots, lanes, chunk, 10)
        res.append(nbytes / (ms * 1e-3) / 1e9)
    print(f"row of {slots} bins ({slots * 16} B), {rows} rows, chunks of {chunk}: linear fill {res[0]:6.0f} GB/s, row-lockstep {res[1]:6.0f} GB/s")
