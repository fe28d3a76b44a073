"""Store-only ceiling with ordinary against non-temporal stores (row-lockstep pattern of the forward kernel), by matrix size."""
import sys

import torch

sys.path.insert(0, ".")
from sdft_amd import capi

lib = capi.load()
buf = torch.empty(17 << 30, dtype=torch.uint8, device="cuda")
for rows, chunk in ((1000000, 1960), (48000, 192), (262144, 512), (12000, 64)):
    nbytes = rows * 1024 * 16
    line = []
    for rep in range(2):
        for pattern, name in ((2, "ordinary"), (3, "non-temporal")):
            ms = lib.sdft_hip_store_ceiling(buf.data_ptr(), nbytes, pattern, 1024, 8, chunk, 10)
            line.append(f"{name} {nbytes / (ms * 1e-3) / 1e9:6.0f}")
    print(f"{rows:8d} rows of 16 KiB ({nbytes / 1e9:6.2f} GB), chunks of {chunk}: " + " | ".join(line) + " GB/s")
