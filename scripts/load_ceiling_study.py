"""Load-only ceilings by access shape and placement (the synthesis' side of the round-5 study): linear grid-stride loads against
whole rows read in step by one workgroup per chunk of rows, workgroup b -> chunk b against R regions in turn.
    python scripts/load_ceiling_study.py [rounds]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd import capi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = capi.load()
m = 1024
print(f"device: {torch.cuda.get_device_name(0)}")
for rows, chunk_lens in ((1_000_000, (1960, 3907, 245, 64)), (64 * 48000, (6000, 1500, 64))):
    buf = torch.zeros((rows, m), dtype=torch.complex128, device="cuda")
    nbytes = rows * m * 16
    cases = [("linear grid-stride", None, 0)]
    for cl in chunk_lens:
        for R in (0, 2, 8, 64):
            cases.append((f"rows in step, chunks of {cl:5d} rows, {('%d regions' % R) if R else 'b -> chunk b'}", cl, R))
    res = {c[0]: [] for c in cases}
    for r in range(rounds):
        for label, cl, R in cases:
            torch.cuda.synchronize()
            ms = lib.sdft_hip_load_ceiling(buf.data_ptr(), nbytes, 3) if cl is None else lib.sdft_hip_load_rows_ceiling(buf.data_ptr(), nbytes, m, cl, R, 3)
            res[label].append(nbytes / (ms * 1e-3) / 1e9)
    for label, *_ in cases:
        v = res[label]
        print(f"{nbytes / 1e9:5.1f} GB  {label:60s} median {np.median(v):7.0f} GB/s = {np.median(v) / 8000:5.1%} of 8 TB/s")
    del buf
    torch.cuda.empty_cache()
