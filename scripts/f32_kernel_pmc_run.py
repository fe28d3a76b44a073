"""The bin-pair analysis kernel alone on a quarter of the chip (192 CUs held, so that HBM does not bound it) for a counter pass:
rocprofv3 --pmc ... -- python3 scripts/f32_kernel_pmc_run.py"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdft_amd import capi
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

lib = capi.load()
m, n = 4096, 65536
x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
p = SDFT(m, "blackman", 1.0, "f32f32")
p.set_option("float_carry_parallel", 1)
import os
p.set_option("rows_split", int(os.environ.get("SDFT_SPLIT", "1")))
d = p.sdft(x)
p.set_option("async", 1)
held = int(sys.argv[1]) if len(sys.argv) > 1 else 192
if held:
    lib.sdft_hip_hold_cus(held, 200.0)
    time.sleep(0.01)
t0 = time.perf_counter()
for _ in range(4):
    p.sdft(x, d)
p.synchronize()
w = (time.perf_counter() - t0) / 4
lib.sdft_hip_hold_cus(0, 0.0)
print(f"held {held}: {n * (m * 8 + 4) / w / 1e9:.0f} GB/s, {n * (m * 8 + 4) / w / 1e9 / (256 - held):.1f} GB/s per free CU")
p.close()
