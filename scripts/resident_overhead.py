"""Development probe: what the resident protocol costs per call -- the hop loop at sizes where the work is next to nothing, and at the reference's size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
for m, hop, total in ((64, 8, 1600), (1000, 8, 1600), (1000, 100, 20000), (1024, 256, 51200)):
    r = bench.hop_streaming(torch, np, SDFT, sine_sweep, "f32f64", np.float32, torch.complex128, 0, m=m, hop=hop, total=total)
    print(f"m={m} hop={hop}: sync {r['us_per_hop_sync']} us, resident {r['us_per_hop_resident_sync']} us, async {r['us_per_hop_async']} us; kernels {r['forward_kernel_us']} + {r['inverse_kernel_us']} us; {r['resident']}", flush=True)
