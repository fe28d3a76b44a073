"""One geometry of the exact-carry relay on configs[2], a few calls, for a counter pass: python3 scripts/relay_variant.py default|groups2 (see relay_on_64_cus.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
n, m = 262144, 4096
x = torch.from_numpy(sine_sweep(n, dtype=np.float32)).cuda()
d = torch.empty((n, m), dtype=torch.complex64, device="cuda")
p = SDFT(m, "blackman", 1.0, "f32f32")
if len(sys.argv) > 1 and sys.argv[1] == "groups2":
    for k, v in (("chain_block", 64), ("relay_groups", 2), ("relay_waves", 8)):
        p.set_option(k, v)
for _ in range(4):
    p.sdft(x, d)
p.synchronize()
p.close()
