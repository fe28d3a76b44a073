"""Stress of the resident kernel's protocol (round 6): the hop loop with pauses around its idle time-out (200 us), so that calls race with the kernel leaving;
every hop compared bit for bit with a plan that launches.  Prints calls, launches, missed (calls that found the kernel gone and were rung again)."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import noise
random.seed(7)
m, hop, hops = 1000, 100, 3000
x = torch.from_numpy(noise(hop * hops, seed=3)).cuda()
pa, pb = SDFT(m, "hann", 1.0, "f32f64"), SDFT(m, "hann", 1.0, "f32f64")
pa.set_option("resident", 1)
da = torch.empty((hop, m), dtype=torch.complex128, device="cuda"); db = torch.empty_like(da)
ya = torch.empty(hop, dtype=torch.float32, device="cuda"); yb = torch.empty_like(ya)
bad = 0
pauses = [0, 0, 0, 50e-6, 150e-6, 190e-6, 200e-6, 210e-6, 230e-6, 300e-6, 1e-3]
t0 = time.perf_counter()
for i in range(hops):
    seg = x[i * hop:(i + 1) * hop]
    pa.sdft(seg, da)
    p = random.choice(pauses)
    if p:
        t = time.perf_counter()
        while time.perf_counter() - t < p: pass
    pa.isdft(da, ya)
    if i % 3 == 0:
        p = random.choice(pauses)
        t = time.perf_counter()
        while time.perf_counter() - t < p: pass
    if i % 50 == 0:                                  # compare (the blocking copies also retire / wait for the kernel)
        pb.sdft(seg, db); pb.isdft(db, yb)
        if not (torch.equal(da, db) and torch.equal(ya, yb)): bad += 1
    else:
        pb.sdft(seg, db)                             # keep the reference plan's state in step
print(f"{hops} hops in {time.perf_counter() - t0:.2f} s: mismatches {bad}; resident calls {pa.get_option('resident_calls')}, launches {pa.get_option('resident_launches')}, "
      f"missed {pa.get_option('resident_missed')}, still on: {pa.get_option('resident')}; warning: {pa.api.last_warning()}")
sa, sb = pa.state(), pb.state()
print("state equal:", all(np.array_equal(a, b) for a, b in zip(sa[:3], sb[:3])) and sa[3] == sb[3])
pa.close(); pb.close()
