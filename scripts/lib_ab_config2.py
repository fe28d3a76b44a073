"""configs[2] analysis with the library of another tree (a checkout under _ab/<name>, built there) against this tree's, each in
a child process on the same box, several rounds in turn.    python scripts/lib_ab_config2.py _ab/r04 [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
m, window, n = 4096, "blackman", 262144
x = torch.from_numpy(sine_sweep(n)).cuda()
d = torch.empty((n, m), dtype=torch.complex64, device="cuda")
p = SDFT(m, window, 1.0, "f32f32")
for _ in range(2): p.sdft(x, d)
p.synchronize()
out = {}
for mode in ("sync", "async"):
    p.set_option("async", 1 if mode == "async" else 0)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): p.sdft(x, d)
    p.synchronize()
    out[mode] = (time.perf_counter() - t0) / 5 * 1e3
p.set_option("profile", 1)
for _ in range(3): p.sdft(x, d)
pr = p.profile()
out["carry_ms"] = (pr["delta"][0] + pr["carry"][0]) / max(pr["forward"][1], 1)
out["forward_ms"] = pr["forward"][0] / max(pr["forward"][1], 1)
print(json.dumps(out))
'''
other = os.path.join(ROOT, sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in range(rounds):
    for label, tree in (("this tree", ROOT), (sys.argv[1], other)):
        q = subprocess.run([sys.executable, "-c", CHILD, tree], capture_output=True, text=True, cwd=tree)
        line = [l for l in q.stdout.splitlines() if l.startswith("{")]
        print(f"{label:12s}", line[-1] if line else q.stderr[-400:])
