"""The same store-only probe every second for a minute, with the clocks and the socket power rocm-smi reports while it runs: does
the achieved store rate switch between two levels on one lease, and what else changes when it does?
    python scripts/mode_monitor.py [seconds]"""
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from sdft_amd import capi

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = capi.load()
n, m = 1_000_000, 1024
out = torch.empty((n, m), dtype=torch.complex128, device="cuda")
nbytes = n * m * 16
latest = {"smi": ""}
stop = threading.Event()


def smi():
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True)
        vals = []
        for key in ("fclk", "mclk", "sclk", "socclk"):
            mm = re.search(key + r" clock level: \S+ \((\d+)Mhz\)", r.stdout)
            vals.append(f"{key} {mm.group(1) if mm else '?'}")
        mm = re.search(r"Power \(W\): ([0-9.]+)", r.stdout)
        vals.append(f"power {mm.group(1) if mm else '?'} W")
        mm = re.findall(r"Temperature \(Sensor (\w+)\) \(C\): ([0-9.]+)", r.stdout)
        vals.append(" ".join(f"{a} {b}C" for a, b in mm[:3]))
        latest["smi"] = ", ".join(vals)


th = threading.Thread(target=smi, daemon=True)
th.start()
t0 = time.time()
while time.time() - t0 < seconds:
    ms2 = lib.sdft_hip_store_ceiling(out.data_ptr(), nbytes, 2, m, 8, 1960, 60)
    ms4 = lib.sdft_hip_store_ceiling(out.data_ptr(), nbytes, 4, m, 8, 1960, 60)
    print(f"[{time.time() - t0:5.1f} s] store-only {nbytes / ms2 / 1e6:5.0f} / {nbytes / ms4 / 1e6:5.0f} GB/s   {latest['smi']}", flush=True)
stop.set()
