"""The analysis launches with every XCD taking a contiguous eighth of the (channel, chunk) workgroups (option xcd_map = 1)
against workgroup b -> chunk b (0), and one round of the chip against two at n = 1e6; interleaved, asynchronous calls, HIP
events around the kernel.    python scripts/xcd_map_ab.py [rounds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep, sweep_batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(f"device: {torch.cuda.get_device_name(0)}")


def run(label, m, n, ch, combo, variants, reps):
    esz = 16 if combo[3:] == "f64" else 8
    x = torch.from_numpy(sweep_batch(ch, n) if ch > 1 else sine_sweep(n)).cuda()
    d = torch.empty((ch, n, m) if ch > 1 else (n, m), dtype=torch.complex128 if esz == 16 else torch.complex64, device="cuda")
    plans = []
    for vl, opts in variants:
        p = SDFT(m, "hann", 1.0, combo, channels=ch)
        p.set_option("async", 1)
        for k, v in opts.items():
            p.set_option(k, v)
        for _ in range(2):
            p.sdft(x, d)
        p.synchronize()
        plans.append((vl, p))
    res = {vl: [] for vl, _ in plans}
    for r in range(rounds):
        for vl, p in plans:
            p.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                p.sdft(x, d)
            p.synchronize()
            res[vl].append((time.perf_counter() - t0) / reps)
    b = ch * n * (m * esz + 4)
    for vl, p in plans:
        w = float(np.median(res[vl]))
        print(f"{label:34s} {vl:34s} {w * 1e3:8.3f} ms per call = {b / w / 1e9:6.0f} GB/s = {b / w / 8e12:5.1%} of peak   chunks {p.get_option('last_chunks')} x {p.get_option('last_chunk_len')}")
        p.close()
    del x, d
    torch.cuda.empty_cache()


V = (("xcd_map=1", {"xcd_map": 1}), ("xcd_map=0", {"xcd_map": 0}))
run("n=1e6 m=1024 f32f64", 1024, 1_000_000, 1, "f32f64", V + (("xcd_map=1, one round (chunk 3912)", {"xcd_map": 1, "chunk": 3912}), ("xcd_map=0, one round (chunk 3912)", {"xcd_map": 0, "chunk": 3912}),
                                                              ("xcd_map=1, four rounds (chunk 984)", {"xcd_map": 1, "chunk": 984})), 8)
run("n=48000 m=1024 f32f64", 1024, 48000, 1, "f32f64", V, 60)
run("n=262144 m=1024 f32f64", 1024, 262144, 1, "f32f64", V, 20)
run("64 ch x 48000 m=1024 f32f64", 1024, 48000, 64, "f32f64", V, 4)
run("n=262144 m=4096 f32f32", 4096, 262144, 1, "f32f32", V, 6)
run("n=1e6 m=1000 f32f64", 1000, 1_000_000, 1, "f32f64", V, 8)
