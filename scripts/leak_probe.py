import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep
xl = torch.from_numpy(sine_sweep(6000)).cuda(); xh = torch.from_numpy(sine_sweep(100)).cuda()
gain = np.linspace(1.0, 0.5, 256)
def free():
    torch.cuda.synchronize(); torch.cuda.empty_cache(); return torch.cuda.mem_get_info()[0]
def run(name, body, opts={}, combo="f32f64", cycles=100):
    for _ in range(3):
        with SDFT(256, "hann", 1.0, combo) as p:
            for k, v in opts.items(): p.set_option(k, v)
            body(p)
    f0 = free()
    for _ in range(cycles):
        with SDFT(256, "hann", 1.0, combo) as p:
            for k, v in opts.items(): p.set_option(k, v)
            body(p)
    f1 = free()
    print(f"{name}: {(f0 - f1) / cycles / 1024:.1f} KiB per cycle", flush=True)
run("alloc/free only", lambda p: None)
run("sdft long", lambda p: p.sdft(xl))
run("sdft hop", lambda p: p.sdft(xh))
run("isdft", lambda p: p.isdft(p.sdft(xl)))
run("process long", lambda p: p.process(xl))
run("process gain", lambda p: p.process(xl, "gain", gain=gain))
run("process hop", lambda p: p.process(xh))
run("process ordered", lambda p: p.process(xl), {"fused_exact": 1})
run("sdft carry=1", lambda p: p.sdft(xl), {"carry": 1})
run("sdft f32f32", lambda p: p.sdft(xl), {}, "f32f32")
