"""Ad-hoc perf probe (development aid): forward/inverse device time for a few geometries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdft_amd.sdft import SDFT
from sdft_amd.signals import sine_sweep

def run(n, m, window="hann", combo="f32f64", channels=1, reps=5, **opts):
    td = np.float32 if combo[:3] == "f32" else np.float64
    x = torch.from_numpy(np.stack([sine_sweep(n, dtype=td)] * channels) if channels > 1 else sine_sweep(n, dtype=td)).cuda()
    p = SDFT(m, window, 1.0, combo, channels)
    for k, v in opts.items(): p.set_option(k, v)
    p.set_option("profile", 1); p.set_option("async", 1)
    cdt = torch.complex128 if combo[3:] == "f64" else torch.complex64
    shape = (n, m) if channels == 1 else (channels, n, m)
    out = torch.empty(shape, dtype=cdt, device="cuda")
    y = None
    for _ in range(2):
        p.sdft(x, out); y = p.isdft(out)
    p.synchronize(); p.profile()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.sdft(x, out); p.isdft(out, y)
    p.synchronize(); wall = (time.perf_counter() - t0) / reps
    pr = p.profile()
    esz = 16 if combo[3:] == "f64" else 8
    byts = channels * n * (m * esz + x.element_size())
    f = pr["forward"][0] / pr["forward"][1]; i = pr["inverse"][0] / pr["inverse"][1]
    c = pr["carry"][0] / max(pr["carry"][1], 1); d = pr["delta"][0] / max(pr["delta"][1], 1)
    print(f"n={n} m={m} {window} {combo} ch={channels} opts={opts} chunks={p.get_option('last_chunks')} len={p.get_option('last_chunk_len')} chain={p.get_option('last_chain')} flow={p.get_option('last_flow')}: "
          f"fwd {f:.3f} ms ({byts/f/1e9:.0f} GB/s, {channels*n/f/1e3:.1f} Msamp/s) carry {c:.3f} delta {d:.3f} inv {i:.3f} ms ({byts/i/1e9:.0f} GB/s) wall/iter {wall*1e3:.3f} ms", flush=True)
    p.close(); del out

def run_process(n, m, window="hann", combo="f32f64", channels=1, reps=5, op="identity", **opts):
    td = np.float32 if combo[:3] == "f32" else np.float64
    x = torch.from_numpy(np.stack([sine_sweep(n, dtype=td)] * channels) if channels > 1 else sine_sweep(n, dtype=td)).cuda()
    p = SDFT(m, window, 1.0, combo, channels)
    for k, v in opts.items(): p.set_option(k, v)
    p.set_option("async", 1)
    gain = torch.linspace(0.5, 1.5, m, dtype=torch.float64 if combo[3:] == "f64" else torch.float32, device="cuda")
    y = torch.empty_like(x)
    for _ in range(2): p.process(x, op, gain=gain, shift=3, out=y)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): p.process(x, op, gain=gain, shift=3, out=y)
    p.synchronize(); wall = (time.perf_counter() - t0) / reps
    print(f"process n={n} m={m} {window} {combo} ch={channels} op={op} opts={opts} path={p.get_option('last_process_path')} "
          f"exact_order={p.get_option('last_fused_exact')} chunks={p.get_option('last_chunks')}: {wall*1e3:.3f} ms  {channels*n/wall/1e6:.1f} Msamples/s", flush=True)
    p.close()

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "process":
        for fe in (0, 1):
            run_process(1000000, 1024, fused_exact=fe)
        run_process(1000000, 1024, op="gain")
        run_process(1000000, 1024, op="shift")
        run_process(1000000, 1024, "blackman", fused_exact=0)
        run_process(48000, 1024, channels=64, fused_exact=0)
        run_process(48000, 1024, channels=64, fused_exact=1)
        run_process(48000, 1024)
        run_process(262144, 1024, "hann", "f32f32")
        run_process(262144, 2048, "blackman", "f32f32")
        run_process(262144, 2048, "blackman", "f32f32", fused_exact=0)
        run_process(48000, 2048, "hann", "f32f32", channels=64)
        run_process(1000000, 1024, carry=1)
        run(1000000, 1024)
        sys.exit(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "base"
    if which == "procj":
        for kw in ({}, {"proc_slots": 2}, {"proc_slots": 4}):
            run_process(1000000, 1024, fused_exact=0, reps=5, **kw)
            run_process(48000, 1024, fused_exact=0, reps=20, **kw)
            run_process(48000, 1024, channels=64, fused_exact=0, reps=3, **kw)
        sys.exit(0)
    if which == "process1":
        # the fused call alone (counter passes): folded form, n = 1e6
        run_process(1000000, 1024, fused_exact=0, reps=3)
        sys.exit(0)
    if which == "invonly":
        # synthesis alone, repeated on the same matrix (no analysis in between), against the round trip
        for n, m, ch in ((48000, 1024, 1), (131072, 1024, 1), (12000, 1024, 1), (48000, 1024, 4)):
            x = torch.from_numpy(sine_sweep(n) if ch == 1 else np.stack([sine_sweep(n)] * ch)).cuda()
            pl = SDFT(m, "hann", 1.0, "f32f64", ch); pl.set_option("async", 1)
            out = torch.empty((n, m) if ch == 1 else (ch, n, m), dtype=torch.complex128, device="cuda")
            pl.sdft(x, out); y = pl.isdft(out); pl.synchronize()
            for rw in (1, 4, 16, 32):
                pl.set_option("inverse_rows", rw)
                pl.isdft(out, y); pl.synchronize()
                t0 = time.perf_counter()
                for _ in range(50): pl.isdft(out, y)
                pl.synchronize(); dt = (time.perf_counter() - t0) / 50
                print(f"n={n} m={m} ch={ch} inverse only, rows per wave {rw}: {dt*1e3:.3f} ms ({ch*n*m*16/dt/1e12:.2f} TB/s)", flush=True)
            pl.set_option("inverse_rows", 0)
            for what in ("inverse only", "forward only", "round trip"):
                t0 = time.perf_counter()
                for _ in range(50):
                    if what != "inverse only": pl.sdft(x, out)
                    if what != "forward only": pl.isdft(out, y)
                pl.synchronize(); dt = (time.perf_counter() - t0) / 50
                print(f"n={n} m={m} ch={ch} {what}: {dt*1e3:.3f} ms per iteration ({ch*n*m*16/dt/1e12:.2f} TB/s per pass)", flush=True)
            pl.close(); del out
    if which == "self":
        # self-carried chunks (one launch) against the pre-pass form: asynchronous and synchronous wall time per call
        import os
        M = int(os.environ.get("SDFT_M", "1024"))                 # 1000: the reference test size (mixed-radix FFT in the kernel)
        sizes = [int(a) for a in sys.argv[2:]] or [1024, 4096, 12000, 24000, 48000, 131072, 262144, 1000000]
        for n in sizes:
            x = torch.from_numpy(sine_sweep(n)).cuda()
            out = torch.empty((n, M), dtype=torch.complex128, device="cuda")
            y = torch.empty_like(x)
            for sc in (0, 1):
                pl = SDFT(M, "hann", 1.0, "f32f64"); pl.set_option("self_carry", sc); pl.set_option("self_carry_max", 1 << 30)
                res = {}
                for mode in ("async", "sync"):
                    pl.set_option("async", 1 if mode == "async" else 0)
                    reps = 200 if n <= 48000 else 20
                    for what in ("sdft", "process"):
                        f = (lambda: pl.sdft(x, out)) if what == "sdft" else (lambda: pl.process(x, "identity", out=y))
                        f(); f(); pl.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(reps): f()
                        pl.synchronize(); res[(mode, what)] = (time.perf_counter() - t0) / reps
                b = n * (M * 16 + 4)
                print(f"n={n} self_carry={sc} last_self={pl.get_option('last_self')} chunks={pl.get_option('last_chunks')} len={pl.get_option('last_chunk_len')}: "
                      f"sdft async {res[('async','sdft')]*1e6:.1f} us ({b/res[('async','sdft')]/8e12:.3f} of peak) sync {res[('sync','sdft')]*1e6:.1f} us ({b/res[('sync','sdft')]/8e12:.3f}) | "
                      f"process async {res[('async','process')]*1e6:.1f} us sync {res[('sync','process')]*1e6:.1f} us", flush=True)
                pl.close()
            del out
    if which == "relay":
        # exact carries: relay form against the serial pass, alone (profile events) and inside the call
        for shape in ((262144, 4096, "blackman", "f32f32"), (262144, 1024, "hann", "f32f32"), (1000000, 1024, "hann", "f32f64")):
            n, m, win, combo = shape
            extra = {"carry": 1} if combo == "f32f64" else {}
            for opts in ({"chain": 0}, {}, {"relay_waves": 6}, {"relay_waves": 7}, {"relay_flow": 0}, {"relay_flow": 0, "segments": 4}, {"segments": 1}):
                run(n, m, win, combo, **extra, **opts)
    if which == "mid":
        # calls between a hop and the north star's 48000 samples: chunk length against wall time per call
        for combo in ("f32f64", "f32f32"):
            for n in (1024, 2048, 4096, 12000, 24000, 48000):
                x = torch.from_numpy(sine_sweep(n)).cuda()
                out = torch.empty((n, 1024), dtype=torch.complex128 if combo == "f32f64" else torch.complex64, device="cuda")
                for ch in (0, 32, 64, 96, 128, 192):
                    pl = SDFT(1024, "hann", 1.0, combo); pl.set_option("async", 1); pl.set_option("chunk", ch)
                    pl.sdft(x, out); pl.sdft(x, out); pl.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(50): pl.sdft(x, out)
                    pl.synchronize(); dt = (time.perf_counter() - t0) / 50
                    print(f"{combo} n={n} chunk={ch} -> chunks={pl.get_option('last_chunks')} len={pl.get_option('last_chunk_len')}: {dt*1e6:.1f} us per call ({n*1024*out.element_size()/dt/1e12:.2f} TB/s)", flush=True)
                    pl.close()
    if which == "segs":
        for ch, sg in ((0, 0), (128, 8), (128, 4), (128, 16), (192, 6), (64, 8), (64, 16), (256, 8)):
            run(262144, 4096, "blackman", "f32f32", chunk=ch, segments=sg)
        for ch, sg in ((0, 0), (128, 8), (128, 16), (64, 16)):
            run(262144, 1024, "hann", "f32f32", chunk=ch, segments=sg)
            run(1000000, 1024, carry=1, chunk=ch * 2, segments=sg)
    if which == "syn2":
        # two-slot rows through the fused call, next to the two calls
        run_process(262144, 4096, "blackman", "f32f32")
        run_process(262144, 4096, "blackman", "f32f32", fused_exact=0)
        run(262144, 4096, "blackman", "f32f32")
        run_process(48000, 2048, "hann", "f32f64", channels=64)
        run_process(48000, 2048, "hann", "f32f64", channels=64, fused_exact=1)
        run_process(500000, 2048, "hann", "f32f64")
        run(500000, 2048)
        run_process(48000, 2048, "hann", "f32f32", channels=64)
        run_process(48000, 2048, "hann", "f32f32", channels=64, fused_exact=0)
    if which == "c3f32":
        for sg in (0, 8):
            for ch in (0, 1504, 752):
                run(48000, 2048, "hann", "f32f32", channels=64, segments=sg, chunk=ch)
        run(48000, 2048, "hann", "f32f32", channels=64, chain=2)
    if which == "sizes":
        # partial-sum forms: 2N a power of two (FFT), 2/3/5-smooth (mixed radix), other primes (direct sums)
        for m in (1024, 1000, 960, 1022, 1018, 1009, 500, 509, 2048, 2000, 2042):
            run(1000000 if m <= 1024 else 500000, m)
        for m in (1024, 1000, 1022):
            run(48000, m, reps=20)
    if which == "inv":
        for n in (4096, 12000, 48000, 131072, 262144):
            for rw in (1, 4, 16, 32):
                run(n, 1024, inverse_rows=rw, reps=10)
        for rw in (1, 4, 16):
            run(48000, 1024, "hann", "f32f32", inverse_rows=rw, reps=10)
            run(20000, 4096, "blackman", "f32f32", inverse_rows=rw, reps=10)
    if which == "ns48":
        for rep in range(2):
            for ch in (0, 96, 128, 160, 192, 256, 384):
                run(48000, 1024, chunk=ch, reps=20)
        run(48000, 1024, fft_carry=0, reps=20)
    if which == "chainlen":
        for ch in (192, 256, 384, 768):
            run(262144, 1024, "hann", "f32f32", chain=2, segments=1, chunk=ch)
        for ch in (192, 384):
            for sg in (4, 8):
                run(262144, 4096, "blackman", "f32f32", chain=2, segments=sg, chunk=ch)
    if which == "exact":
        for chain in (0, 2):
            run(262144, 4096, "blackman", "f32f32", chain=chain)
            run(262144, 4096, "blackman", "f32f32", chain=chain, segments=8)
        run(262144, 4096, "blackman", "f32f32", chain=2, segments=16)
        for chain in (0, 2):
            run(48000, 2048, "hann", "f32f32", channels=64, chain=chain)
            run(1000000, 1024, carry=1, chain=chain)
        run(1000000, 1024)
        run(262144, 1024, "hann", "f32f32", chain=0)
        run(262144, 1024, "hann", "f32f32", chain=2)
    if which == "v2":
        for rep in range(2):
            run(1000000, 1024)
            run(1000000, 1024, rows_kernel=0)
        for ch in (488, 976, 1952, 3904, 7808):
            run(1000000, 1024, chunk=ch)
        run(48000, 1024)
        run(48000, 1024, rows_kernel=0)
        for ch in (96, 192, 376):
            run(48000, 1024, chunk=ch)
        run(48000, 1024, channels=64)
        run(48000, 1024, channels=64, rows_kernel=0)
        run(262144, 2048, "blackman", "f32f32")
    if which == "f32":
        run(262144, 4096, "blackman", "f32f32")
        run(262144, 2048, "blackman", "f32f32")
        run(262144, 1024, "hann", "f32f32")
        run(48000, 1024, "hann", "f32f32", channels=64)
        run(1000000, 1024, carry=1)
        run(48000, 1024)
        run(1000000, 1024)
    if which == "v21":
        for rep in range(2):
            run(1000000, 1024)
            run(1000000, 1024)
        run(1000000, 1024, rows_kernel=0)
        run(48000, 1024)
        run(48000, 1024, channels=64)
        run(48000, 1024, channels=64)
        run(262144, 2048, "blackman", "f32f32")
        run(1000000, 1000)
        run(1000000, 1024, "blackman")
    if which == "fft":
        for rep in range(2):
            run(1000000, 1024)
            run(1000000, 1024, fft_carry=0)
        run(48000, 1024)
        run(48000, 1024, fft_carry=0)
        run(48000, 1024, channels=64)
        run(48000, 1024, channels=64, fft_carry=0)
        run(48000, 2048, channels=16)
    if which == "ab":
        for rep in range(3):
            run(1000000, 1024)
            if os.environ.get("SDFT_HIP_LIBRARY") is None:
                run(1000000, 1024, fused=1)
        run(48000, 1024, channels=64)
    if which == "align":
        n, m = 1000000, 1024
        x = torch.from_numpy(sine_sweep(n)).cuda()
        extra = 64 * 1024 * 1024 // 16
        for trial in range(3):
            big = torch.empty(n * m + extra, dtype=torch.complex128, device="cuda")
            print("base address mod 2^30:", hex(big.data_ptr() % (1 << 30)), "mod 2MB:", hex(big.data_ptr() % (1 << 21)), flush=True)
            p = SDFT(m); p.set_option("profile", 1); p.set_option("async", 1)
            for off_bytes in (0, 256, 4096, 65536, 1 << 20, (1 << 21) + 4096, 16 << 20, 0):
                off = off_bytes // 16
                out = big[off:off + n * m].view(n, m)
                for _ in range(2): p.sdft(x, out)
                p.synchronize(); p.profile()
                for _ in range(5): p.sdft(x, out)
                p.synchronize(); pr = p.profile()
                print(f"  offset {off_bytes:>9} B: fwd {pr['forward'][0] / pr['forward'][1]:.3f} ms", flush=True)
            p.close(); del big, out
            junk = torch.empty((trial + 1) * 3_000_000_000 // 16, dtype=torch.complex128, device="cuda")   # perturb the next allocation
    if which == "align2":
        n, m = 1000000, 1024
        x = torch.from_numpy(sine_sweep(n)).cuda()
        slack = 2 << 30
        big = torch.empty((n * m * 16 + slack) // 16, dtype=torch.complex128, device="cuda")
        base = big.data_ptr()
        print("base", hex(base), flush=True)
        p = SDFT(m); p.set_option("profile", 1); p.set_option("async", 1)
        G1 = 1 << 30
        up = (-base) % G1
        for name, off_bytes in (("1GB aligned", up), ("1GB+2MB", up + (2 << 20)), ("1GB+16MB", up + (16 << 20)), ("1GB+256MB", up + (256 << 20)),
                                ("1GB+512MB", up + (512 << 20)), ("1GB+768MB+2MB", up + (770 << 20)), ("base", 0), ("1GB aligned", up)):
            off = off_bytes // 16
            out = big[off:off + n * m].view(n, m)
            for _ in range(2): p.sdft(x, out)
            p.synchronize(); p.profile()
            for _ in range(5): p.sdft(x, out)
            p.synchronize(); pr = p.profile()
            print(f"  {name:>14} addr {hex(out.data_ptr())}: fwd {pr['forward'][0] / pr['forward'][1]:.3f} ms", flush=True)
    if which == "slots":
        for rep in range(2):
            run(262144, 4096, "blackman", "f32f32")
            run(262144, 4096, "blackman", "f32f32", row_slots_max=1)
            run(48000, 2048, channels=16)
            run(48000, 2048, channels=16, row_slots_max=1)
        run(262144, 2048, "hann")
        run(262144, 2048, "hann", row_slots_max=1)
        run(1000000, 1024)
    if which == "hop":
        import time as _t
        for m, hop in ((1000, 100), (1024, 256), (1024, 1)):
            for where in ("device", "host"):
                p = SDFT(m)
                if len(sys.argv) > 2: p.set_option("pointers", 1 if where == "device" else 2)
                if len(sys.argv) > 3 and where == "device": p.set_option("async", 1); p.set_option("profile", 1)
                n = hop * 400
                xh = sine_sweep(n)
                if where == "device":
                    x = torch.from_numpy(xh).cuda(); out = torch.empty((hop, m), dtype=torch.complex128, device="cuda"); y = torch.empty(hop, dtype=torch.float32, device="cuda")
                else:
                    x = xh; out = np.empty((hop, m), dtype=np.complex128); y = np.empty(hop, dtype=np.float32)
                for i in range(0, 20 * hop, hop): p.sdft(x[i:i + hop], out); p.isdft(out, y)
                torch.cuda.synchronize(); t0 = _t.perf_counter()
                for i in range(20 * hop, n, hop): p.sdft(x[i:i + hop], out); p.isdft(out, y)
                torch.cuda.synchronize(); dt = (_t.perf_counter() - t0) / (n // hop - 20)
                print(f"hop streaming m={m} hop={hop} {where} pointers: {dt * 1e6:.1f} us per hop (sdft_n + isdft_n) -> {hop / dt / 1e6:.3f} Msamples/s", flush=True)
                if len(sys.argv) > 3 and where == "device":
                    pr = p.profile(); print("   per-call device ms:", {k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items()}, flush=True)
                p.close()
    if which == "host":
        import time as _t, ctypes as C
        for n, hint in ((1024, 0), (1024, 1), (48000, 1), (64, 1)):
            p = SDFT(1024); p.set_option("async", 1); p.set_option("pointers", hint)
            x = torch.from_numpy(sine_sweep(n)).cuda(); out = torch.empty((n, 1024), dtype=torch.complex128, device="cuda")
            for _ in range(20): p.sdft(x, out)
            p.synchronize()
            xp, op = C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr())
            t0 = _t.perf_counter()
            for _ in range(300): p.api.sdft_n(p._p, n, xp, op)          # raw C-ABI call, returns after enqueueing
            t1 = _t.perf_counter(); p.synchronize(); t2 = _t.perf_counter()
            print(f"n={n} pointers-hint={hint}: host {1e6 * (t1 - t0) / 300:.1f} us per call to return, {1e6 * (t2 - t0) / 300:.1f} us per call incl. drain; chunks={p.get_option('last_chunks')}", flush=True)
            p.close()
    if which == "blocks":
        for rep in range(2):
            for blocks in (256, 512, 768, 1024, 1280, 2048):
                ln = -(-1000000 // blocks); ln = -(-ln // 8) * 8
                run(1000000, 1024, chunk=ln)
    if which == "inv2":
        for rep in range(2):
            for rw in (16, 32, 64):
                run(1000000, 1024, inverse_rows=rw)
            run(1000000, 1024, exact_inverse=0)
        for rw in (16, 32, 64):
            run(262144, 4096, "blackman", "f32f32", inverse_rows=rw)
        for rw in (16, 32, 64):
            run(48000, 1024, channels=64, inverse_rows=rw)
        for rw in (16, 32):
            run(48000, 1000, inverse_rows=rw)
    if which == "inv":
        for rep in range(2):
            run(1000000, 1024)
            run(1000000, 1024, exact_inverse=0)
        run(262144, 4096, "blackman", "f32f32")
        run(262144, 4096, "blackman", "f32f32", exact_inverse=0)
        run(48000, 1024, channels=64)
        run(48000, 1024, channels=64, exact_inverse=0)
        run(48000, 1000)
        run(48000, 1000, exact_inverse=0)
    if which == "ceiling3":
        from sdft_amd import capi
        lib = capi.load()
        nbytes = 1000000 * 1024 * 16
        buf = torch.empty(nbytes // 16, dtype=torch.complex128, device="cuda")
        torch.cuda.synchronize()
        for rep in range(2):
            for ln in (488, 976, 1160, 1952, 2320, 3000, 3904, 4096, 5000, 7800):
                for sync in (0, 8):
                    ms = lib.sdft_hip_store_ceiling(buf.data_ptr(), nbytes, 2, 1024, sync, ln, 5)
                    print(f"rowgroup len={ln} sync={sync} blocks={(1000000 + ln - 1) // ln}: {ms:.3f} ms -> {nbytes / ms / 1e9:.2f} TB/s", flush=True)
            for lanes in (56, 64):
                for ln in (1160, 2320, 3000, 5000):
                    ms = lib.sdft_hip_store_ceiling(buf.data_ptr(), nbytes, 1, 1024, lanes, ln, 5)
                    print(f"tiled lanes={lanes} len={ln}: {ms:.3f} ms -> {nbytes / ms / 1e9:.2f} TB/s", flush=True)
    if which == "ceiling2":
        from sdft_amd import capi
        lib = capi.load()
        nbytes = 1000000 * 1024 * 16
        buf = torch.empty(nbytes // 16, dtype=torch.complex128, device="cuda")
        torch.cuda.synchronize()
        for lanes in (16, 32, 56, 64):
            for ln in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
                ms = lib.sdft_hip_store_ceiling(buf.data_ptr(), nbytes, 1, 1024, lanes, ln, 5)
                tiles = (1024 + lanes - 1) // lanes; chunks = (1000000 + ln - 1) // ln
                print(f"store-only lanes={lanes} len={ln} waves={tiles*chunks}: {ms:.3f} ms -> {nbytes / ms / 1e9:.2f} TB/s", flush=True)
    if which == "ceiling":
        from sdft_amd import capi
        lib = capi.load()
        nbytes = 1000000 * 1024 * 16
        buf = torch.empty(nbytes // 16, dtype=torch.complex128, device="cuda")
        torch.cuda.synchronize()
        for name, args in (("linear", (0, 1024, 64, 1)), ("tiled 64 lanes len1160", (1, 1024, 64, 1160)), ("tiled 56 lanes len1160", (1, 1024, 56, 1160)),
                           ("tiled 62 lanes len1160", (1, 1024, 62, 1160)), ("tiled 64 lanes len4096", (1, 1024, 64, 4096)), ("tiled 64 lanes len256", (1, 1024, 64, 256)),
                           ("linear again", (0, 1024, 64, 1))):
            ms = lib.sdft_hip_store_ceiling(buf.data_ptr(), nbytes, args[0], args[1], args[2], args[3], 10)
            print(f"store ceiling {name}: {ms:.3f} ms -> {nbytes / ms / 1e9:.2f} TB/s", flush=True)
        run(1000000, 1024)
    if which == "sweep":
        run(1000000, 1024)
        run(48000, 1024)
        for kw in ({"interior": 48}, {"interior": 40}, {"interior": 32}, {"target_waves": 8192}, {"target_waves": 32768},
                   {"chunk": 2048}, {"chunk": 4096}):
            run(1000000, 1024, **kw)
        for kw in ({"chunk": 128}, {"chunk": 256}, {"chunk": 512}, {"chunk": 1024}):
            run(48000, 1024, **kw)
        run(48000, 1024, channels=64)
        run(48000, 2048, channels=16)
        import torch
        out = torch.empty((1000000, 1024), dtype=torch.complex128, device="cuda")
        for _ in range(3): out.zero_()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out.zero_()
        e1.record(); torch.cuda.synchronize()
        print("torch zero_ 16.4GB:", e0.elapsed_time(e1) / 10, "ms ->", 16.384e9 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12, "TB/s")
    if which == "base":
        run(48000, 1024)
        run(1000000, 1024)
        for tw in (4096, 8192, 32768, 65536):
            run(1000000, 1024, target_waves=tw)
        for il in (56, 60):
            run(1000000, 1024, interior=il)
        run(262144, 4096, "blackman", "f32f32")
        run(48000, 1024, channels=64)
