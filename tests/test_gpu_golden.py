"""HIP path vs the committed golden vectors (generated from the genuine reference build by
tests/golden/make_golden.py).  These run on the GPU box, where /root/reference does not exist."""

import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {"f64": 1e-6, "f32": 1e-4}


def load(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def rel_err(a, b):
    scale = float(np.abs(b).max())
    return float(np.abs(np.asarray(a) - np.asarray(b)).max()) / (scale if scale else 1.0)


def digest_of(d, block=4096):
    """per-row checksums (sum re, sum im, sum |.|^2, sum (k+1) re), computed in row blocks to keep the
    temporaries small"""
    k = np.arange(1, d.shape[1] + 1, dtype=np.float64)
    out = np.empty((d.shape[0], 4), dtype=np.float64)
    for i in range(0, d.shape[0], block):
        re = d[i:i + block].real.astype(np.float64); im = d[i:i + block].imag.astype(np.float64)
        out[i:i + block] = np.stack([re.sum(1), im.sum(1), (re * re + im * im).sum(1), (re * k).sum(1)], axis=1)
    return out


def test_tiny_golden_cases_bit_exact():
    from sdft_amd.sdft import SDFT
    z = load("tiny_cases.npz")
    keys = sorted({k.rsplit("/", 1)[0] for k in z})
    for key in keys:
        combo, window, lat, m = key.split("/")
        with SDFT(int(m), window, int(lat) / 100.0, combo) as p:
            d = p.sdft(z[key + "/x"])
            assert np.array_equal(d, z[key + "/d"]), key
            assert rel_err(p.isdft(d), z[key + "/y"]) <= TOL[combo[3:]], key


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "mid_*.npz"))))
def test_mid_golden_streaming_bit_exact(path):
    """Ragged hop-wise streaming (hop 37 < 512 samples: serial path) is bit-identical to the reference."""
    from sdft_amd.sdft import SDFT
    g = load(os.path.basename(path))
    m, window, latency, combo, hop = int(g["dftsize"]), str(g["window"]), float(g["latency"]), str(g["combo"]), int(g["hop"])
    x = g["x"]
    with SDFT(m, window, latency, combo) as p:
        d = np.concatenate([p.sdft(x[i:i + hop]) for i in range(0, x.size, hop)])
        y = p.isdft(d)
    assert np.array_equal(d[g["rows_idx"]], g["rows"])
    assert np.array_equal(digest_of(d), g["digest"])
    assert rel_err(y, g["y"]) <= TOL[combo[3:]]


@pytest.mark.parametrize("name,chunk", [("cfg1_sweep48000_m1024_hann_f32f64.npz", 0),
                                        ("cfg4_sweep12000_m2048_hann_f32f64.npz", 0),
                                        ("cfg3_sweep12000_m4096_blackman_f32f32.npz", 0),
                                        ("cfg3_sweep12000_m4096_blackman_f32f32.npz", 750)])
def test_baseline_shapes_against_golden(name, chunk):
    """BASELINE config shapes with the default (time-chunked) launch geometry, device pointers."""
    import torch
    from sdft_amd.sdft import SDFT
    g = load(name)
    m, window, latency, combo = int(g["dftsize"]), str(g["window"]), float(g["latency"]), str(g["combo"])
    tol = TOL[combo[3:]]
    with SDFT(m, window, latency, combo) as p:
        if chunk:
            p.set_option("chunk", chunk)
        d = p.sdft(torch.from_numpy(g["x"]).cuda())
        y = p.isdft(d).cpu().numpy()
        d = d.cpu().numpy()
        assert p.get_option("last_chunks") > 1
    if combo.endswith("f32"):
        assert np.array_equal(d[g["rows_idx"]], g["rows"])            # exact-carry mode
        assert np.array_equal(digest_of(d), g["digest"])
    else:
        assert rel_err(d[g["rows_idx"]], g["rows"]) <= 1e-11
        got, want = digest_of(d), g["digest"]                          # checksums cancel: scale per column
        assert (np.abs(got - want).max(axis=0) <= 1e-9 * np.abs(want).max(axis=0)).all()
    assert rel_err(y, g["y"]) <= tol


def test_reference_test_wav_pattern():
    """The reference's own end-to-end test (test/main.sh:3-6, test/test.c:69-83): test.wav, m=1000,
    hop=100, Hann, latency 1 -- first DFT row of each hop and the synthesised waveform."""
    from sdft_amd.sdft import SDFT
    g = load("testwav_m1000_hop100_hann_f32f64.npz")
    hop, x = int(g["hop"]), g["x"]
    firsts, ys = [], []
    with SDFT(1000, "hann", 1.0, "f32f64") as p:
        for i in range(0, x.size, hop):
            d = p.sdft(x[i:i + hop])
            firsts.append(d[0]); ys.append(p.isdft(d))
    assert np.array_equal(np.stack(firsts), g["hop_first_rows"])
    assert rel_err(np.concatenate(ys), g["y"]) <= 1e-6
    # the reference's own acceptance rule (test/main.py:70,78): np.allclose defaults
    assert np.allclose(np.concatenate(ys), g["y"]) and np.allclose(np.stack(firsts), g["hop_first_rows"])


def test_reference_test_wav_pattern_with_the_resident_kernel():
    """Round 6, option "resident" = 1: the reference driver's loop -- sdft_sdft_n + sdft_isdft_n per hop, synchronous, device
    pointers (test/test.c:69-83) -- served by ONE kernel that stays on the chip: a doorbell per call instead of a launch.  Same device
    functions as the launches: every row of every hop and every sample bit-identical to the plan without it and to the reference's golden
    vectors.  The kernel leaves by itself after 200 us without a call (a pause in the loop: it is started again, nothing changes); a plain
    blocking hipMemcpy right after the loop returns within that time-out; every other entry point retires it first."""
    import ctypes as C
    import time
    import torch
    from sdft_amd import capi
    from sdft_amd.sdft import SDFT
    g = load("testwav_m1000_hop100_hann_f32f64.npz")
    hop, x = int(g["hop"]), g["x"]
    hops = x.size // hop
    xd = torch.from_numpy(x).cuda()
    res = {}
    for resident in (0, 1):
        with SDFT(1000, "hann", 1.0, "f32f64") as p:
            p.set_option("resident", resident)
            d = torch.empty((hop, 1000), dtype=torch.complex128, device="cuda")
            y = torch.empty(x.size, dtype=torch.float32, device="cuda")
            rows = []
            for i in range(hops):
                p.sdft(xd[i * hop:(i + 1) * hop], d)
                if i % 7 == 0:
                    rows.append(d.cpu().numpy().copy())          # (a blocking copy on the null stream in the middle of the loop: waits for the kernel to leave)
                else:
                    rows.append(None)
                p.isdft(d, y[i * hop:(i + 1) * hop])
                if i == hops // 2:
                    time.sleep(0.01)                              # a pause longer than the idle time-out: the kernel has left and is started again
            if resident:
                assert p.get_option("resident_calls") >= 2 * hops - 2 * (hops // 7 + 2), (p.get_option("resident_calls"), hops)
                assert 2 <= p.get_option("resident_launches") <= hops // 7 + 4, p.get_option("resident_launches")
                # a plain hipMemcpy right after the loop: the plan's stream is a blocking stream, the copy waits for the kernel to leave
                hip = C.CDLL(capi.hip_runtime)
                hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                p.sdft(xd[:hop], d); p.isdft(d, y[:hop])          # (the kernel is alive: two resident calls)
                assert p.get_option("resident_alive") == 1
                host_t = torch.empty(hop, dtype=torch.float32).pin_memory()  # (pinned: not the runtime's path for pageable memory)
                host = host_t.numpy()
                t0 = time.perf_counter()
                assert hip.hipMemcpy(C.c_void_p(host_t.data_ptr()), C.c_void_p(y.data_ptr()), C.c_size_t(host.nbytes), 2) == 0
                waited = time.perf_counter() - t0
                assert waited < 5e-3, waited                       # (200 us of idle time-out and the copy itself; 5 ms leaves room for a busy box)
                # undo those two calls' effect on the comparison below: run the other plan's loop the same way
            else:
                assert p.get_option("resident_calls") == 0
                p.sdft(xd[:hop], d); p.isdft(d, y[:hop])
            # every other entry point retires the kernel first: the state, a long call, a host-pointer call
            st = p.state()
            assert p.get_option("resident_alive") == 0
            tail = p.sdft(x[:37])                                  # host pointers: the ordinary route
            res[resident] = (rows, y.cpu().numpy(), st, tail)
    rows0, y0, st0, tail0 = res[0]
    rows1, y1, st1, tail1 = res[1]
    for a, b in zip(rows0, rows1):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    assert np.array_equal(y0, y1) and np.array_equal(tail0, tail1)
    for a, b in zip(st0[:3], st1[:3]):
        assert np.array_equal(a, b)
    assert st0[3] == st1[3]
    assert np.array_equal(np.stack([r[0] for r in rows1 if r is not None]), g["hop_first_rows"][::7][:len([r for r in rows1 if r is not None])])
    assert rel_err(y1[hop:], g["y"][hop:y1.size]) <= 1e-6


def test_resident_kernel_calls_that_race_with_its_idle_time_out():
    """The hop loop with pauses around the resident kernel's idle time-out (200 us): calls find the kernel gone (it is started again) or race with its leaving
    (the exit word says the call was not served: rung again on a fresh launch).  Every compared hop and the final state are bit-identical to a plan that launches."""
    import random
    import time
    import torch
    from sdft_amd.sdft import SDFT
    from sdft_amd.signals import noise
    random.seed(11)
    m, hop, hops = 1000, 100, 600
    x = torch.from_numpy(noise(hop * hops, seed=3)).cuda()
    pauses = [0, 0, 50e-6, 150e-6, 190e-6, 200e-6, 210e-6, 230e-6, 300e-6, 1e-3]
    with SDFT(m, "hann", 1.0, "f32f64") as pa, SDFT(m, "hann", 1.0, "f32f64") as pb:
        pa.set_option("resident", 1)
        da = torch.empty((hop, m), dtype=torch.complex128, device="cuda"); db = torch.empty_like(da)
        ya = torch.empty(hop, dtype=torch.float32, device="cuda"); yb = torch.empty_like(ya)
        for i in range(hops):
            seg = x[i * hop:(i + 1) * hop]
            pa.sdft(seg, da)
            t = time.perf_counter(); p = random.choice(pauses)
            while time.perf_counter() - t < p:
                pass
            pa.isdft(da, ya)
            if i % 3 == 0:
                t = time.perf_counter(); p = random.choice(pauses)
                while time.perf_counter() - t < p:
                    pass
            pb.sdft(seg, db)
            if i % 25 == 0:
                pb.isdft(db, yb)
                assert torch.equal(da, db) and torch.equal(ya, yb), i
        assert pa.get_option("resident") == 1 and pa.get_option("resident_calls") >= 2 * hops - 60 and pa.get_option("resident_launches") > 50
        assert pa.api.last_warning() is None
        sa, sb = pa.state(), pb.state()
        assert sa[3] == sb[3] and all(np.array_equal(a, b) for a, b in zip(sa[:3], sb[:3]))


@pytest.mark.parametrize("combo", ["f32f64", "f64f64", "f32f32"])
def test_single_sample_calls_through_the_resident_kernel(combo):
    """sdft_sdft / sdft_isdft (sdft.h:562, :635), one sample per call, with option "resident" = 1: the sample rides in the doorbell line, the result comes back
    through the plan's pinned scratch -- no launch per sample.  Bit-identical to the launches and to the reference, across the roll-over (t = 2N - 1)."""
    import ctypes as C
    import torch
    from oracle import oracle as O
    from sdft_amd.sdft import SDFT
    from sdft_amd.signals import noise
    td, fd, fdx = O.combo_types(combo)
    m, n = 96, 2 * 96 + 37
    x = noise(n, seed=4, dtype=td)
    ref = O.best(m, "hamming", 1.0, combo)
    want = ref.sdft(x); ywant = ref.isdft(want)
    res = {}
    for resident in (0, 1):
        with SDFT(m, "hamming", 1.0, combo) as p:
            p.set_option("resident", resident)
            row = torch.empty(m, dtype=torch.complex128 if fd == np.float64 else torch.complex64, device="cuda")
            rows, ys = [], []
            for i in range(n):
                p.api.sdft(p._p, td(x[i]).item(), C.c_void_p(row.data_ptr()))
                ys.append(p.api.isdft(p._p, C.c_void_p(row.data_ptr())))
                if i % 16 == 0 or i >= 2 * m - 2:
                    rows.append((i, row.cpu().numpy().copy()))
            if resident:
                assert p.get_option("resident_calls") >= 2 * n - 4 * len(rows) - 4, (p.get_option("resident_calls"), n)
            res[resident] = (rows, np.array(ys, dtype=td), p.state())
    for (i, a), (j, b) in zip(res[0][0], res[1][0]):
        assert i == j and np.array_equal(a, b) and np.array_equal(a, want[i]), i
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[1][1], ywant)
    assert res[0][2][3] == res[1][2][3] and all(np.array_equal(a, b) for a, b in zip(res[0][2][:3], res[1][2][:3]))
