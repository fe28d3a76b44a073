"""Fused analysis -> spectral operation -> synthesis (sdft_hip_process_n) against what
a host of the reference computes with sdft_sdft_n, its own loop over the matrix, and sdft_isdft_n
(README.md:42-47 of the reference): the oracle's analysis, the operation in numpy, the oracle's synthesis."""

import numpy as np
import pytest

from oracle import oracle as O
from sdft_amd.signals import noise, sine_sweep

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-6, "f32": 1e-4}


def cgain_of(gain, fdx):
    """The complex factors the "cgain" cases use: the test's real gains with a phase that turns with the bin."""
    g = np.asarray(gain, dtype=np.float64)
    return (g * np.exp(1j * 0.37 * np.arange(g.size))).astype(fdx)


def make(dftsize, window="hann", latency=1.0, combo="f32f64", channels=1, **opts):
    from sdft_amd.sdft import SDFT
    p = SDFT(dftsize, window, latency, combo, channels)
    for k, v in opts.items():
        p.set_option(k, v)
    plain = p.process

    def process(x, op="identity", gain=None, shift=0, **kw):          # "cgain": complex factors derived from the real ones
        if op == "cgain":
            gain = cgain_of(gain, p.fdx)
        return plain(x, op, gain=gain, shift=shift, **kw)
    p.process = process
    return p


def apply_op(d, op, gain, shift):
    """What the host does to the (n, N) matrix between the two reference calls."""
    if op == "gain":
        return (d * gain[None, :].astype(d.real.dtype)).astype(d.dtype)     # complex * real, per component
    if op == "cgain":
        g = cgain_of(gain, d.dtype)
        out = np.empty_like(d)                                               # (ac - bd) + (ad + bc)i, every operation rounded
        out.real = d.real * g.real[None, :] - d.imag * g.imag[None, :]
        out.imag = d.real * g.imag[None, :] + d.imag * g.real[None, :]
        return out
    if op == "shift":
        out = np.zeros_like(d)
        n = d.shape[1]
        if shift >= 0:
            out[:, shift:] = d[:, :n - shift] if shift < n else 0
        else:
            out[:, :n + shift] = d[:, -shift:]
        return out
    return d


def reference(ref, x, op, gain, shift):
    d = apply_op(ref.sdft(x), op, gain, shift)
    return ref.isdft(d), d


def rel(a, b):
    s = float(np.abs(b).max())
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max()) / (s if s else 1.0)


OPS = [("identity", 0), ("gain", 0), ("shift", 3), ("shift", -5), ("cgain", 0)]


@pytest.mark.parametrize("op,shift", OPS)
@pytest.mark.parametrize("latency", [1.0, 0.5])
def test_fused_double_fast_path(op, shift, latency):
    """FD double, chunk-parallel analysis (not bit-exact by design): fused kernel with the tree sum,
    and with the reference's summation order on request; state carries over to the next call."""
    import torch
    m, n = 1024, 6000
    x = sine_sweep(n)
    gain = np.linspace(0.0, 2.0, m)
    ref = O.best(m, "hann", latency, "f32f64")
    want, wd = reference(ref, x, op, gain, shift)
    x2 = noise(700, seed=3)
    want2, _ = reference(ref, x2, op, gain, shift)
    for fused_exact in (-1, 1):
        with make(m, "hann", latency, "f32f64", fused_exact=fused_exact) as p:
            xd = torch.from_numpy(x).cuda()
            y = p.process(xd, op, gain=gain, shift=shift)
            assert p.get_option("last_process_path") == 1 and p.get_option("last_chunks") > 1
            assert p.get_option("last_fused_exact") == (1 if fused_exact == 1 else 0)
            assert rel(y.cpu().numpy(), want) <= TOL["f64"], (op, shift, latency, fused_exact)
            if op != "shift":                                    # copy of the processed spectrum
                p.reset()
                dd = torch.empty((n, m), dtype=torch.complex128, device="cuda")
                y = p.process(xd, op, gain=gain, dfts=dd)
                assert rel(y.cpu().numpy(), want) <= TOL["f64"]
                assert float(np.abs(dd.cpu().numpy() - wd).max()) <= 1e-11 * float(np.abs(wd).max())
            y2 = p.process(torch.from_numpy(x2).cuda(), op, gain=gain, shift=shift)      # continues the stream
            assert rel(y2.cpu().numpy(), want2) <= TOL["f64"]


@pytest.mark.parametrize("combo,opts", [("f32f32", {"fused_exact": 1}), ("f64f32", {"fused_exact": 1}),
                                        ("f32f64", {"carry": 1}), ("f64f64", {"carry": 1})])
@pytest.mark.parametrize("window", ["hann", "blackman", "boxcar"])
def test_fused_exact_modes_are_bit_identical(combo, opts, window):
    """Exact carries (FD float with fused_exact = 1, FD double with carry = 1): the fused kernel walks the bins in
    the reference's order, so the samples equal the two reference calls bit for bit -- for every operation,
    both synthesis branches, rows that do not fill the last wave, batched channels."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    for m, n, latency in ((256, 5000, 1.0), (1000, 3000, 0.5), (72, 1500, 1.0)):
        gain = np.cos(np.arange(m) * 0.1).astype(fd)
        ch = 2
        xb = np.stack([noise(n, seed=7 + c, dtype=td) for c in range(ch)])
        for op, shift in OPS:
            with make(m, window, latency, combo, ch, **opts) as p:
                y = p.process(torch.from_numpy(xb).cuda(), op, gain=gain, shift=shift).cpu().numpy()
                assert p.get_option("last_process_path") == 1 and p.get_option("last_fused_exact") == 1
                for c in range(ch):
                    want, _ = reference(O.best(m, window, latency, combo), xb[c], op, gain, shift)
                    assert np.array_equal(y[c], want), (combo, window, m, op, shift, c, rel(y[c], want))


@pytest.mark.parametrize("combo,opts,m,window", [("f32f32", {}, 4096, "blackman"), ("f32f32", {}, 2100, "hann"),
                                                  ("f64f32", {}, 3000, "hamming"), ("f32f64", {"carry": 1}, 2048, "hann"),
                                                  ("f64f64", {"carry": 1}, 1100, "blackman")])
def test_fused_two_slot_rows_bit_identical(combo, opts, m, window):
    """Rows of two slots per lane (1024 < N <= 2048 at FD double, 2048 < N <= 4096 at FD float; configs[2]
    is N = 4096 Blackman FD float): the fused kernel takes four samples per lockstep group; exact carries
    and the reference's summation order give the two reference calls bit for bit."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    n, ch = 2500, 2
    gain = np.cos(np.arange(m) * 0.05).astype(fd)
    xb = np.stack([noise(n, seed=11 + c, dtype=td) for c in range(ch)])
    for latency, (op, shift) in ((1.0, OPS[0]), (0.5, OPS[1]), (1.0, OPS[2]), (0.5, OPS[3]), (1.0, OPS[4])):
        want = [reference(O.best(m, window, latency, combo), xb[c], op, gain, shift)[0] for c in range(ch)]
        # FD float: fused_exact = 1 takes the two passes for these shapes (the ordered walk over 4096 bins costs
        # more than the pass it saves); fused_exact = 2 insists on the fused kernel -- same bits either way
        for fused_exact in ((1, 2) if combo[3:] == "f32" else (-1,)):
            with make(m, window, latency, combo, ch, fused_exact=fused_exact, **opts) as p:
                y = p.process(torch.from_numpy(xb).cuda(), op, gain=gain, shift=shift).cpu().numpy()
                fused = not (combo[3:] == "f32" and fused_exact == 1)
                assert p.get_option("last_process_path") == (1 if fused else 3)
                assert not fused or p.get_option("last_fused_exact") == 1
                assert p.get_option("last_chunks") > 1
                for c in range(ch):
                    assert np.array_equal(y[c], want[c]), (combo, window, m, op, shift, c, fused_exact, rel(y[c], want[c]))


@pytest.mark.parametrize("combo,m", [("f32f64", 2048), ("f32f64", 1500), ("f32f32", 4096)])
def test_fused_two_slot_rows_tree_sum(combo, m):
    """Two-slot rows with the wave-parallel sum over bins (the default of the chunk-parallel FD double path;
    on request for FD float), with and without a copy of the processed spectrum."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    n = 4000
    x = sine_sweep(n, dtype=td)
    gain = np.linspace(0.5, 1.5, m).astype(fd)
    cdt = torch.complex128 if combo[3:] == "f64" else torch.complex64
    for op, shift in OPS:
        want, wd = reference(O.best(m, "hann", 1.0, combo), x, op, gain, shift)
        with make(m, "hann", 1.0, combo, fused_exact=0) as p:
            xd = torch.from_numpy(x).cuda()
            y = p.process(xd, op, gain=gain, shift=shift)
            assert p.get_option("last_process_path") == 1 and p.get_option("last_fused_exact") == 0
            assert rel(y.cpu().numpy(), want) <= TOL[combo[3:]], (combo, m, op, shift)
            if op != "shift":
                p.reset()
                dd = torch.empty((n, m), dtype=cdt, device="cuda")
                y = p.process(xd, op, gain=gain, dfts=dd)
                assert rel(y.cpu().numpy(), want) <= TOL[combo[3:]]
                tol = 1e-11 if combo[3:] == "f64" else 0.0          # FD float analysis is bit-identical
                assert float(np.abs(dd.cpu().numpy() - wd).max()) <= tol * float(np.abs(wd).max())


FOLDED_SHAPES = [("f32f64", 1024), ("f32f64", 1000), ("f32f64", 72), ("f32f64", 1500), ("f64f64", 2048),
                 ("f32f64", 4096), ("f32f64", 2500), ("f32f32", 1000), ("f32f32", 3000), ("f64f32", 4096)]


# every window on the rows of up to 1500 bins; the long rows (seconds of oracle each) take a 3-tap and the 5-tap window
@pytest.mark.parametrize("window,combo,m", [(w, c, m) for c, m in FOLDED_SHAPES for w in ("hann", "hamming", "blackman", "boxcar")
                                            if m <= 1500 or w in ("hann", "blackman")])
def test_folded_form_matches_reference(window, combo, m):
    """The tree-sum flavour of the fused call folds window, operation and synthesis into per-bin coefficients
    (process_rows_kernel): every window (3 and 5 taps, mirror images at both ends of the spectrum), every
    operation, both synthesis branches, one / two / four bins per lane, a roll-over inside the call, the
    stream continued by a second call -- against the two reference calls within the path's bar; with exact
    carries (FD float) the stream state stays bit-identical."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    n = 2 * m + 700
    x = sine_sweep(n, dtype=td) + noise(n, seed=5, dtype=td) * td(0.1)
    x2 = noise(900, seed=6, dtype=td)
    gain = (1.0 + 0.5 * np.sin(np.arange(m) * 0.37)).astype(fd)
    tol = TOL[combo[3:]]
    analysed = None
    for latency, (op, shift) in ((1.0, OPS[0]), (1.0, OPS[1]), (0.5, OPS[1]), (1.0, OPS[2]), (0.5, OPS[3]), (0.5, OPS[0]), (1.0, OPS[4]), (0.5, OPS[4])):
        # (the analysis does not depend on the latency or the operation: the oracle runs it once per test)
        if analysed is None:
            ana = O.best(m, window, 1.0, combo)
            analysed = (ana.sdft(x), ana.sdft(x2))
        ref = O.best(m, window, latency, combo)
        want = ref.isdft(apply_op(analysed[0], op, gain, shift))
        want2 = ref.isdft(apply_op(analysed[1], op, gain, shift))
        for fold in ((1, 0) if m <= (2048 if combo[3:] == "f64" else 4096) else (1,)):    # longer rows: folded form only
            with make(m, window, latency, combo, fused_exact=0, fold=fold) as p:
                y = p.process(torch.from_numpy(x).cuda(), op, gain=gain, shift=shift).cpu().numpy()
                assert p.get_option("last_process_path") == 1 and p.get_option("last_fused_exact") == 0
                assert p.get_option("last_fused_fold") == fold
                assert rel(y, want) <= tol, (combo, window, m, latency, op, shift, fold, rel(y, want))
                y2 = p.process(torch.from_numpy(x2).cuda(), op, gain=gain, shift=shift).cpu().numpy()
                assert rel(y2, want2) <= tol, (combo, window, m, latency, op, shift, fold, "second call")
                if combo[3:] == "f32" and fold == 1 and op == "identity":
                    ref2 = O.best(m, window, latency, combo)
                    ref2.sdft(x); ref2.sdft(x2)
                    x3 = noise(300, seed=8, dtype=td)
                    assert np.array_equal(p.sdft(x3), ref2.sdft(x3))           # exact carries: the state is the reference's


def test_folded_form_batched_channels_and_host_pointers():
    """Folded form on a batched plan (channels on the grid), host-pointer input / output staged by the library."""
    m, n, ch = 1024, 5000, 3
    xb = np.stack([noise(n, seed=20 + c) for c in range(ch)])
    gain = np.linspace(2.0, 0.0, m)
    with make(m, "hann", 1.0, "f32f64", ch) as p:
        y = p.process(xb, "gain", gain=gain)
        assert p.get_option("last_fused_fold") == 1
        for c in range(ch):
            want, _ = reference(O.best(m, "hann", 1.0, "f32f64"), xb[c], "gain", gain, 0)
            assert rel(y[c], want) <= TOL["f64"], c


@pytest.mark.parametrize("combo", O.COMBOS)
def test_hop_sized_calls_and_host_pointers(combo):
    """Calls of one time chunk (the reference's hop loop, test/test.c:69-83), host and device pointers.
    fused_exact = 1: hop kernel + one-wave-per-row synthesis with the operation applied on the way in,
    bit-identical.  Default: differences + the folded kernel, within the bar, and the stream state stays the
    reference's (the next analysis call is bit-identical)."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    m, hop = 1000, 100
    x = sine_sweep(12 * hop, dtype=td)
    gain = (1.0 / (1.0 + np.arange(m) / 100.0)).astype(fd)
    for op, shift in OPS:
        for fused_exact in (1, -1):
            ref = O.best(m, "hann", 1.0, combo)
            with make(m, "hann", 1.0, combo, fused_exact=fused_exact) as p:
                gots, wants = [], []
                for i in range(0, x.size, hop):
                    want, _ = reference(ref, x[i:i + hop], op, gain, shift)
                    seg = x[i:i + hop]
                    got = p.process(seg, op, gain=gain, shift=shift) if (i // hop) % 2 else \
                        p.process(torch.from_numpy(seg).cuda(), op, gain=gain, shift=shift).cpu().numpy()
                    if fused_exact == 1:
                        assert p.get_option("last_process_path") == 2
                        assert np.array_equal(got, want), (combo, op, shift, i)
                    else:
                        assert p.get_option("last_process_path") == 1 and p.get_option("last_fused_fold") == 1
                        assert p.get_option("last_chunks") == 1 and p.get_option("last_hop_pipe") == 1
                    gots.append(got); wants.append(want)
                # (the first hops of a stream are cancellation residue four orders below the signal: the bar is
                # relative to the stream, not to one hop)
                assert rel(np.concatenate(gots), np.concatenate(wants)) <= TOL[combo[3:]], (combo, op, shift, fused_exact)
                x3 = noise(77, seed=9, dtype=td)
                assert np.array_equal(p.sdft(x3), ref.sdft(x3)), (combo, op, fused_exact)      # state untouched by the flavour


def test_shapes_outside_the_fused_kernel_take_the_two_pass_path():
    """Rows beyond every fused kernel (m = 5000), rows the folded form covers but is switched off for (m = 4096 at FD
    double with fold = 0) and tiny rows: analysis + synthesis through the bounded workspace, same results (m = 3000
    at FD float is a fused shape)."""
    import torch
    for m, n, combo in ((4096, 3000, "f32f64"), (5000, 2500, "f32f64"), (5, 900, "f32f32"), (3000, 2000, "f32f32")):
        td, fd, fdx = O.combo_types(combo)
        x = noise(n, seed=1, dtype=td)
        gain = np.linspace(1.0, 0.0, m).astype(fd)
        for op, shift in (("gain", 0), ("shift", 2)):
            want, _ = reference(O.best(m, "hamming", 1.0, combo), x, op, gain, shift)
            with make(m, "hamming", 1.0, combo, stage_bytes=1 << 22, fold=0 if m == 4096 else 1) as p:
                y = p.process(torch.from_numpy(x).cuda(), op, gain=gain, shift=shift).cpu().numpy()
                assert p.get_option("last_process_path") == (1 if m == 3000 else 3)
                assert rel(y, want) <= TOL[combo[3:]], (m, op)


def test_process_argument_errors():
    import ctypes as C
    import torch
    from sdft_amd.capi import Api
    api = Api("f32f64")
    plan = api.alloc(64)
    x = torch.zeros(600, dtype=torch.float32, device="cuda")
    y = torch.full((600,), 7.0, dtype=torch.float32, device="cuda")
    d = torch.zeros((600, 64), dtype=torch.complex128, device="cuda")
    s = C.c_long(1)
    for op, params, dfts in ((9, None, None), (1, None, None), (2, C.cast(C.byref(s), C.c_void_p), C.c_void_p(d.data_ptr()))):
        assert api.process_n(plan, 600, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), op, params, dfts) == -1
        assert api.last_error()
        api.lib.sdft_hip_clear_error()
    assert float(y.min()) == 7.0                                  # outputs untouched
    assert api.process_n(plan, 0, None, None, 0, None, None) == 0
    api.free(plan)


# ---------------------------------------------------------------------------------------------
# round 3: gains that change with time, and operations that are not linear in the spectrum
# ---------------------------------------------------------------------------------------------
def rows_reference(ref, x, gains, hop, complex_gain):
    d = ref.sdft(x)
    r = np.minimum(np.arange(d.shape[0]) // hop, gains.shape[0] - 1)
    out = np.empty_like(d)
    if complex_gain:
        g = gains[r]
        out.real = d.real * g.real - d.imag * g.imag
        out.imag = d.real * g.imag + d.imag * g.real
    else:
        out = (d * gains[r].astype(d.real.dtype)).astype(d.dtype)
    return ref.isdft(out), out


@pytest.mark.parametrize("combo,m,n,hop", [("f32f64", 1024, 12000, 512), ("f32f64", 256, 9000, 100), ("f32f32", 512, 8000, 1000),
                                           ("f64f64", 128, 6000, 37), ("f32f64", 2048, 9000, 4096), ("f32f64", 3000, 5000, 1024)])
@pytest.mark.parametrize("op", ["gain_rows", "cgain_rows"])
def test_gains_that_change_with_time(combo, m, n, hop, op):
    """A host of the reference that recomputes its mask every hop (README.md:42-47 leaves the loop over the matrix to it):
    gains[rows][N], row r for the samples [r*hop, (r+1)*hop), the last row for the rest -- through the folded kernel
    (coefficient vectors reloaded at the row changes), the reference's summation order, the copy of the spectrum,
    the two-pass path, a call of one time chunk, a second call (rows count from the start of each call)."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    rng = np.random.default_rng(7)
    rows = (n + hop - 1) // hop - 1                                  # one row short: the last row serves the tail
    gains = rng.uniform(0.2, 1.5, size=(max(rows, 1), m)).astype(fd)
    if op == "cgain_rows":
        gains = (gains * np.exp(1j * rng.uniform(-1, 1, size=gains.shape))).astype(fdx)
    x = noise(n, seed=3, dtype=td)
    x2 = noise(300, seed=4, dtype=td)
    tol = TOL[combo[3:]]
    for opts in ({}, {"fused_exact": 1}, {"fold": 0}, {"rows_kernel": 0, "fold": 0}):
        if m > 2048 and opts.get("fused_exact"):
            continue
        ref = O.best(m, "hann", 1.0, combo)
        want, dwant = rows_reference(ref, x, gains, hop, op == "cgain_rows")
        with make(m, "hann", 1.0, combo, **opts) as p:
            got = p.process(torch.from_numpy(x).cuda(), op, gain=torch.from_numpy(gains).cuda(), hop=hop).cpu().numpy()
            assert rel_err(got, want) <= tol, (combo, m, hop, opts, rel_err(got, want))
            if opts.get("fused_exact") and combo.endswith("f32"):
                assert np.array_equal(got, want)
            # a call of one time chunk with three rows of 100 samples
            want2, _ = rows_reference(ref, x2, gains[:3], 100, op == "cgain_rows")
            got2 = p.process(x2, op, gain=gains[:3], hop=100)
            assert rel_err(got2, want2) <= tol, (combo, m, hop, opts, "hop-sized", rel_err(got2, want2))
    # copy of the processed spectrum
    if m <= 2048:
        ref = O.best(m, "hann", 1.0, combo)
        want, dwant = rows_reference(ref, x, gains, hop, op == "cgain_rows")
        with make(m, "hann", 1.0, combo) as p:
            dd = torch.empty((n, m), dtype=getattr(torch, np.dtype(fdx).name), device="cuda")
            got = p.process(torch.from_numpy(x).cuda(), op, gain=gains, hop=hop, dfts=dd).cpu().numpy()
            assert rel_err(got, want) <= tol and rel_err(dd.cpu().numpy(), dwant) <= tol


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    scale = float(np.abs(b).max())
    return float(np.abs(a - b).max()) / scale if scale else float(np.abs(a).max())


def nonlinear_reference(ref, x, op, p0, p1):
    d = ref.sdft(x)
    fd = d.real.dtype
    mag2 = d.real * d.real + d.imag * d.imag                      # FD arithmetic, as the kernels do it
    if op == "gate":
        thr2 = fd.type(p0) * fd.type(p0)
        f = np.where(mag2 < thr2, fd.type(p1), fd.type(1))
        out = (d * f).astype(d.dtype)
        out = np.where(mag2 < thr2, out, d)
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            f = fd.type(p1) * np.power(mag2, (fd.type(p0) - fd.type(1)) * fd.type(0.5))
        f = np.where(mag2 > 0, f, 0).astype(fd)
        out = (d * f).astype(d.dtype)
    return ref.isdft(out), out


@pytest.mark.parametrize("combo,m,op,p0,p1", [
    ("f32f64", 1024, "gate", 0.02, 0.0), ("f32f64", 1024, "power", 0.6, 1.3), ("f32f32", 512, "gate", 0.05, 0.25), ("f32f32", 512, "power", 1.5, 0.8),
    ("f32f64", 2048, "gate", 0.05, 0.25), ("f32f64", 2048, "power", 1.5, 0.8), ("f64f64", 100, "gate", 0.02, 0.0), ("f64f64", 100, "power", 0.6, 1.3),
    ("f32f32", 4096, "gate", 0.02, 0.0), ("f32f32", 4096, "power", 1.5, 0.8), ("f32f64", 2500, "gate", 0.05, 0.25), ("f32f64", 2500, "power", 0.6, 1.3)])
def test_operations_that_are_not_linear(combo, m, op, p0, p1):
    """Spectral gate (|X| < threshold -> X * floor) and magnitude power law (|X'| = scale * |X|^p, phase kept) on the
    windowed spectrum inside the row-group kernel; two passes where the row does not fit; copy of the spectrum."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    n = 8000 if m <= 2048 else 6000
    x = (noise(n, seed=11, dtype=td) * 0.5 + sine_sweep(n, dtype=td) * 0.5).astype(td)
    tol = TOL[combo[3:]]
    kw = dict(threshold=p0, floor=p1) if op == "gate" else dict(exponent=p0, scale=p1)
    for opts in ({}, {"fused_exact": 1}):
        ref = O.best(m, "hamming", 1.0, combo)
        want, dwant = nonlinear_reference(ref, x, op, p0, p1)
        with make(m, "hamming", 1.0, combo, **opts) as p:
            got = p.process(torch.from_numpy(x).cuda(), op, **kw).cpu().numpy()
            assert p.get_option("last_fused_fold") == 0
            assert rel_err(got, want) <= tol, (combo, m, op, opts, rel_err(got, want))
            if op == "gate" and opts.get("fused_exact") and combo.endswith("f32") and m <= 4096:
                assert np.array_equal(got, want)                  # same bins gated, same sums
            hop = x[:200] * 0.7
            wh, _ = nonlinear_reference(ref, hop, op, p0, p1)
            assert rel_err(p.process(hop, op, **kw), wh) <= tol
    if m <= 2048:
        ref = O.best(m, "hamming", 1.0, combo)
        want, dwant = nonlinear_reference(ref, x, op, p0, p1)
        with make(m, "hamming", 1.0, combo) as p:
            dd = torch.empty((n, m), dtype=getattr(torch, np.dtype(fdx).name), device="cuda")
            got = p.process(torch.from_numpy(x).cuda(), op, dfts=dd, **kw).cpu().numpy()
            assert rel_err(got, want) <= tol and rel_err(dd.cpu().numpy(), dwant) <= tol


def test_many_channels_long_call_takes_the_fused_kernel():
    """512 channels leave one time chunk per channel however long the call is: still the fused kernel,
    never the (channels, n, N) workspace."""
    import torch
    C, m, n = 512, 64, 1500
    x = np.stack([noise(n, seed=200 + c) for c in range(C)])
    gain = np.linspace(1.0, 0.3, m)
    with make(m, "hann", 1.0, "f32f64", C) as p:
        got = p.process(torch.from_numpy(x).cuda(), "gain", gain=gain).cpu().numpy()
        assert p.get_option("last_process_path") == 1 and p.get_option("last_chunks") == 1
        for c in (0, 17, 255, 511):
            ref = O.best(m, "hann", 1.0, "f32f64")
            want, _ = reference(ref, x[c], "gain", gain, 0)
            assert rel_err(got[c], want) <= 1e-6
    # batched plans on the two-pass path run in segments of the workspace
    C, m, n = 3, 96, 5000
    x = np.stack([noise(n, seed=300 + c) for c in range(C)])
    with make(m, "hann", 1.0, "f32f64", C, fold=0, rows_kernel=0, stage_bytes=C * m * 16 * 700) as p:
        got = p.process(x, "identity")
        assert p.get_option("last_process_path") == 3
        for c in range(C):
            ref = O.best(m, "hann", 1.0, "f32f64")
            assert rel_err(got[c], ref.isdft(ref.sdft(x[c]))) <= 1e-6


EXPR = ("const sdft_fd_t m2 = re * re + im * im;"
        "const sdft_fd_t g = m2 / (m2 + p[0] * (sdft_fd_t)(1 + k)) * (p[1] + p[2] * cos((sdft_fd_t)t * p[3])) * (sdft_fd_t)(1 + ch);"
        "re *= g; im *= g;")


def expr_reference(ref, x, pv, ch=0, t0=0):
    """EXPR in numpy on the oracle's spectrum (a smooth function of the bin: no threshold a rounding could cross)."""
    d = ref.sdft(x)
    k = np.arange(d.shape[1])[None, :]
    t = (t0 + np.arange(d.shape[0]))[:, None]
    m2 = d.real.astype(np.float64) ** 2 + d.imag.astype(np.float64) ** 2
    g = m2 / (m2 + pv[0] * (1 + k)) * (pv[1] + pv[2] * np.cos(t * pv[3])) * (1 + ch)
    out = (d * g).astype(d.dtype)
    return ref.isdft(out), out


@pytest.mark.parametrize("combo,m,window", [("f32f64", 1024, "hann"), ("f32f32", 512, "hamming"), ("f64f64", 100, "blackman"), ("f32f64", 2048, "boxcar"),
                                            ("f32f64", 2500, "hann"), ("f64f32", 64, "hann")])
def test_operation_handed_in_as_code(combo, m, window):
    """sdft_hip_op_expr: the host's statements run on every windowed bin inside the fused kernel (compiled at run time), or on
    the stored rows where no workgroup holds a row / the call is one time chunk; against oracle-sdft -> numpy -> oracle-isdft."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    n = 9000
    x = (noise(n, seed=21, dtype=td) * 0.5 + sine_sweep(n, dtype=td) * 0.5).astype(td)
    pv = [0.3, 0.8, 0.2, 0.01]
    tol = TOL[combo[3:]]
    ref = O.best(m, window, 1.0, combo)
    want, dwant = expr_reference(ref, x, pv)
    with make(m, window, 1.0, combo) as p:
        got = p.process(torch.from_numpy(x).cuda(), "expr", expr=EXPR, expr_params=pv).cpu().numpy()
        assert p.get_option("last_process_path") == (1 if m <= 2048 else 3), p.get_option("last_process_path")
        assert rel_err(got, want) <= tol, (combo, m, rel_err(got, want))
        # a hop-sized call (one time chunk: analysis -> expression on the rows -> synthesis), host pointers
        hop = (x[:150] * 0.7).astype(td)
        wh, _ = expr_reference(ref, hop, pv)
        assert rel_err(p.process(hop, "expr", expr=EXPR, expr_params=pv), wh) <= tol
        assert p.get_option("last_process_path") == 2
    # the processed spectrum on request; other parameters with the same (cached) code
    ref = O.best(m, window, 1.0, combo)
    pv2 = [0.05, 1.0, 0.5, 0.002]
    want, dwant = expr_reference(ref, x, pv2)
    with make(m, window, 1.0, combo) as p:
        dd = torch.empty((n, m), dtype=getattr(torch, np.dtype(fdx).name), device="cuda")
        got = p.process(torch.from_numpy(x).cuda(), "expr", expr=EXPR, expr_params=pv2, dfts=dd).cpu().numpy()
        assert rel_err(got, want) <= tol and rel_err(dd.cpu().numpy(), dwant) <= tol


def test_operation_handed_in_as_code_batched_and_errors():
    import torch
    from sdft_amd.capi import SdftHipError
    C, m, n = 3, 256, 7000
    x = np.stack([noise(n, seed=400 + c) for c in range(C)])
    pv = [0.2, 1.0, 0.1, 0.02]
    with make(m, "hann", 1.0, "f32f64", C) as p:
        got = p.process(torch.from_numpy(x).cuda(), "expr", expr=EXPR, expr_params=pv).cpu().numpy()
        for c in range(C):
            ref = O.best(m, "hann", 1.0, "f32f64")
            want, _ = expr_reference(ref, x[c], pv, ch=c)
            assert rel_err(got[c], want) <= 1e-6, (c, rel_err(got[c], want))
    # more than eight parameters travel through device memory instead of the kernel arguments; a parameter picked by a run-time index
    m, n = 128, 3000
    x1 = noise(n, seed=405)
    pv = np.linspace(0.2, 1.4, 12)
    ref = O.best(m, "hann", 1.0, "f32f64")
    d = ref.sdft(x1)
    want = ref.isdft((d * pv[np.arange(m) % 12][None, :]).astype(d.dtype))
    for params, code in ((pv, "const sdft_fd_t g = p[k % 12]; re *= g; im *= g;"), (pv[:8], "const sdft_fd_t g = p[k % 8]; re *= g; im *= g;")):
        with make(m, "hann", 1.0, "f32f64") as p:
            got = p.process(x1, "expr", expr=code, expr_params=params)
            w = want if len(params) == 12 else ref.isdft((d * pv[np.arange(m) % 8][None, :]).astype(d.dtype))
            assert rel_err(got, w) <= 1e-6, (len(params), rel_err(got, w))
            hop = x1[:120]
            ref2 = O.best(m, "hann", 1.0, "f32f64"); d2 = ref2.sdft(hop)
            p.reset()
            assert rel_err(p.process(hop, "expr", expr=code, expr_params=params), ref2.isdft((d2 * pv[np.arange(m) % len(params)][None, :]).astype(d2.dtype))) <= 1e-6
    # reference order of the sum (fused_exact = 1) with an expression that is exact in any arithmetic: bit-identical at FD float
    m, n = 512, 6000
    x1 = noise(n, seed=410)
    ref = O.best(m, "hamming", 1.0, "f32f32")
    d = ref.sdft(x1)
    d[:, 1::2] = 0
    want = ref.isdft(d)
    with make(m, "hamming", 1.0, "f32f32", fused_exact=1) as p:
        got = p.process(x1, "expr", expr="if (k & 1) { re = 0; im = 0; }")
        assert np.array_equal(got, want)
        # statements that do not compile: the compiler's words, and the plan goes on
        with pytest.raises(SdftHipError) as e:
            p.process(x1, "expr", expr="re = nonsense(im);")
        assert "does not compile" in str(e.value) and "nonsense" in str(e.value)
        ref.reset(); p.reset()
        assert np.array_equal(p.sdft(x1[:3000]), ref.sdft(x1[:3000]))


def test_reference_bits_by_rounding_interval():
    """fused_exact with float samples and double bins: the tree sum of a sample's terms plus the bound on what ANY summation
    order can differ by (2 * n * 2^-53 * sum|term|) pins the float the reference's ordered sum rounds to, unless the interval
    straddles a rounding boundary -- then the terms are walked in order.  Bit-identical either way; both ways must occur."""
    import torch
    for m, window, n in ((1024, "hann", 60000), (2048, "blackman", 24000), (300, "hamming", 40000)):
        x = (noise(n, seed=61) * 0.3 + sine_sweep(n) * 0.7).astype(np.float32)
        ref = O.best(m, window, 1.0, "f32f64")
        gain = np.linspace(1.0, 0.2, m)
        d = ref.sdft(x)
        want = ref.isdft((d * gain[None, :]).astype(d.dtype))
        with make(m, window, 1.0, "f32f64", carry=1, fused_exact=2) as p:
            got = p.process(torch.from_numpy(x).cuda(), "gain", gain=gain).cpu().numpy()
            assert p.get_option("last_fused_exact") == 1 and p.get_option("last_process_path") == 1
            assert np.array_equal(got, want), (m, int((got != want).sum()))
            walks = p.get_option("ordered_walks")
            assert 0 < walks < n // 20, (m, walks, n)              # a fraction of a percent to a few percent of the samples
        # double samples have no rounding to hide behind: every sample is walked, same bits
    x = noise(6000, seed=62, dtype=np.float64)
    ref = O.best(256, "hann", 1.0, "f64f64")
    with make(256, "hann", 1.0, "f64f64", carry=1, fused_exact=2) as p:
        assert np.array_equal(p.process(x, "identity"), ref.isdft(ref.sdft(x)))
        assert p.get_option("ordered_walks") == 0                    # (the counter belongs to the interval test)


@pytest.mark.parametrize("combo,m,n", [("f32f64", 1024, 48000),      # self-carried chunks (one launch, workgroups read earlier chunks' samples)
                                       ("f32f64", 1000, 20000),      # pre-pass route, 2N = 2/3/5-smooth
                                       ("f32f64", 256, 100),         # a hop: the one-launch hop kernel
                                       ("f32f64", 1024, 100000),     # beyond the self-carried form's limit for the fused call
                                       ("f32f32", 512, 30000),       # exact carries (relay + flow mode)
                                       ("f64f64", 256, 9000)])
def test_in_place_and_overlapping_calls(combo, m, n):
    """out == samples on the device, and out overlapping samples: the two reference calls read every sample before they
    write the first output sample (sdft.h:607-613 then :666-672), so the fused call must give what it gives out of place
    -- the one-launch forms read earlier chunks' samples while other workgroups write their outputs (round-3 advisor
    finding); the library copies the samples aside when the ranges overlap."""
    import torch
    td = np.float32 if combo.startswith("f32") else np.float64
    x = noise(n + 64, seed=900 + m).astype(td)
    gain = (0.5 + 0.5 * np.cos(np.arange(m) * 0.01)).astype(np.float64)
    ref = O.best(m, "hann", 1.0, combo)
    want_ref, _ = reference(ref, x[:n], "gain", gain.astype(ref.fd if hasattr(ref, "fd") else np.float64), 0)
    x2 = noise(700, seed=901).astype(td)
    with make(m, "hann", 1.0, combo) as a, make(m, "hann", 1.0, combo) as b, make(m, "hann", 1.0, combo) as c:
        g = gain.astype(a.fd)
        xa = torch.from_numpy(x[:n]).cuda()
        want = a.process(xa, "gain", gain=g)                                  # out of place
        assert rel_err(want.cpu().numpy(), want_ref) <= TOL[combo[3:]]
        xb = xa.clone()
        got = b.process(xb, "gain", gain=g, out=xb)                           # in place
        assert got.data_ptr() == xb.data_ptr()
        assert torch.equal(want, got), rel_err(got.cpu().numpy(), want.cpu().numpy())
        buf = torch.from_numpy(x).cuda()                                      # out starts 64 samples into the input
        xc, yc = buf[:n], buf[64:64 + n]
        got_c = c.process(xc, "gain", gain=g, out=yc)
        assert torch.equal(want, got_c), rel_err(got_c.cpu().numpy(), want.cpu().numpy())
        # the stream state (delay line included) all three calls leave is the same
        t2 = torch.from_numpy(x2).cuda()
        w2 = a.process(t2, "gain", gain=g)
        assert torch.equal(w2, b.process(t2, "gain", gain=g)) and torch.equal(w2, c.process(t2, "gain", gain=g))


@pytest.mark.parametrize("m,n,opts", [(512, 6000, {}),                          # fused kernel: the compilation fails at the K1 launch
                                      (512, 6000, {"carry": 1}),                # ... with the carry launches already queued
                                      (256, 120, {}),                           # a hop: analysis, then the row synthesis with the statements
                                      (6000, 3000, {})])                        # rows no workgroup holds: analysis -> statements -> synthesis
def test_failed_expression_leaves_the_stream_where_it_was(m, n, opts):
    """Statements that do not compile: rc = -1 AND the plan's stream state is untouched, so a host that fixes the typo and
    sends the same samples again gets what a fresh plan gives (round-3 advisor finding: the delay line was flipped, or the
    whole analysis had advanced the state, before the compilation failed)."""
    from sdft_amd.capi import SdftHipError
    x0, x1 = noise(2500, seed=77), noise(n, seed=78)
    good = "re *= p[0]; im *= p[0];"
    with make(m, "hann", 1.0, "f32f64", **opts) as p, make(m, "hann", 1.0, "f32f64", **opts) as q:
        p.process(x0); q.process(x0)                                           # some history and a cursor
        before = p.state()
        with pytest.raises(SdftHipError) as e:
            p.process(x1, "expr", expr="re *= p[0]; im *= nonsense(p[0]);", expr_params=[0.5])
        assert "does not compile" in str(e.value)
        after = p.state()
        for u, v in zip(before, after):
            assert np.array_equal(np.asarray(u), np.asarray(v))
        assert np.array_equal(p.process(x1, "expr", expr=good, expr_params=[0.5]), q.process(x1, "expr", expr=good, expr_params=[0.5]))
