"""Parity of the HIP path (through the C-ABI) against the oracle, on a real MI355X.

Bars (BASELINE.json north_star): max|a-b| / max|b| <= 1e-6 for FD double, <= 1e-4 for FD float.
Where the HIP path is time-serial (single chunk, or the exact-carry mode that FD float always
uses) the forward result is required to be bit-identical.
"""

import numpy as np
import pytest

from oracle import oracle as O
from sdft_amd.signals import noise, sine_sweep, sweep_batch

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-6, "f32": 1e-4}


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    scale = float(np.abs(b).max())
    if scale == 0.0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max()) / scale


def make(dftsize, window="hann", latency=1.0, combo="f32f64", channels=1, **opts):
    from sdft_amd.sdft import SDFT
    p = SDFT(dftsize, window, latency, combo, channels)
    for k, v in opts.items():
        p.set_option(k, v)
    return p


def same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)) or np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------
# small exhaustive: every combo x window x latency, tiny and ragged sizes, host-pointer path
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("combo", O.COMBOS)
@pytest.mark.parametrize("window", ["boxcar", "hann", "hamming", "blackman"])
def test_small_sizes_bit_exact(combo, window):
    td, fd, fdx = O.combo_types(combo)
    for m in (1, 2, 3, 4, 5, 7, 61, 62, 63, 64, 100, 125, 130):
        for latency in (1.0, 0.5):
            n = 5 * m + 3
            x = noise(n, seed=m, dtype=td)
            ref = O.best(m, window, latency, combo)
            want = ref.sdft(x)
            with make(m, window, latency, combo, chunk=1 << 30) as p:
                got = p.sdft(x)                      # numpy in -> staged host path, single chunk
                assert np.array_equal(got, want), (combo, window, m, latency, rel_err(got, want))
                y = p.isdft(got)
                assert rel_err(y, ref.isdft(want)) <= TOL[combo[3:]], (combo, window, m, latency)
                acc, fid, hist, cur = p.state()
                racc, rfid, rhist, rcur = ref.state()
                assert cur == rcur and np.array_equal(acc, racc) and np.array_equal(fid, rfid) and np.array_equal(hist, rhist)


@pytest.mark.parametrize("combo", O.COMBOS)
def test_streaming_hops_match_one_call(combo):
    """test.c:69-83 pattern: hop-wise calls with persistent state == the reference, bit for bit."""
    td, fd, fdx = O.combo_types(combo)
    m, hop = 100, 37
    x = sine_sweep(20 * hop, dtype=td)
    ref = O.best(m, "hann", 1.0, combo)
    with make(m, "hann", 1.0, combo) as p:            # hops < 512 samples are never time-chunked
        for i in range(0, x.size, hop):
            want = ref.sdft(x[i:i + hop])
            got = p.sdft(x[i:i + hop])
            assert np.array_equal(got, want)
            assert rel_err(p.isdft(got), ref.isdft(want)) <= TOL[combo[3:]]
        p.reset(); ref.reset()
        assert np.array_equal(p.sdft(x[:50]), ref.sdft(x[:50]))


# ---------------------------------------------------------------------------------------------
# time-chunked paths
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("window", ["hann", "blackman", "hamming", "boxcar"])
def test_fast_carry_double_chunked(window):
    """FD double, chunk-parallel carries (sum order differs from the serial reference)."""
    m, n = 1024, 20000
    x = sine_sweep(n)
    ref = O.best(m, window, 1.0, "f32f64")
    want = ref.sdft(x)
    with make(m, window, 1.0, "f32f64", chunk=512, carry=0) as p:
        got = p.sdft(x)
        assert p.get_option("last_chunks") == 40
        e = rel_err(got, want)
        assert e <= 1e-11, e                     # far inside the 1e-6 bar
        # second call continues from the chunked state
        x2 = noise(3000)
        e2 = rel_err(p.sdft(x2), ref.sdft(x2))
        assert e2 <= 1e-11, e2


@pytest.mark.parametrize("combo", ["f32f32", "f64f32", "f32f64"])
@pytest.mark.parametrize("m,block", [(256, 32), (1000, 16), (100, 8), (7, 0)])
def test_exact_carry_chain_form_bit_exact(combo, m, block):
    """Exact carries in relay form: fid regenerated from the plan's seed table off the chain, one dependent addition per
    sample on it (identical waves take the blocks in turn; a block's products stay in registers, acc is a token).  Must
    reproduce the reference bit for bit for every block length the geometry allows (2N divisible by 32 / 16 / 8;
    otherwise the serial pass runs), for calls that start mid-period and mid-block, across roll-overs, in time segments,
    for batched channels -- and agree with the serial pass it replaces."""
    td, fd, fdx = O.combo_types(combo)
    ch = 2
    lens = (3 * m + 5, 4 * m + 8 * 37, 555)                      # cursors at arbitrary offsets
    xb = np.stack([noise(sum(lens), seed=3 + c, dtype=td) for c in range(ch)])
    for segments in (1, 3):
        refs = [O.best(m, "blackman", 1.0, combo) for _ in range(ch)]
        with make(m, "blackman", 1.0, combo, ch, chunk=64, carry=1, chain=2, segments=segments) as p, \
             make(m, "blackman", 1.0, combo, ch, chunk=64, carry=1, chain=0, segments=segments) as q:
            i = 0
            for n in lens:
                seg = np.ascontiguousarray(xb[:, i:i + n])
                got, old = p.sdft(seg), q.sdft(seg)
                assert p.get_option("last_chain") == (3 if block else 0) and q.get_option("last_chain") == 0
                assert p.get_option("last_chunks") > 2 or m == 7
                for c in range(ch):
                    want = refs[c].sdft(seg[c])
                    assert np.array_equal(got[c], want), (combo, m, n, c, rel_err(got[c], want))
                assert np.array_equal(got, old)
                i += n
            acc, fid, hist, cur = p.state()
            for c in range(ch):
                racc, rfid, rhist, rcur = refs[c].state()
                assert cur == rcur and np.array_equal(acc[c], racc) and np.array_equal(fid[c], rfid)
    # forced block lengths / wave counts
    if block:
        x = noise(5 * m + 77, seed=9, dtype=td)
        want = O.best(m, "hann", 1.0, combo).sdft(x)
        for L, W in ((8, 1), (8, 8), (8, 3), (block if block < 32 else 32, 2)):
            with make(m, "hann", 1.0, combo, chunk=96, carry=1, chain=2, chain_block=L, relay_waves=W) as p:
                got = p.sdft(x)
                assert p.get_option("last_chain") == 3
                assert np.array_equal(got, want), (combo, m, L, W, "relay")


@pytest.mark.parametrize("combo,m,chunk,want_l", [("f32f32", 4096, 1024, 128), ("f32f32", 1024, 0, 128), ("f64f32", 512, 192, 64),
                                                  ("f32f64", 1024, 512, 64), ("f64f64", 256, 96, 32), ("f32f32", 2048, 128, 128)])
def test_exact_carry_relay_form_long_blocks(combo, m, chunk, want_l):
    """Relay form at the block lengths the big shapes use (128 products in 128 registers per lane at FD float, 64 at FD
    double): calls that start at cursor 0, mid-block (the ragged first block adds nothing for the steps before the call)
    and just before the roll-over; batched; in overlap segments; state and the hop that follows bit-identical."""
    td, fd, fdx = O.combo_types(combo)
    ref = O.best(m, "blackman", 1.0, combo)
    lens = (9000 + 37, 2 * m + 4096 + 5, 6 * 1024, 4096 + 63)
    with make(m, "blackman", 1.0, combo, chunk=chunk, carry=1, chain=2) as p:
        for i, n in enumerate(lens):
            x = noise(n, seed=50 + i, dtype=td)
            want = ref.sdft(x)
            got = p.sdft(x)
            assert p.get_option("last_chain") == 3, (p.get_option("last_chain"), p.get_option("last_chunk_len"))
            assert np.array_equal(got, want), (combo, m, n, rel_err(got, want))
            hop = noise(77, seed=60 + i, dtype=td)
            assert np.array_equal(p.sdft(hop), ref.sdft(hop))
        acc, fid, hist, cur = p.state()
        racc, rfid, rhist, rcur = ref.state()
        assert cur == rcur and np.array_equal(acc, racc) and np.array_equal(fid, rfid) and np.array_equal(hist, rhist)
    # batched, segments on the auxiliary stream, odd wave counts
    C = 3
    xb = np.stack([noise(12000, seed=70 + c, dtype=td) for c in range(C)])
    wants = [O.best(m, "hann", 1.0, combo).sdft(xb[c]) for c in range(C)]
    for waves, segments in ((0, 0), (2, 3), (5, 1), (8, 4)):
        with make(m, "hann", 1.0, combo, C, chunk=chunk, carry=1, chain=2, relay_waves=waves, segments=segments) as p:
            got = p.sdft(xb)
            assert p.get_option("last_chain") == 3
            for c in range(C):
                assert np.array_equal(got[c], wants[c]), (combo, m, waves, segments, c)


def test_exact_chain_falls_back_while_fid_is_off_the_canonical_sequence():
    """The seed table holds the rotation sequence that starts from 1 at the roll-over.  After a
    chunk-parallel call (fid seeded from the closed-form table) or set_state the stream's fid is not
    on it: the serial pass runs until the next roll-over puts fid back to exactly 1."""
    m = 128
    x = noise(40 * m, seed=2)
    ref = O.best(m, "hann", 1.0, "f32f64")
    with make(m, "hann", 1.0, "f32f64", chunk=64, carry=0) as p:
        a = p.sdft(x[:700])                                      # fast carries: fid leaves the sequence
        ref.sdft(x[:700])
        assert p.get_option("last_chain") == 0
        p.set_option("carry", 1); p.set_option("chain", 2)
        b = p.sdft(x[700:700 + 5 * m])                           # serial pass (and crosses the roll-over)
        assert p.get_option("last_chain") == 0
        c = p.sdft(x[700 + 5 * m:])                              # back on the sequence: chain form
        assert p.get_option("last_chain") >= 1
        wb = ref.sdft(x[700:700 + 5 * m]); wc = ref.sdft(x[700 + 5 * m:])
        assert rel_err(b, wb) <= 1e-11 and rel_err(c, wc) <= 1e-11
    # from a clean start the same sequence of exact calls is bit-identical throughout
    ref = O.best(m, "hann", 1.0, "f32f64")
    with make(m, "hann", 1.0, "f32f64", chunk=64, carry=1, chain=2) as p:
        for lo, hi in ((0, 700), (700, 700 + 5 * m), (700 + 5 * m, 40 * m)):
            assert np.array_equal(p.sdft(x[lo:hi]), ref.sdft(x[lo:hi]))
            assert p.get_option("last_chain") >= 1


@pytest.mark.parametrize("combo", ["f32f32", "f64f32", "f32f64"])
@pytest.mark.parametrize("window", ["hann", "blackman"])
@pytest.mark.parametrize("segments", [1, 3, 7, 30])
def test_exact_carry_chunked_bit_exact(combo, window, segments):
    """Exact-carry mode: time chunks seeded by the serial pre-pass reproduce the reference bit for bit,
    also when the pass runs in time segments on the auxiliary stream, overlapped with the forward
    launches of earlier segments."""
    td, fd, fdx = O.combo_types(combo)
    m, n = 256, 6000
    x = noise(n, dtype=td)
    ref = O.best(m, window, 1.0, combo)
    want = ref.sdft(x)
    with make(m, window, 1.0, combo, chunk=200, carry=1, segments=segments) as p:
        got = p.sdft(x)
        assert p.get_option("last_chunks") == 30 and p.get_option("last_segments") == segments
        assert np.array_equal(got, want), rel_err(got, want)
        x2 = noise(777, seed=5, dtype=td)
        assert np.array_equal(p.sdft(x2), ref.sdft(x2))
        acc, fid, hist, cur = p.state()
        racc, rfid, rhist, rcur = ref.state()
        assert cur == rcur and np.array_equal(acc, racc) and np.array_equal(fid, rfid) and np.array_equal(hist, rhist)


def test_config3_shape_float_blackman_roundtrip():
    """BASELINE config 3 at parity size: m=4096, Blackman, FD float, latency 1, n=20000."""
    m, n = 4096, 20000
    x = sine_sweep(n)
    ref = O.best(m, "blackman", 1.0, "f32f32")
    want = ref.sdft(x)
    with make(m, "blackman", 1.0, "f32f32", chunk=1000) as p:
        got = p.sdft(x)
        assert np.array_equal(got, want), rel_err(got, want)
        y = p.isdft(got)
        assert rel_err(y, ref.isdft(want)) <= 1e-4


def test_device_pointers_and_batch():
    """Device-resident tensors (no copies) and a batched plan == per-channel reference plans."""
    import torch
    ch, m, n = 5, 200, 3000
    xb = sweep_batch(ch, n)
    with make(m, "hann", 1.0, "f32f64", channels=ch, chunk=256) as p:
        xd = torch.from_numpy(xb).cuda()
        d = p.sdft(xd)
        y = p.isdft(d)
        torch.cuda.synchronize()
        dn, yn = d.cpu().numpy(), y.cpu().numpy()
    for c in range(ch):
        ref = O.best(m, "hann", 1.0, "f32f64")
        want = ref.sdft(xb[c])
        assert rel_err(dn[c], want) <= 1e-11
        assert rel_err(yn[c], ref.isdft(want)) <= 1e-6


def test_latency_synthesis_branch():
    m, n = 512, 4000
    x = noise(n)
    for combo in ("f32f64", "f32f32"):
        for latency in (0.5, 0.25):
            ref = O.best(m, "hann", latency, combo)
            want = ref.sdft(x)
            with make(m, "hann", latency, combo) as p:
                got = p.sdft(x)
                assert rel_err(got, want) <= 1e-11 if combo == "f32f64" else np.array_equal(got, want)
                assert rel_err(p.isdft(got), ref.isdft(want)) <= TOL[combo[3:]]


def test_config1_shape_n48000():
    """BASELINE config 1/2 shape (m=1024, Hann, TD float / FD double) at n=48000, default chunking."""
    m, n = 1024, 48000
    x = sine_sweep(n)
    ref = O.best(m, "hann", 1.0, "f32f64")
    dig, yref = O.Port(m, "hann", 1.0, "f32f64").digest(x)
    want = ref.sdft(x)
    import torch
    with make(m) as p:
        d = p.sdft(torch.from_numpy(x).cuda())
        y = p.isdft(d)
        got = d.cpu().numpy()
        assert p.get_option("last_chunks") > 1
        assert rel_err(got, want) <= 1e-11
        assert rel_err(y.cpu().numpy(), yref) <= 1e-6
        assert np.allclose(got.real.sum(axis=1), dig[:, 0], rtol=0, atol=1e-9)


@pytest.mark.parametrize("combo,m", [("f32f64", 64), ("f32f64", 1000), ("f32f64", 1024), ("f32f64", 127), ("f32f32", 2048),
                                     ("f32f32", 130), ("f64f64", 512), ("f64f32", 1001),
                                     # two slots per lane (rows longer than one pass of the 16 waves)
                                     ("f32f64", 2048), ("f32f64", 1025), ("f32f64", 1500), ("f32f32", 4096), ("f32f32", 2049),
                                     ("f64f32", 3001)])
@pytest.mark.parametrize("window", ["hann", "blackman", "boxcar", "hamming"])
def test_row_group_kernel_equals_tile_kernel_and_oracle(combo, m, window):
    """The two forward kernels (row-group with LDS halo exchange / independent tiles with halo lanes)
    must agree bit for bit with each other for the same carries, and with the oracle."""
    td, fd, fdx = O.combo_types(combo)
    n = 3000
    x = noise(n, seed=11, dtype=td)
    want = O.best(m, window, 1.0, combo).sdft(x)
    got = {}
    for rows_kernel in (1, 0):
        with make(m, window, 1.0, combo, chunk=504, carry=1, rows_kernel=rows_kernel) as p:
            got[rows_kernel] = p.sdft(x)
            assert p.get_option("last_kernel") == (2 if rows_kernel else 1)
            assert p.get_option("last_chunks") == 6
            x2 = noise(100, seed=12, dtype=td)           # state written by the last chunk
            got[(rows_kernel, "next")] = p.sdft(x2)
    assert np.array_equal(got[1], got[0])
    assert np.array_equal(got[1], want), rel_err(got[1], want)
    assert np.array_equal(got[(1, "next")], got[(0, "next")])


@pytest.mark.parametrize("m,chunk", [(1024, 512), (1024, 3000), (256, 600), (64, 128), (512, 5000),
                                     # mixed radix (2N = 2000, 1500, 192, 250, 1800) and a size that falls back (1001)
                                     (1000, 352), (1000, 2600), (750, 400), (96, 64), (125, 1000), (900, 512), (1001, 256)])
def test_fft_carry_equals_direct_sums(m, chunk):
    """Chunk partial sums by the in-LDS FFT (power-of-two N) vs the direct sums vs the oracle, incl.
    chunks longer than 2N (folding) and a second call that continues from a non-zero cursor."""
    n = 20000
    x = noise(n, seed=21)
    ref = O.best(m, "hann", 1.0, "f32f64")
    want = ref.sdft(x)
    x2 = sine_sweep(7000)
    want2 = ref.sdft(x2)
    outs = {}
    for fft in (1, 0):
        with make(m, "hann", 1.0, "f32f64", chunk=chunk, carry=0, fft_carry=fft, self_carry=0) as p:     # the pre-pass forms
            outs[fft] = (p.sdft(x), p.sdft(x2))
            assert p.get_option("last_chunks") > 1 and p.get_option("last_self") == 0
    for fft in (1, 0):
        assert rel_err(outs[fft][0], want) <= 1e-11, (fft, rel_err(outs[fft][0], want))
        assert rel_err(outs[fft][1], want2) <= 1e-11, (fft, rel_err(outs[fft][1], want2))
    assert rel_err(outs[1][0], outs[0][0]) <= 1e-12


@pytest.mark.parametrize("combo", O.COMBOS)
@pytest.mark.parametrize("latency", [1.0, 0.5])
def test_inverse_is_bit_identical(combo, latency):
    """The default inverse sums the bins of a row in the reference's order (LDS transpose kernel):
    y is bit-identical for every type, ragged sizes, host and device pointers, batches."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    for m, n in ((1, 5), (7, 70), (100, 333), (1000, 129), (1024, 4100), (2050, 64)):
        x = noise(n, seed=m, dtype=td)
        ref = O.best(m, "hann", latency, combo)
        d = ref.sdft(x)
        want = ref.isdft(d)
        with make(m, "hann", latency, combo) as p:
            assert np.array_equal(p.isdft(d), want), (combo, latency, m, n)
            got = p.isdft(torch.from_numpy(d).cuda()).cpu().numpy()
            assert np.array_equal(got, want), (combo, latency, m, n)
            # every streaming form of the exact-order kernel (rows per wave x tiles in flight; round 4 added 8 x 4)
            for rows in (4, 8, 16) + ((32,) if combo[3:] == "f64" else ()):
                p.set_option("inverse_rows", rows)
                assert np.array_equal(p.isdft(torch.from_numpy(d).cuda()).cpu().numpy(), want), (combo, latency, m, n, rows)
            p.set_option("inverse_rows", 0)
            p.set_option("exact_inverse", 0)                       # wave-parallel sum: inside the bar, not identical
            assert rel_err(p.isdft(d), want) <= TOL[combo[3:]]
    ch, m, n = 3, 96, 200
    xb = np.stack([noise(n, seed=c, dtype=td) for c in range(ch)])
    refs = [O.best(m, "hann", latency, combo) for _ in range(ch)]
    db = np.stack([refs[c].sdft(xb[c]) for c in range(ch)])
    with make(m, "hann", latency, combo, channels=ch) as p:
        yb = p.isdft(db)
    for c in range(ch):
        assert np.array_equal(yb[c], refs[c].isdft(db[c]))


@pytest.mark.parametrize("combo,m,n,channels", [("f32f64", 1024, 70000, 1), ("f64f64", 1000, 66000, 1), ("f32f32", 2048, 66000, 1), ("f32f64", 256, 9000, 8),
                                                ("f64f64", 1000, 20000, 1), ("f32f64", 512, 30000, 1)])
def test_long_synthesis_calls_find_their_form_and_keep_their_bits(combo, m, n, channels):
    """Round 5: from 8 Ki rows on the plan measures the bit-identical streaming forms of the synthesis (4 / 8 / 16 / 32 rows per
    wave, 256- / 512-byte row segments, the tree sum with the rounding-interval proof) on the host's own calls and keeps the
    fastest (logic::FormTuner).  Every call of the trial phase and after it gives the same samples -- the reference's."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    rng = np.random.default_rng(m + n)
    shape = (channels, n, m) if channels > 1 else (n, m)
    d = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(fdx)
    d[..., ::9, :] *= 1e-6
    refs = [O.best(m, "hann", 1.0, combo) for _ in range(channels)]
    want = np.stack([refs[c].isdft(d[c]) for c in range(channels)]) if channels > 1 else refs[0].isdft(d)
    dd = torch.from_numpy(d).cuda()
    with make(m, "hann", 1.0, combo, channels) as p:
        forms = []
        for call in range(12):
            y = p.isdft(dd).cpu().numpy()
            assert np.array_equal(y, want), (combo, call, p.get_option("last_inverse_tuned"))
            forms.append(p.get_option("last_inverse_tuned"))
        assert forms[-1] >= 10 and forms[-1] == forms[-2], forms       # decided (tens digit), and stays decided
        assert len(set(f % 10 for f in forms)) >= 2, forms             # more than one form was tried on the way
        # another shape starts over; a forced form is not tuned
        p.isdft(dd[..., : n - 16, :].contiguous())
        assert p.get_option("last_inverse_tuned") < 10
        p.set_option("inverse_rows", 16)
        before = p.get_option("last_inverse_tuned")
        assert np.array_equal(p.isdft(dd).cpu().numpy(), want) and p.get_option("last_inverse_tuned") == before


@pytest.mark.parametrize("m,n,latency,channels", [(1024, 30000, 1.0, 1), (1000, 12000, 0.5, 1), (77, 50000, 1.0, 1), (2050, 3000, 1.0, 1), (256, 9000, 0.7, 3)])
def test_inverse_bits_by_rounding_interval(m, n, latency, channels):
    """Medium calls, float samples from double bins: the synthesis sums a row by a tree and proves the reference's float from
    the bound on what any summation order can differ by; rows it cannot prove are added in ascending bin order.  Spectra
    of any kind (not only what the analysis produces), the streaming kernel with the reference's order as the second witness."""
    import torch
    rng = np.random.default_rng(m + n)
    d = (rng.standard_normal((channels, n, m)) * np.exp(rng.standard_normal((channels, n, 1)) * 3) + 1j * rng.standard_normal((channels, n, m))).astype(np.complex128)
    d[:, ::7, :] *= 1e-9                                              # rows near zero, rows of mixed magnitude
    d[:, 5::11, ::2] = 0
    refs = [O.best(m, "hamming", latency, "f32f64") for _ in range(channels)]
    want = np.stack([r.isdft(d[c]) for c, r in enumerate(refs)])
    dd = torch.from_numpy(d if channels > 1 else d[0]).cuda()
    with make(m, "hamming", latency, "f32f64", channels) as p:
        got = p.isdft(dd).cpu().numpy()
        assert p.get_option("last_inverse_form") == 2
        assert np.array_equal(got if channels > 1 else got[None], want), int((got != want).sum())
        p.set_option("inverse_verify", 0)
        old = p.isdft(dd).cpu().numpy()
        assert p.get_option("last_inverse_form") == 1 and np.array_equal(old, got)


@pytest.mark.parametrize("m,latency", [(1024, 1.0), (1000, 0.5), (64, 1.0)])
def test_rows_on_a_rounding_boundary(m, latency):
    """Adversarial rows for the rounding-interval test: bin 0 of a random row is bisected (with the CPU reference) to the
    value at which the reference's float sample flips to its neighbour, and the row is offered at that value and a few
    units in the last place around it -- sums that sit on a rounding boundary of the float, where only the reference's own
    order decides.  Every form of the synthesis (row kernel, verified tree sum, streaming kernel) and the fused call in
    the reference's order must reproduce the reference's choice."""
    import torch
    rng = np.random.default_rng(7 * m)
    ref = O.best(m, "hann", latency, "f32f64")
    rows = []
    for _ in range(40):
        row = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex128) * 10.0 ** rng.integers(-3, 4)
        lo, hi = row[0].real - 1.0, row[0].real + 1.0
        def y_at(v):
            r = row.copy(); r[0] = complex(v, row[0].imag)
            return ref.isdft(r[None, :])[0]
        ylo, yhi = y_at(lo), y_at(hi)
        if ylo == yhi:
            continue
        target = ylo
        for _ in range(80):                                          # the largest v that still gives ylo
            mid = 0.5 * (lo + hi)
            if mid == lo or mid == hi:
                break
            if y_at(mid) == target: lo = mid
            else: hi = mid
        for steps in range(-3, 4):
            v = lo
            for _ in range(abs(steps)):
                v = np.nextafter(v, np.inf if steps > 0 else -np.inf)
            r = row.copy(); r[0] = complex(v, row[0].imag)
            rows.append(r)
    d = np.stack(rows)
    n = d.shape[0]
    assert n >= 100
    want = ref.isdft(d)
    assert len(np.unique(want)) > n // 8                              # the rows do straddle boundaries
    reps = (2000 + n - 1) // n                                        # enough rows for the tree-sum and the streaming kernels
    big = np.tile(d, (reps, 1)); wbig = np.tile(want, reps)
    with make(m, "hann", latency, "f32f64") as p:
        assert np.array_equal(p.isdft(torch.from_numpy(d).cuda()).cpu().numpy(), want)            # row kernel (<= 1024 rows)
        got = p.isdft(torch.from_numpy(big).cuda()).cpu().numpy()
        assert p.get_option("last_inverse_form") == 2 and np.array_equal(got, wbig)                # verified tree sum
        p.set_option("inverse_verify", 0)
        assert np.array_equal(p.isdft(torch.from_numpy(big).cuda()).cpu().numpy(), wbig)           # streaming kernel
        if latency == 1.0:
            p.set_option("inverse_verify", 1); p.set_option("inverse_step", 1)
            got = p.isdft(torch.from_numpy(big).cuda()).cpu().numpy()
            assert p.get_option("last_inverse_form") == 3 and np.array_equal(got, wbig)            # whole rows in step (round 5)


@pytest.mark.parametrize("combo,m,opts", [("f32f64", 2500, {"carry": 0, "chunk": 256}), ("f32f64", 2500, {"carry": 1, "chunk": 96, "segments": 3}),
                                          ("f32f32", 4100, {"chunk": 160, "segments": 4}), ("f64f64", 2049, {"carry": 0}),
                                          ("f32f32", 4096, {"rows_kernel": 0, "chunk": 128, "segments": 2})])
def test_batched_plans_beyond_the_row_group_limit(combo, m, opts):
    """Batched plans whose rows do not fit the row-group kernel (independent-tile kernel), both carry
    modes, segmented exact pass; channel by channel against independent reference plans."""
    td, fd, fdx = O.combo_types(combo)
    ch, n = 3, 1500
    xb = np.stack([noise(n, seed=40 + c, dtype=td) for c in range(ch)])
    refs = [O.best(m, "blackman", 1.0, combo) for _ in range(ch)]
    want = [r.sdft(xb[c]) for c, r in enumerate(refs)]
    with make(m, "blackman", 1.0, combo, channels=ch, **opts) as p:
        got = p.sdft(xb)
        exact = bool(p.get_option("carry"))
        assert p.get_option("last_chunks") > 1
        y = p.isdft(got)
        x2 = np.stack([noise(300, seed=70 + c, dtype=td) for c in range(ch)])
        got2 = p.sdft(x2)
    for c in range(ch):
        if exact:
            assert np.array_equal(got[c], want[c]), (combo, m, c)
            assert np.array_equal(got2[c], refs[c].sdft(x2[c])), (combo, m, c)
        else:
            assert rel_err(got[c], want[c]) <= 1e-11
            assert rel_err(got2[c], refs[c].sdft(x2[c])) <= 1e-11
        assert np.array_equal(y[c], refs[c].isdft(got[c]))


@pytest.mark.parametrize("combo,m,n", [("f32f64", 20000, 2500), ("f32f32", 30001, 1200), ("f64f64", 65536, 700)])
def test_very_large_dftsize(combo, m, n):
    """Bin counts far beyond the row-group limit (independent-tile kernel, direct-sum carries):
    index arithmetic and chunk geometry at N up to 65536."""
    td, fd, fdx = O.combo_types(combo)
    x = noise(n, seed=3, dtype=td)
    ref = O.best(m, "hamming", 0.5, combo)
    want = ref.sdft(x)
    with make(m, "hamming", 0.5, combo, chunk=128) as p:
        got = p.sdft(x)
        assert p.get_option("last_chunks") > 1 and p.get_option("last_kernel") == 1
        exact = bool(p.get_option("carry"))
        y = p.isdft(got)
    if exact:
        assert np.array_equal(got, want)
    else:
        assert rel_err(got, want) <= 1e-11
    assert np.array_equal(y, ref.isdft(got))


# ---------------------------------------------------------------------------------------------
# self-carried chunks: the chunk-parallel FD double call as ONE launch (round 3)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("combo,m,chunk", [("f32f64", 1024, 192), ("f32f64", 1024, 3000), ("f32f64", 8, 64), ("f32f64", 64, 128),
                                           ("f32f64", 256, 600), ("f32f64", 512, 5000), ("f32f64", 2048, 512), ("f64f64", 1024, 512),
                                           ("f64f64", 128, 1000), ("f32f64", 16, 8), ("f32f64", 1024, 0),
                                           # 2N = 2/3/5-smooth: mixed-radix FFT in the kernel (N = 1000 is the reference's test size)
                                           ("f32f64", 1000, 0), ("f64f64", 1000, 700), ("f32f64", 96, 256), ("f32f64", 250, 600),
                                           ("f32f64", 1200, 1000), ("f32f64", 45, 100)])
def test_self_carried_chunks(combo, m, chunk):
    """Every workgroup derives its carry-in from the raw samples (fold by cursor + one 2N-point FFT in LDS) and forms its
    own differences: no pre-pass launch.  Against the oracle, against the pre-pass form, over calls that start at
    cursor 0, mid-period and right before the roll-over, followed by a hop that reads the state the call left."""
    td, fd, fdx = O.combo_types(combo)
    window = "blackman" if (m // 8 + chunk) % 2 else "hann"            # (both windows over the geometries, one per geometry)
    ref = O.best(m, window, 1.0, combo)
    calls = [noise(12000, seed=31, dtype=td), sine_sweep(7001, dtype=td), noise(2 * m - 1 + 4096, seed=32, dtype=td), noise(3 * m + 700, seed=33, dtype=td)]
    with make(m, window, 1.0, combo, chunk=chunk, carry=0) as p, make(m, window, 1.0, combo, chunk=chunk, carry=0, self_carry=0) as q:
        for x in calls:
            want = ref.sdft(x)
            got = p.sdft(x)
            several = p.get_option("last_chunks") > 1
            if m & (m - 1): several = several and p.get_option("last_chunk_len") > 64      # Plan::forward_launch's rule for the mixed-radix form
            assert p.get_option("last_self") == (1 if several else 0)
            old = q.sdft(x)
            assert q.get_option("last_self") == 0
            assert rel_err(got, want) <= 1e-11, rel_err(got, want)
            assert rel_err(got, old) <= 1e-12, rel_err(got, old)
            hop = noise(100, seed=34, dtype=td)
            assert rel_err(p.sdft(hop), ref.sdft(hop)) <= 1e-11
            q.sdft(hop)
            acc, fid, hist, cur = p.state()
            racc, rfid, rhist, rcur = ref.state()
            assert cur == rcur and np.array_equal(hist, rhist)
            assert rel_err(acc, racc) <= 1e-11 and rel_err(fid, rfid) <= 1e-11


def test_self_carried_chunks_batched_and_fused():
    """Batched plans (channels ride on the grid) and the fused call (folded form) in the self-carried form."""
    import torch
    m, n, C = 1024, 12000, 3
    x = sweep_batch(C, n)
    refs = [O.best(m, "hann", 1.0, "f32f64") for _ in range(C)]
    want = [r.sdft(x[c]) for c, r in enumerate(refs)]
    with make(m, "hann", 1.0, "f32f64", channels=C) as p:
        xd = torch.from_numpy(x).cuda()
        d = p.sdft(xd)
        assert p.get_option("last_self") == 1
        got = d.cpu().numpy()
        for c in range(C):
            assert rel_err(got[c], want[c]) <= 1e-11
        x2 = sweep_batch(C, 5000)[:, ::-1].copy()
        got2 = p.sdft(torch.from_numpy(x2).cuda()).cpu().numpy()
        for c in range(C):
            assert rel_err(got2[c], refs[c].sdft(x2[c])) <= 1e-11
    for mm, lat in ((1024, 1.0), (2048, 1.0), (256, 0.5), (1000, 1.0), (240, 1.0)):
        ref = O.best(mm, "hamming", lat, "f32f64")
        xs = noise(30000, seed=41)
        gain = np.linspace(1.0, 0.25, mm)
        with make(mm, "hamming", lat, "f32f64") as p:
            for part in (xs[:17000], xs[17000:]):
                dd = ref.sdft(part)
                want_y = ref.isdft((dd * gain[None, :]).astype(dd.dtype))
                got_y = p.process(torch.from_numpy(part).cuda(), "gain", gain=gain).cpu().numpy()
                # (2N not a power of two: two Stockham buffers have to fit the transpose tiles, else the pre-pass form)
                assert p.get_option("last_self") == (1 if mm & (mm - 1) == 0 or mm == 240 else p.get_option("last_self"))
                assert p.get_option("last_process_path") == 1
                assert rel_err(got_y, want_y) <= 1e-6, rel_err(got_y, want_y)
            p.set_option("self_carry", 0)
            p.reset(); ref.reset()
            dd = ref.sdft(xs)
            want_y = ref.isdft((dd * gain[None, :]).astype(dd.dtype))
            assert rel_err(p.process(xs, "gain", gain=gain), want_y) <= 1e-6
            assert p.get_option("last_self") == 0


def test_float_plans_with_chunk_parallel_carries():
    """Option float_carry_parallel: FD float plans take the chunk-parallel carries.  The float reference drifts from the
    double-precision result by its own rounding (about 2e-4 of the largest bin per 262144 samples), so no path other
    than the bit-exact one stays within 1e-4 of it on long calls; what this option promises instead is checked here:
    closer to the double-precision reference than the float reference is, and within 1e-4 of it."""
    for m, window, n in ((1024, "hann", 60000), (4096, "blackman", 12000), (1000, "hamming", 20000), (256, "boxcar", 20000)):
        x = noise(n, seed=71) if m != 1024 else sine_sweep(n)
        ref32, ref64 = O.best(m, window, 1.0, "f32f32"), O.best(m, window, 1.0, "f32f64")
        want32, truth = ref32.sdft(x), ref64.sdft(x)
        with make(m, window, 1.0, "f32f32") as p:
            assert p.get_option("carry") == 1
            p.set_option("carry", 0)
            assert p.get_option("carry") == 1                       # FD float stays exact without the explicit option
            p.set_option("float_carry_parallel", 1)
            assert p.get_option("carry") == 0
            got = p.sdft(x)
            assert p.get_option("last_chunks") > 1 and p.get_option("last_chain") == 0
            assert rel_err(got, truth) <= 1e-4, (m, rel_err(got, truth))
            assert rel_err(got, truth) <= rel_err(want32, truth), (m, rel_err(got, truth), rel_err(want32, truth))
            hop = noise(300, seed=72)
            got2 = p.sdft(hop)
            assert rel_err(got2, ref64.sdft(hop)) <= 1e-4
            ref32.sdft(hop)
            # the fused call follows the option too
            part = noise(6000, seed=73)
            y = p.process(part, "identity")
            assert rel_err(y, ref64.isdft(ref64.sdft(part))) <= 1e-4
            # and back: a reset plan without the option is bit-identical again
            p.set_option("float_carry_parallel", 0)
            assert p.get_option("carry") == 1
            p.reset(); ref32.reset()
            assert np.array_equal(p.sdft(x[:20000]), ref32.sdft(x[:20000]))


def test_exact_carry_relay_flow_mode():
    """Flow mode (default with the relay form): one relay launch on the auxiliary stream, one forward launch whose
    workgroups wait for their chunk's carries (time-major numbering for batched plans), against the segmented form and
    the oracle; the fused call takes it too."""
    import torch
    for combo, m, C, n in (("f32f32", 1024, 1, 60000), ("f32f32", 256, 3, 20000), ("f32f64", 512, 2, 30000), ("f32f32", 4096, 1, 20000)):
        td, fd, fdx = O.combo_types(combo)
        xb = np.stack([noise(n, seed=90 + c, dtype=td) for c in range(C)])
        x2 = np.stack([noise(5000 + 77, seed=95 + c, dtype=td) for c in range(C)])
        refs = [O.best(m, "hamming", 1.0, combo) for _ in range(C)]
        want = [r.sdft(xb[c]) for c, r in enumerate(refs)]
        want2 = [r.sdft(x2[c]) for c, r in enumerate(refs)]
        for flow in (1, 0):
            with make(m, "hamming", 1.0, combo, C, carry=1, chain=2, relay_flow=flow) as p:
                got = p.sdft(torch.from_numpy(xb if C > 1 else xb[0]).cuda()).cpu().numpy()
                assert p.get_option("last_chain") == 3 and p.get_option("last_flow") == flow, (combo, m, p.get_option("last_chain"), p.get_option("last_flow"))
                got2 = p.sdft(x2 if C > 1 else x2[0])            # second call: mid-block start, host pointers
                assert p.get_option("last_flow") == flow
                for c in range(C):
                    assert np.array_equal(got[c] if C > 1 else got, want[c]), (combo, m, flow, c)
                    assert np.array_equal(got2[c] if C > 1 else got2, want2[c]), (combo, m, flow, c)
    # the fused call (exact carries at FD float: folded form on top of the relay's carries)
    m, n = 1024, 40000
    x = noise(n, seed=99)
    ref = O.best(m, "hann", 1.0, "f32f32")
    d = ref.sdft(x)
    with make(m, "hann", 1.0, "f32f32") as p:
        y = p.process(torch.from_numpy(x).cuda(), "identity").cpu().numpy()
        assert p.get_option("last_flow") == 1 and p.get_option("last_chain") == 3
        assert rel_err(y, ref.isdft(d)) <= 1e-4
        acc, fid, hist, cur = p.state()
        racc, rfid, rhist, rcur = ref.state()
        assert cur == rcur and np.array_equal(acc, racc) and np.array_equal(fid, rfid)


@pytest.mark.parametrize("combo,m,window", [("f32f32", 128, "boxcar"), ("f32f32", 128, "blackman"), ("f32f32", 1024, "hann"), ("f64f32", 256, "hamming"),
                                            ("f32f32", 2048, "blackman"), ("f32f32", 4096, "hann"), ("f32f32", 4096, "blackman"), ("f64f32", 1024, "boxcar")])
def test_bin_pair_kernel_is_the_generic_kernel_bit_for_bit(combo, m, window):
    """forward_rows_f32_kernel (round 4: the lane's two adjacent bins are the halves of every packed operand) against
    forward_rows_kernel<float, 2, ...> (option rows_f32 = 0) and the oracle: calls that cross the roll-over, start
    mid-block, batched channels, the state they leave, and the chunk-parallel carries (float_carry_parallel), where the
    two kernels must agree with each other."""
    td, fd, fdx = O.combo_types(combo)
    C = 2
    lens = (6 * m + 4101, 2 * m + 3000 + 7, 4096) if m < 2048 else (2 * m + 2101, 2 * m + 1000 + 7, 2048)   # (the oracle at m = 4096: 40 ns per bin-sample)
    xb = np.stack([noise(sum(lens), seed=90 + c, dtype=td) for c in range(C)])
    refs = [O.best(m, window, 1.0, combo) for _ in range(C)]
    with make(m, window, 1.0, combo, C) as p, make(m, window, 1.0, combo, C, rows_f32=0) as q:
        i = 0
        for n in lens:
            seg = np.ascontiguousarray(xb[:, i:i + n])
            got, old = p.sdft(seg), q.sdft(seg)
            assert p.get_option("last_rows_f32") == 1 and q.get_option("last_rows_f32") == 0 and p.get_option("last_chunks") > 1
            for c in range(C):
                assert np.array_equal(got[c], refs[c].sdft(seg[c])), (combo, m, window, n, c)
            assert np.array_equal(got, old)
            i += n
        acc, fid, hist, cur = p.state()
        for c in range(C):
            racc, rfid, rhist, rcur = refs[c].state()
            assert cur == rcur and np.array_equal(acc[c], racc) and np.array_equal(fid[c], rfid) and np.array_equal(hist[c], rhist)
    x = xb[0, :min(5 * m + 3000, xb.shape[1])]
    with make(m, window, 1.0, combo, float_carry_parallel=1) as p, make(m, window, 1.0, combo, float_carry_parallel=1, rows_f32=0) as q:
        assert np.array_equal(p.sdft(x), q.sdft(x)) and p.get_option("last_rows_f32") == 1 and p.get_option("last_chain") == 0


@pytest.mark.parametrize("combo,m,n", [("f64f64", 1000, 30000), ("f32f64", 1024, 20000), ("f32f32", 4096, 9000), ("f32f64", 512, 700)])
def test_streaming_loads_change_no_bit(combo, m, n):
    """Option inverse_nt (non-temporal loads of the matrix in the synthesis kernels; by default chosen by the matrix size)
    is a cache policy, not arithmetic: same bits either way, and the oracle's."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    ref = O.best(m, "hann", 0.5, combo)
    d = ref.sdft(noise(n, seed=5, dtype=td))
    want = ref.isdft(d)
    dd = torch.from_numpy(d).cuda()
    outs = []
    for nt in (0, 1, -1):
        with make(m, "hann", 0.5, combo, inverse_nt=nt) as p:
            outs.append(p.isdft(dd).cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and np.array_equal(outs[0], want)
    # round 5: the rows read first take ordinary loads whatever the kind of load (option inverse_nt_skip_mb: here the first 3 MB read, so that a call has
    # rows of both kinds in every form -- streaming forms of 4 / 16 rows per wave, the tree sum, whole rows in step)
    for opts in ({"inverse_rows": 4}, {"inverse_rows": 16}, {"inverse_tune": 0}, {"inverse_step": 1}):
        with make(m, "hann", 0.5 if "inverse_step" not in opts else 1.0, combo, inverse_nt=1, inverse_nt_skip_mb=3, **opts) as p:
            if "inverse_step" in opts:
                ref1 = O.best(m, "hann", 1.0, combo)
                assert np.array_equal(p.isdft(dd).cpu().numpy(), ref1.isdft(d))
            else:
                assert np.array_equal(p.isdft(dd).cpu().numpy(), want)
            assert p.get_option("last_inverse_nt") == 1 and p.get_option("last_inverse_skip") > 0


@pytest.mark.gpu
@pytest.mark.parametrize("combo", ["f32f64", "f64f64", "f32f32", "f64f32"])
@pytest.mark.parametrize("m,n,channels,latency", [(1024, 70000, 1, 1.0), (1000, 9001, 1, 1.0), (1024, 5003, 2, 0.5), (320, 20001, 3, 1.0), (64, 70, 1, 1.0),
                                                  (6, 777, 1, 1.0), (2048, 4097, 1, 1.0), (998, 6001, 1, 0.25)])
def test_synthesis_by_whole_rows_with_the_ordered_sum_keeps_the_bits(combo, m, n, channels, latency):
    """Round 6: every type pair and latency can be synthesised by workgroups that read WHOLE ROWS of a chunk of the matrix while ONE wave
    adds the terms in the reference's order, a lane per row (inverse_rows_ordered_kernel: loader waves, row slots in LDS, two counters
    per group of slots).  Forced (inverse_ordered = 1) on shapes that exercise full and partly filled pieces, rows padded to whole blocks,
    ragged chunk ends, channels, a matrix of 70 rows and rows too long for the form (2048 double bins: the form does not apply and the
    call takes another): the bits of the streaming forms and of the reference."""
    import torch
    td, fd, fdx = O.combo_types(combo)
    x = noise(n * channels, seed=500 + m, dtype=td).reshape(channels, n) if channels > 1 else noise(n, seed=500 + m, dtype=td)
    applies = m * np.dtype(fd).itemsize <= 8192
    with make(m, "hann", latency, combo, channels=channels) as p:
        if combo.endswith("f32"): p.set_option("float_carry_parallel", 1)      # (only to fill the matrix quickly: the synthesis is what is tested)
        d = p.sdft(torch.from_numpy(x).cuda())
        p.set_option("inverse_ordered", -1); p.set_option("inverse_tune", 0)
        y0 = p.isdft(d).cpu().numpy()
        assert p.get_option("last_inverse_form") != 4
        p.set_option("inverse_ordered", 1)
        y1 = p.isdft(d).cpu().numpy()
        assert (p.get_option("last_inverse_form") == 4) == applies
        assert np.array_equal(y0.view(np.uint8), y1.view(np.uint8))
        ref = O.best(m, "hann", latency, combo)
        dh = d.cpu().numpy()
        rows = dh if channels == 1 else dh[channels - 1]
        want = ref.isdft(rows[: min(n, 3000)])
        got = y1 if channels == 1 else y1[channels - 1]
        assert np.array_equal(got[: len(want)], want)
        # both kinds of load in one call (the rows read first take ordinary loads), and a matrix whose rows do not start on 16 bytes
        p.set_option("inverse_nt", 1); p.set_option("inverse_nt_skip_mb", 1)
        y2 = p.isdft(d).cpu().numpy()
        assert np.array_equal(y0.view(np.uint8), y2.view(np.uint8))
        if channels == 1 and combo.endswith("f32") and n > 100:
            flat = torch.empty(n * m + 1, dtype=d.dtype, device="cuda")
            off = flat[1:].view(n, m)                           # 8 bytes past a 16-byte boundary
            off.copy_(d)
            y3 = p.isdft(off).cpu().numpy()
            assert p.get_option("last_inverse_form") != 4
            assert np.array_equal(y0.view(np.uint8), y3.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,channels", [(1024, 70000, 1), (1000, 9001, 1), (2048, 5003, 2), (320, 20001, 3), (64, 70, 1), (1536, 4097, 1)])
def test_synthesis_by_whole_rows_in_step_keeps_the_bits(m, n, channels):
    """Round 5: float samples from double bins at latency 1 can be synthesised by workgroups that read WHOLE ROWS of a chunk of the
    matrix in step (inverse_rows_body: one bin per lane, tree sums through the wave and through LDS, the rounding-interval proof,
    rows on a rounding boundary added again in the reference's order).  Forced (inverse_step = 1) on shapes that exercise one and
    two bins per lane, partly filled waves, ragged chunk ends, channels, and a matrix of 64 rows: the bits of the streaming form
    and of the reference."""
    import torch
    x = noise(n * channels, seed=400 + m).reshape(channels, n) if channels > 1 else noise(n, seed=400 + m)
    with make(m, "hann", 1.0, "f32f64", channels=channels) as p:
        d = p.sdft(torch.from_numpy(x).cuda())
        p.set_option("inverse_step", -1); p.set_option("inverse_tune", 0)
        y0 = p.isdft(d).cpu().numpy()
        assert p.get_option("last_inverse_form") != 3
        p.set_option("inverse_step", 1)
        y1 = p.isdft(d).cpu().numpy()
        assert p.get_option("last_inverse_form") == 3
        assert np.array_equal(y0.view(np.uint32), y1.view(np.uint32))
        ref = O.best(m, "hann", 1.0, "f32f64")
        dh = d.cpu().numpy()
        rows = dh if channels == 1 else dh[channels - 1]
        want = ref.isdft(rows[: min(n, 3000)])
        got = y1 if channels == 1 else y1[channels - 1]
        assert np.array_equal(got[: len(want)], want)
        # other latencies and type pairs do not take the form
        p.set_option("inverse_step", 1)
    with make(m, "hann", 0.5, "f32f64") as q:
        q.set_option("inverse_step", 1)
        q.isdft(q.sdft(torch.from_numpy(noise(2000, seed=5)).cuda()))
        assert q.get_option("last_inverse_form") != 3
