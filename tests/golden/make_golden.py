"""Generates the golden vectors under tests/golden/ from the GENUINE reference.

Run in the build container only (needs /root/reference):

    make -C oracle            # compiles the reference header into oracle/_ref/
    python tests/golden/make_golden.py

Every fixture is data: inputs (or the seed/generator that makes them), the reference's outputs
and the parameters.  The reference publishes no golden vectors of its own;
its only committed test data file is test/test.wav, whose first samples are included as the
input of the `testwav` fixture (PCM24 mono, decoded like the reference's test/wav.py:24-26).

Fixture layout (npz):  params..., x (input), y (full synthesis output), rows_idx + rows (selected
DFT rows; all rows for tiny cases), digest (per-row checksums for all rows: sum re, sum im,
sum |.|^2, sum (k+1) re).
"""

import os
import sys
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O                      # noqa: E402
from sdft_amd.signals import noise, sine_sweep      # noqa: E402


def digest_of(d, block=4096):
    """per-row checksums (sum re, sum im, sum |.|^2, sum (k+1) re), computed in row blocks to keep the
    temporaries small"""
    k = np.arange(1, d.shape[1] + 1, dtype=np.float64)
    out = np.empty((d.shape[0], 4), dtype=np.float64)
    for i in range(0, d.shape[0], block):
        re = d[i:i + block].real.astype(np.float64); im = d[i:i + block].imag.astype(np.float64)
        out[i:i + block] = np.stack([re.sum(1), im.sum(1), (re * re + im * im).sum(1), (re * k).sum(1)], axis=1)
    return out


def select_rows(n, m):
    idx = set(range(0, min(8, n)))
    for c in (2 * m, 4 * m):
        idx |= {t for t in range(c - 3, c + 3) if 0 <= t < n}
    idx |= set(range(max(0, n - 8), n))
    return np.array(sorted(idx), dtype=np.int64)


def run_reference(m, window, latency, combo, x, hop=None):
    ref = O.Reference(m, window, latency, combo)
    if hop is None:
        d = ref.sdft(x)
    else:
        d = np.concatenate([ref.sdft(x[i:i + hop]) for i in range(0, x.size, hop)])
    y = ref.isdft(d)
    return d, y


def save(name, m, window, latency, combo, x, d, y, full=False, extra=None):
    idx = np.arange(d.shape[0]) if full else select_rows(d.shape[0], m)
    out = dict(dftsize=m, window=window, latency=latency, combo=combo, x=x, y=y,
               rows_idx=idx, rows=d[idx], digest=digest_of(d))
    if extra:
        out.update(extra)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: n={x.size} m={m} {window} lat={latency} {combo} -> {os.path.getsize(path) / 1024:.0f} KiB")


def read_pcm24(path, count):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 3 and w.getnchannels() == 1
        raw = w.readframes(count)
    b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
    v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    v = np.where(v >= (1 << 23), v - (1 << 24), v)
    return ((v + 0.5) / ((1 << 23) - 0.5)).astype(np.float32)      # reference test/wav.py:24-26


def main():
    assert O.have_reference(), "build oracle/_ref first (make -C oracle)"
    # 1. tiny exhaustive cases: every combo x window x latency, full matrices, packed in one file
    #    under keys "<combo>/<window>/<latency*100>/<m>/{x,d,y}"
    tiny = {}
    for combo in O.COMBOS:
        td = O.combo_types(combo)[0]
        for window in ("boxcar", "hann", "hamming", "blackman"):
            for latency in (1.0, 0.5, 0.25):
                for m in (1, 2, 3, 5, 8):
                    x = noise(5 * m + 3, seed=100 + m, dtype=td)
                    d, y = run_reference(m, window, latency, combo, x)
                    key = f"{combo}/{window}/{int(latency * 100)}/{m}"
                    tiny[key + "/x"], tiny[key + "/d"], tiny[key + "/y"] = x, d, y
    np.savez_compressed(os.path.join(HERE, "tiny_cases.npz"), **tiny)
    print(f"tiny_cases: {len(tiny) // 3} cases")
    # 2. mid-size, incl. roll-over at t = 2N-1 and ragged streaming (state persistence)
    for combo in O.COMBOS:
        td = O.combo_types(combo)[0]
        for window, m in (("hann", 64), ("blackman", 100), ("hamming", 125)):
            x = noise(5 * m + 11, seed=7, dtype=td)
            d, y = run_reference(m, window, 1.0, combo, x, hop=37)
            save(f"mid_{combo}_{window}_m{m}", m, window, 1.0, combo, x, d, y, extra=dict(hop=37))
    # 3. BASELINE config 1 / 2 shape: m=1024, Hann, TD float, FD double
    x = sine_sweep(48000)
    d, y = run_reference(1024, "hann", 1.0, "f32f64", x)
    save("cfg1_sweep48000_m1024_hann_f32f64", 1024, "hann", 1.0, "f32f64", x, d, y)
    # 4. BASELINE config 3 shape at parity size: m=4096, Blackman, FD float
    x = sine_sweep(12000)
    d, y = run_reference(4096, "blackman", 1.0, "f32f32", x)
    save("cfg3_sweep12000_m4096_blackman_f32f32", 4096, "blackman", 1.0, "f32f32", x, d, y)
    # 5. BASELINE config 4 shape, one channel: m=2048, Hann
    x = sine_sweep(12000, channel=3, channels=64)
    d, y = run_reference(2048, "hann", 1.0, "f32f64", x)
    save("cfg4_sweep12000_m2048_hann_f32f64", 2048, "hann", 1.0, "f32f64", x, d, y, extra=dict(channel=3, channels=64))
    # 6. the reference's own test: test.wav, m=1000, hop=100, Hann, latency 1 (test/main.sh:3-6);
    #    first DFT row of each hop like test/test.c:82
    wav = os.path.join(os.environ.get("SDFT_REF_DIR", "/root/reference"), "test", "test.wav")
    x = read_pcm24(wav, 11000)
    hop = 100
    ref = O.Reference(1000, "hann", 1.0, "f32f64")
    firsts, ys = [], []
    for i in range(0, x.size, hop):
        d = ref.sdft(x[i:i + hop])
        ys.append(ref.isdft(d))
        firsts.append(d[0])
    path = os.path.join(HERE, "testwav_m1000_hop100_hann_f32f64.npz")
    np.savez_compressed(path, dftsize=1000, window="hann", latency=1.0, combo="f32f64", hop=hop, x=x,
                        y=np.concatenate(ys), hop_first_rows=np.stack(firsts))
    print(f"testwav: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
