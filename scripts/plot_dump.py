"""Spectrogram of a DFT dump (what /root/reference/test/plot.py:27-68 draws for the reference's dumps): the file holds one
row of `dftsize` interleaved complex doubles per hop, as `examples/sdft_wav` (and the reference's test/dump.h) writes them.

    python scripts/plot_dump.py dump.dfts --dftsize 1000 --sr 44100 --hop 100 -o dump.png [--floor -120] [--log]

Magnitudes in dB (20 log10 |X|, clipped at --floor), time in seconds across, frequency 0 ... sr/2 up.  Needs matplotlib;
draws off-screen (Agg) unless --show is given."""
import argparse
import sys

import numpy as np


def load(path, dftsize, complex_dtype=np.complex128):
    raw = np.fromfile(path, dtype=complex_dtype)
    if dftsize <= 0 or raw.size == 0 or raw.size % dftsize:
        raise SystemExit(f"{path}: {raw.size} complex values are not whole rows of {dftsize} bins")
    return raw.reshape(-1, dftsize)


def decibels(rows, floor):
    mag = np.abs(rows)
    out = np.full(mag.shape, float(floor))
    nz = mag > 0
    out[nz] = np.maximum(20.0 * np.log10(mag[nz]), floor)
    return out


def draw(rows, sr, hop, floor=-120.0, log=False, title=None):
    import matplotlib.pyplot as plt
    db = decibels(rows, floor)
    hops, bins = db.shape
    t1 = (hops - 1) * hop / sr if hops > 1 else hop / sr
    fig, ax = plt.subplots(figsize=(10, 5))
    im = ax.imshow(db.T, origin="lower", aspect="auto", cmap="inferno", interpolation="nearest", extent=(0.0, t1, 0.0, sr / 2.0), vmin=floor, vmax=0.0)
    ax.set_xlabel("s")
    ax.set_ylabel("Hz")
    if log:
        ax.set_yscale("symlog", linthresh=max(sr / 2.0 / bins, 1.0))
    if title:
        ax.set_title(title)
    fig.colorbar(im, ax=ax).set_label("dB")
    fig.tight_layout()
    return fig


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("dump")
    ap.add_argument("--dftsize", type=int, required=True)
    ap.add_argument("--sr", type=float, default=44100.0)
    ap.add_argument("--hop", type=int, default=1, help="samples between two dumped rows")
    ap.add_argument("--floor", type=float, default=-120.0)
    ap.add_argument("--float", dest="single", action="store_true", help="the dump holds complex floats (FD float builds)")
    ap.add_argument("--log", action="store_true", help="logarithmic frequency axis")
    ap.add_argument("--show", action="store_true")
    ap.add_argument("-o", "--output", default=None)
    a = ap.parse_args(argv)
    if not a.show:
        import matplotlib
        matplotlib.use("Agg")
    rows = load(a.dump, a.dftsize, np.complex64 if a.single else np.complex128)
    fig = draw(rows, a.sr, a.hop, a.floor, a.log, title=f"{a.dump}: {rows.shape[0]} rows of {rows.shape[1]} bins")
    out = a.output or (a.dump + ".png")
    fig.savefig(out, dpi=120)
    print(f"{out}: {rows.shape[0]} x {rows.shape[1]}")
    if a.show:
        import matplotlib.pyplot as plt
        plt.show()
    return 0


if __name__ == "__main__":
    sys.exit(main())
