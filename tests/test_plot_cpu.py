"""scripts/plot_dump.py: the spectrogram of a DFT dump (the counterpart of the reference's test/plot.py:27-68)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import oracle as O
from sdft_amd.signals import sine_sweep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spectrogram_of_a_dump(tmp_path):
    pytest.importorskip("matplotlib")
    spec = importlib.util.spec_from_file_location("plot_dump", os.path.join(ROOT, "scripts", "plot_dump.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m, hop, sr = 64, 16, 8000.0
    x = sine_sweep(2048)
    ref = O.best(m, "hann", 1.0, "f32f64")
    rows = ref.sdft(x)[::hop]                                 # the first row of every hop, as examples/sdft_wav dumps them
    dump = tmp_path / "sweep.dfts"
    rows.astype(np.complex128).tofile(dump)
    back = mod.load(str(dump), m)
    assert back.shape == rows.shape and np.array_equal(back, rows)
    db = mod.decibels(back, -120.0)
    assert db.shape == rows.shape and db.min() >= -120.0 and np.isfinite(db).all()
    # the sweep's ridge moves up in frequency with time
    first, last = int(np.argmax(db[2])), int(np.argmax(db[-2]))
    assert last > first
    png = tmp_path / "sweep.png"
    assert mod.main([str(dump), "--dftsize", str(m), "--sr", str(sr), "--hop", str(hop), "-o", str(png)]) == 0
    assert png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n" and png.stat().st_size > 2000
    with pytest.raises(SystemExit):
        mod.load(str(dump), m + 1)
