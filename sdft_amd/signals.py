"""Deterministic synthetic inputs (no RNG state needed on either side of a parity check).

The sweep follows the phase-accumulated chirp of the reference's example
(/root/reference/python/examples/analysis.py:11-15,29): instantaneous frequency rising
linearly from 0 to Nyquist, phase accumulated in double, samples narrowed to the
time-domain type last.
"""

from __future__ import annotations

import numpy as np


def sine_sweep(n: int, sr: float = 48000.0, channel: int = 0, channels: int = 1, dtype=np.float32) -> np.ndarray:
    """Mono linear sine sweep 0 -> sr/2 over n samples, amplitude 1.

    For batches, channel c of C starts at phase 2*pi*c/C and ends at sr/2*(1 - c/(2C)) so that
    every channel is distinct.
    """
    n = int(n)
    if n == 0:
        return np.zeros(0, dtype=dtype)
    i = np.arange(n, dtype=np.float64)
    f_end = 0.5 * sr * (1.0 - channel / (2.0 * channels))
    f = (i / n) * f_end
    phi = 2.0 * np.pi * channel / channels + np.cumsum(2.0 * np.pi * f / sr)
    return np.sin(phi).astype(dtype)


def sweep_batch(channels: int, n: int, sr: float = 48000.0, dtype=np.float32) -> np.ndarray:
    """(channels, n) batch of distinct sweeps."""
    return np.stack([sine_sweep(n, sr, c, channels, dtype) for c in range(channels)])


def noise(n: int, seed: int = 20240601, dtype=np.float32) -> np.ndarray:
    """Uniform noise in [-1, 1) for round-trip SNR checks (the reference's latency example uses
    unseeded truncated normal noise, /root/reference/python/examples/latency.py:11-14)."""
    return np.random.default_rng(seed).uniform(-1.0, 1.0, int(n)).astype(dtype)
