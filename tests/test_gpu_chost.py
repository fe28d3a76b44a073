"""A plain C host (tests/c/host_stream.c, the calling pattern of the reference's test/test.c)
compiled with gcc against include/sdft/sdft.h and linked with libsdft_hip.so, run on the GPU and
compared with the oracle.  Covers dense, row-pointer and single-sample entry points with host
pointers, for the default types and for FD float."""

import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from sdft_amd.signals import noise

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hip_runtime_dir():
    for d in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib"),):
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    pytest.skip("system HIP runtime not found")


@pytest.mark.parametrize("flags,combo", [([], "f32f64"), (["-DSDFT_FD_FLOAT"], "f32f32"),
                                         (["-DSDFT_TD_DOUBLE", "-DSDFT_NO_COMPLEX_H"], "f64f64")])
def test_c_host_streaming(tmp_path, hip_library, flags, combo):
    td, fd, fdx = O.combo_types(combo)
    libdir = os.path.dirname(hip_library)
    rt = hip_runtime_dir()
    exe = tmp_path / "host_stream"
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), *flags,
           os.path.join(ROOT, "tests", "c", "host_stream.c"), "-o", str(exe),
           "-L", libdir, "-lsdft_hip", "-L", rt, "-lamdhip64", "-lm",
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rt}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    m, hop, hops = 100, 50, 12
    x = noise(hop * hops, seed=3, dtype=td)
    x.tofile(tmp_path / "x.raw")
    r = subprocess.run([str(exe), str(m), str(hop), "hann", "1", str(tmp_path / "x.raw"),
                        str(tmp_path / "y.raw"), str(tmp_path / "d.raw")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "C-HOST ok" in r.stdout

    ref = O.best(m, "hann", 1.0, combo)
    firsts, ys = [], []
    for i in range(0, x.size, hop):
        d = ref.sdft(x[i:i + hop])
        firsts.append(d[0]); ys.append(ref.isdft(d))
    got_d = np.fromfile(tmp_path / "d.raw", dtype=fdx).reshape(hops, m)
    got_y = np.fromfile(tmp_path / "y.raw", dtype=td)
    assert np.array_equal(got_d, np.stack(firsts))                 # hops < 512: serial, bit-identical
    tol = 1e-6 if combo.endswith("f64") else 1e-4
    want_y = np.concatenate(ys)
    assert np.abs(got_y - want_y).max() <= tol * np.abs(want_y).max()


@pytest.mark.parametrize("flags,combo", [([], "f32f64"), (["-DSDFT_FD_FLOAT"], "f32f32")])
@pytest.mark.parametrize("op", [0, 1, 2, 3])
def test_c_host_fused_process(tmp_path, hip_library, flags, combo, op):
    """tests/c/host_process.c: sdft_hip_process_n from a gcc-built C host (hops of 100 and calls of 4000
    samples) against the same host running the reference's three steps through the drop-in
    functions, and against the oracle."""
    td, fd, fdx = O.combo_types(combo)
    libdir = os.path.dirname(hip_library)
    rt = hip_runtime_dir()
    exe = tmp_path / "host_process"
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), *flags,
           os.path.join(ROOT, "tests", "c", "host_process.c"), "-o", str(exe),
           "-L", libdir, "-lsdft_hip", "-L", rt, "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rt}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = 256
    x = noise(8000, seed=4, dtype=td)
    x.tofile(tmp_path / "x.raw")
    gain = (1.0 / (1.0 + np.arange(m) / 64.0)).astype(fd)
    for hop in (100, 4000):
        r = subprocess.run([str(exe), str(m), str(hop), str(op), str(tmp_path / "x.raw"), str(tmp_path / "y1.raw"),
                            str(tmp_path / "y2.raw")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "C-PROCESS ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
        y1 = np.fromfile(tmp_path / "y1.raw", dtype=td)
        y2 = np.fromfile(tmp_path / "y2.raw", dtype=td)
        ref = O.best(m, "hann", 1.0, combo)
        want = []
        for i in range(0, x.size, hop):
            d = ref.sdft(x[i:i + hop])
            if op == 1:
                d = (d * gain[None, :]).astype(fdx)
            if op == 2:
                s = np.zeros_like(d); s[:, 3:] = d[:, :m - 3]; d = s
            if op == 3:                                       # gain * i^k: a multiplication by 1, i, -1, -i is exact
                cg = (gain.astype(np.float64) * (1j ** (np.arange(m) % 4))).astype(fdx)
                d = (d * cg[None, :]).astype(fdx)
            want.append(ref.isdft(d))
        want = np.concatenate(want)
        tol = 1e-6 if combo.endswith("f64") else 1e-4
        assert np.abs(y2 - want).max() <= tol * np.abs(want).max()
        assert np.abs(y1 - want).max() <= tol * np.abs(want).max()
        if hop < 512:
            assert np.array_equal(y2, want)                                 # the two calls, one time chunk: bit-identical


@pytest.mark.parametrize("t,f,combo", [("float", "double", "f32f64"), ("double", "double", "f64f64"), ("float", "float", "f32f32")])
def test_cpp_facade_host(tmp_path, hip_library, t, f, combo):
    """C++ host using sdft::SDFT<T, F> (include/sdft/sdft.hpp, the reference's C++ interface) in the
    shape of the reference's cpp/examples/bench.cpp, compared with the oracle."""
    td, fd, fdx = O.combo_types(combo)
    libdir = os.path.dirname(hip_library)
    rt = hip_runtime_dir()
    exe = tmp_path / "host_bench"
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", f"-DHOST_T={t}", f"-DHOST_F={f}", "-I", os.path.join(ROOT, "include", "cpp"),
           os.path.join(ROOT, "tests", "cpp", "host_bench.cpp"), "-o", str(exe),
           "-L", libdir, "-lsdft_hip", "-L", rt, "-lamdhip64", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rt}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m, n = 1000, 441                      # hops below 512 samples: serial path, bit-identical
    x = noise(n, seed=9, dtype=td)
    x.tofile(tmp_path / "x.raw")
    r = subprocess.run([str(exe), str(m), "1", "1", str(tmp_path / "x.raw"), str(tmp_path / "y.raw"), str(tmp_path / "d.raw")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CPP-HOST ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    ref = O.best(m, "hann", 1.0, combo)
    want = ref.sdft(x)
    got = np.fromfile(tmp_path / "d.raw", dtype=fdx).reshape(n, m)
    assert np.array_equal(got, want)
    want_y = ref.isdft(want)
    got_y = np.fromfile(tmp_path / "y.raw", dtype=td)
    tol = 1e-6 if combo.endswith("f64") else 1e-4
    assert np.abs(got_y - want_y).max() <= tol * np.abs(want_y).max()


def test_wav_tool_end_to_end(tmp_path, hip_library):
    """examples/sdft_wav.c = the reference's test driver (test/test.c) as a tool: PCM24 WAV in (the
    reference's test.wav excerpt kept in the golden fixture), hop loop on the GPU, float WAV + DFT
    dump out; compared with the oracle fed the same decoded samples."""
    import struct
    import wave
    libdir = os.path.dirname(hip_library)
    rt = hip_runtime_dir()
    exe = tmp_path / "sdft_wav"
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "sdft_wav.c"),
           "-o", str(exe), "-L", libdir, "-lsdft_hip", "-L", rt, "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rt}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    g = np.load(os.path.join(ROOT, "tests", "golden", "testwav_m1000_hop100_hann_f32f64.npz"))
    pcm = np.round(g["x"].astype(np.float64) * ((1 << 23) - 0.5) - 0.5).astype(np.int64)     # inverse of wav.py:24-26
    pcm = pcm[:5000]
    with wave.open(str(tmp_path / "in.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(3); w.setframerate(44100)
        w.writeframes(b"".join(struct.pack("<i", int(v))[:3] for v in pcm))
    r = subprocess.run([str(exe), "1000", "100", "hann", "1", str(tmp_path / "in.wav"), str(tmp_path / "out.wav"), str(tmp_path / "out.dft")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "5000 44100Hz" in r.stdout, (r.stdout, r.stderr)

    x = (pcm / 8388608.0).astype(np.float32)                # the tool's (dr_wav-style) scaling
    ref = O.best(1000, "hann", 1.0, "f32f64")
    firsts, ys = [], []
    for i in range(0, x.size, 100):
        d = ref.sdft(x[i:i + 100])
        firsts.append(d[0]); ys.append(ref.isdft(d))
    got_d = np.fromfile(tmp_path / "out.dft", dtype=np.complex128).reshape(-1, 1000)
    assert np.array_equal(got_d, np.stack(firsts))
    raw = open(tmp_path / "out.wav", "rb").read()            # IEEE-float WAV (python's wave module reads PCM only)
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data"
    fmt, nch, rate, _, _, bits = struct.unpack("<HHIIHH", raw[20:36])
    assert (fmt, nch, rate, bits) == (3, 1, 44100, 32)
    got_y = np.frombuffer(raw[44:], dtype=np.float32)
    assert np.array_equal(got_y, np.concatenate(ys))


def test_stream_filter_example(tmp_path, hip_library):
    """examples/stream_filter.c: the reference's hop loop with a low-pass mask as ONE fused call per hop.  A two-tone
    signal goes in as 16-bit PCM; what comes out must equal the oracle's three steps (analysis, the mask applied by
    the host, synthesis) within the path's bar, with the tone above the cut-off gone and the other one kept."""
    import struct
    import wave
    libdir = os.path.dirname(hip_library)
    rt = hip_runtime_dir()
    exe = tmp_path / "stream_filter"
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "stream_filter.c"),
           "-o", str(exe), "-L", libdir, "-lsdft_hip", "-L", rt, "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rt}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rate, n, m, hop, cutoff = 44100, 12000, 1000, 100, 2000.0
    t = np.arange(n) / rate
    sig = 0.4 * np.sin(2 * np.pi * 500.0 * t) + 0.4 * np.sin(2 * np.pi * 8000.0 * t)
    pcm = np.round(sig * 32767.0).astype(np.int16)
    with wave.open(str(tmp_path / "in.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(pcm.tobytes())
    r = subprocess.run([str(exe), str(m), str(hop), str(cutoff), str(tmp_path / "in.wav"), str(tmp_path / "out.wav")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "low-pass 2000 Hz" in r.stdout, (r.stdout, r.stderr)
    raw = open(tmp_path / "out.wav", "rb").read()
    got = np.frombuffer(raw[44:], dtype=np.float32)
    assert got.size == n

    x = (pcm / 32768.0).astype(np.float32)                  # the tool's (dr_wav-style) scaling
    hz = np.arange(m) * rate / (2.0 * m)
    edge = cutoff * 1.4142135623730951
    mask = np.where(hz <= cutoff, 1.0, np.where(hz >= edge, 0.0, 0.5 * (1.0 + np.cos(np.pi * (hz - cutoff) / (edge - cutoff)))))
    ref = O.best(m, "hann", 1.0, "f32f64")
    want = np.concatenate([ref.isdft((ref.sdft(x[i:i + hop]) * mask[None, :])) for i in range(0, n, hop)])
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    # the spectrum of the settled part: 500 Hz kept, 8 kHz gone
    tail = got[4000:].astype(np.float64) * np.hanning(n - 4000)
    spec = np.abs(np.fft.rfft(tail))
    f = np.fft.rfftfreq(n - 4000, 1.0 / rate)
    low, high = spec[np.abs(f - 500.0).argmin()], spec[np.abs(f - 8000.0).argmin()]
    assert high < 1e-4 * low, (low, high)
    # the same mask handed in as code from C (sdft_hip_op_expr; sdft_hip_expr_t crosses the C-ABI): the same samples
    r = subprocess.run([str(exe), str(m), str(hop), str(cutoff), str(tmp_path / "in.wav"), str(tmp_path / "out2.wav"), "code"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    got2 = np.frombuffer(open(tmp_path / "out2.wav", "rb").read()[44:], dtype=np.float32)
    assert got2.size == n and np.abs(got2 - want).max() <= 1e-6 * np.abs(want).max()


@pytest.mark.parametrize("rccl", [True, False], ids=["rccl", "no_rccl"])
def test_multichannel_rccl_c_host(tmp_path, hip_library, rccl):
    """examples/multichannel_rccl.c: the multi-channel config from a C host -- one batched plan per GPU,
    RCCL only as a barrier (ncclCommInitAll + 1-element all-reduce), or -DSDFT_NO_RCCL: a host-side
    start line and no librccl at all.  Runs on however many GPUs the box has (1 here); the synthesis
    checksum of channel 0 is compared with the oracle."""
    import re
    from sdft_amd.signals import sine_sweep
    libdir = os.path.dirname(hip_library)
    rt = hip_runtime_dir()
    rocm = os.path.dirname(rt)
    if rccl and not os.path.exists(os.path.join(rt, "librccl.so")):
        pytest.skip("RCCL not present")
    exe = tmp_path / "multichannel_rccl"
    cmd = ["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
           os.path.join(ROOT, "examples", "multichannel_rccl.c"), "-o", str(exe), "-L", libdir, "-lsdft_hip", "-L", rt,
           "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rt}"] + (["-lrccl"] if rccl else ["-DSDFT_NO_RCCL"])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    per_gpu, n, m, steps = 3, 2000, 128, 2
    r = subprocess.run([str(exe), str(per_gpu), str(n), str(m), str(steps), "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    mt = re.search(r"gpus=(\d+) channels=(\d+) .* ([\d.]+) Msamples/s aggregate\s+checksum=([-+.\de]+)", r.stdout)
    assert mt, r.stdout
    gpus, channels, rate, checksum = int(mt.group(1)), int(mt.group(2)), float(mt.group(3)), float(mt.group(4))
    assert gpus == 1 and channels == per_gpu and rate > 0
    x = sine_sweep(n, channel=0, channels=channels)
    ref = O.best(m, "hann", 1.0, "f32f64")
    for _ in range(steps + 1):                       # warm-up + timed steps stream the same block again
        d = ref.sdft(x)
    y = ref.isdft(d).astype(np.float64)
    want = float((y * ((np.arange(n) % 7) + 1)).sum())
    assert abs(checksum - want) <= 1e-6 * max(1.0, abs(want)), (checksum, want)
