"""ctypes front-end of the parity oracle.  TEST INFRASTRUCTURE ONLY.

Two interchangeable back-ends with one interface:

* :class:`Port`      -- ``oracle/_build/liboracle_<combo>.so``, our CPU restatement
                        (``oracle/sdft_oracle.c``); always available after ``make -C oracle``.
* :class:`Reference` -- ``oracle/_ref/libsdft_ref_<combo>.so``, the genuine reference header
                        ``c/src/sdft/sdft.h`` compiled from ``/root/reference`` by
                        ``oracle/Makefile``.  The built library travels to the GPU box; the
                        sources do not.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``sdft_amd/`` does.

``combo`` is ``"<td><fd>"`` with td/fd in {f32, f64}: the time-domain sample type and the
frequency-domain scalar type (reference macros ``SDFT_TD_*`` / ``SDFT_FD_*``, sdft.h:21-37).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
COMBOS = ("f32f64", "f32f32", "f64f64", "f64f32")
WINDOWS = {"boxcar": 0, "hann": 1, "hamming": 2, "blackman": 3}   # sdft.h:127-133

_REAL = {"f32": np.float32, "f64": np.float64}
_CPLX = {"f32": np.complex64, "f64": np.complex128}


def combo_types(combo: str):
    """-> (td real dtype, fd real dtype, fd complex dtype)"""
    td, fd = combo[:3], combo[3:]
    return _REAL[td], _REAL[fd], _CPLX[fd]


def build(ref: bool = True) -> None:
    """Compile the oracle libraries (and the reference build when its sources are present)."""
    subprocess.run(["make", "-C", HERE, "port"], check=True, capture_output=True)
    if ref and os.path.exists(os.environ.get("SDFT_REF_DIR", "/root/reference") + "/c/src/sdft/sdft.h"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


def port_path(combo: str) -> str:
    return os.path.join(HERE, "_build", f"liboracle_{combo}.so")


def ref_path(combo: str) -> str:
    return os.path.join(HERE, "_ref", f"libsdft_ref_{combo}.so")


def build_native(combo: str = "f32f64"):
    """-O3 -march=native builds for bench.py's second CPU figure, made on the machine that runs it (the code does not
    travel).  -> (path, kind) of the strongest one: the reference header when SDFT_REF_DIR points at a checkout, else the port."""
    assert combo == "f32f64"
    ref_dir = os.environ.get("SDFT_REF_DIR")
    if ref_dir and os.path.exists(ref_dir + "/c/src/sdft/sdft.h"):
        subprocess.run(["make", "-C", HERE, "-B", "native-ref"], check=True, capture_output=True)
        return os.path.join(HERE, "_ref", f"libsdft_ref_native_{combo}.so"), "reference"
    subprocess.run(["make", "-C", HERE, "-B", "native"], check=True, capture_output=True)
    return os.path.join(HERE, "_build", f"liboracle_native_{combo}.so"), "port"


def have_port(combo: str = "f32f64") -> bool:
    return os.path.exists(port_path(combo))


def have_reference(combo: str = "f32f64") -> bool:
    return os.path.exists(ref_path(combo))


def _win(window) -> int:
    return WINDOWS[window] if isinstance(window, str) else int(window)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class _Base:
    combo: str

    def _setup(self, combo, dftsize, window, latency):
        self.combo = combo
        self.td, self.fd, self.fdx = combo_types(combo)
        self.dftsize = int(dftsize)
        self.window = _win(window)
        self.latency = float(latency)

    def _as_td(self, x):
        return np.ascontiguousarray(x, dtype=self.td)

    def _as_fdx(self, d):
        d = np.ascontiguousarray(d, dtype=self.fdx)
        assert d.ndim == 2 and d.shape[1] == self.dftsize
        return d


class Port(_Base):
    """Our restatement (oracle/sdft_oracle.c)."""

    kind = "port"

    def __init__(self, dftsize, window="hann", latency=1.0, combo="f32f64", lib_path=None):
        self._setup(combo, dftsize, window, latency)
        lib = C.CDLL(lib_path or port_path(combo))
        lib.oracle_new.restype = C.c_void_p
        lib.oracle_new.argtypes = [C.c_size_t, C.c_int, C.c_double]
        for name in ("oracle_free", "oracle_reset"):
            getattr(lib, name).restype = None
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.oracle_sdft_n.restype = None
        lib.oracle_sdft_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.oracle_isdft_n.restype = None
        lib.oracle_isdft_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.oracle_digest_n.restype = None
        lib.oracle_digest_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_tables.restype = None
        lib.oracle_tables.argtypes = [C.c_void_p] * 4
        lib.oracle_state.restype = None
        lib.oracle_state.argtypes = [C.c_void_p] * 4
        lib.oracle_phase.restype = C.c_size_t
        lib.oracle_phase.argtypes = [C.c_void_p]
        lib.oracle_sizeof_td.restype = C.c_size_t
        lib.oracle_sizeof_fd.restype = C.c_size_t
        assert lib.oracle_sizeof_td() == np.dtype(self.td).itemsize
        assert lib.oracle_sizeof_fd() == np.dtype(self.fd).itemsize
        self._lib = lib
        self._p = lib.oracle_new(self.dftsize, self.window, self.latency)

    def __del__(self):
        if getattr(self, "_p", None):
            self._lib.oracle_free(self._p)
            self._p = None

    def reset(self):
        self._lib.oracle_reset(self._p)

    def sdft(self, x, out=None):
        x = self._as_td(x)
        if out is None:
            out = np.empty((x.size, self.dftsize), dtype=self.fdx)
        self._lib.oracle_sdft_n(self._p, x.size, _ptr(x), _ptr(out))
        return out

    def isdft(self, dfts, out=None):
        dfts = self._as_fdx(dfts)
        if out is None:
            out = np.empty(dfts.shape[0], dtype=self.td)
        self._lib.oracle_isdft_n(self._p, dfts.shape[0], _ptr(dfts), _ptr(out))
        return out

    def digest(self, x, with_y=True):
        """Streaming per-row digests (n, 4) and y, without materialising the matrix."""
        x = self._as_td(x)
        dig = np.empty((x.size, 4), dtype=np.float64)
        y = np.empty(x.size, dtype=self.td) if with_y else None
        self._lib.oracle_digest_n(self._p, x.size, _ptr(x), _ptr(dig), _ptr(y) if with_y else None)
        return dig, y

    def tables(self):
        tw = np.empty(self.dftsize, dtype=self.fdx)
        syn = np.empty(self.dftsize, dtype=self.fdx)
        w = np.empty(2, dtype=self.fd)
        self._lib.oracle_tables(self._p, _ptr(tw), _ptr(syn), _ptr(w))
        return tw, syn, w

    def state(self):
        """-> (acc, fid, history in time order, cursor)"""
        acc = np.empty(self.dftsize, dtype=self.fdx)
        fid = np.empty(self.dftsize, dtype=self.fdx)
        hist = np.empty(self.dftsize * 2, dtype=self.td)
        self._lib.oracle_state(self._p, _ptr(acc), _ptr(fid), _ptr(hist))
        return acc, fid, hist, int(self._lib.oracle_phase(self._p))


class Reference(_Base):
    """The genuine reference header, compiled by oracle/Makefile into oracle/_ref/."""

    kind = "reference"

    def __init__(self, dftsize, window="hann", latency=1.0, combo="f32f64", lib_path=None):
        self._setup(combo, dftsize, window, latency)
        lib = C.CDLL(lib_path or ref_path(combo))
        lib.sdft_alloc_custom.restype = C.c_void_p
        lib.sdft_alloc_custom.argtypes = [C.c_size_t, C.c_int, C.c_double]
        for name in ("sdft_free", "sdft_reset"):
            getattr(lib, name).restype = None
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.sdft_sdft_n.restype = None
        lib.sdft_sdft_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.sdft_isdft_n.restype = None
        lib.sdft_isdft_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        for name in ("ref_analysis_twiddles", "ref_synthesis_twiddles", "ref_accoutput", "ref_fiddles", "ref_input"):
            getattr(lib, name).restype = C.c_void_p
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.ref_cursor.restype = C.c_size_t
        lib.ref_cursor.argtypes = [C.c_void_p]
        for name in ("ref_analysis_weight", "ref_synthesis_weight"):
            getattr(lib, name).restype = C.c_double
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.ref_sizeof_td.restype = C.c_size_t
        lib.ref_sizeof_fd.restype = C.c_size_t
        assert lib.ref_sizeof_td() == np.dtype(self.td).itemsize
        assert lib.ref_sizeof_fd() == np.dtype(self.fd).itemsize
        self._lib = lib
        self._p = lib.sdft_alloc_custom(self.dftsize, self.window, self.latency)

    def __del__(self):
        if getattr(self, "_p", None):
            self._lib.sdft_free(self._p)
            self._p = None

    def reset(self):
        self._lib.sdft_reset(self._p)

    def sdft(self, x, out=None):
        x = self._as_td(x)
        if out is None:
            out = np.empty((x.size, self.dftsize), dtype=self.fdx)
        self._lib.sdft_sdft_n(self._p, x.size, _ptr(x), _ptr(out))
        return out

    def isdft(self, dfts, out=None):
        dfts = self._as_fdx(dfts)
        if out is None:
            out = np.empty(dfts.shape[0], dtype=self.td)
        self._lib.sdft_isdft_n(self._p, dfts.shape[0], _ptr(dfts), _ptr(out))
        return out

    def _view(self, addr, dtype, count):
        buf = (C.c_char * (np.dtype(dtype).itemsize * count)).from_address(addr)
        return np.frombuffer(buf, dtype=dtype, count=count).copy()

    def tables(self):
        n = self.dftsize
        tw = self._view(self._lib.ref_analysis_twiddles(self._p), self.fdx, n)
        syn = self._view(self._lib.ref_synthesis_twiddles(self._p), self.fdx, n)
        w = np.array([self._lib.ref_analysis_weight(self._p), self._lib.ref_synthesis_weight(self._p)], dtype=self.fd)
        return tw, syn, w

    def state(self):
        n = self.dftsize
        acc = self._view(self._lib.ref_accoutput(self._p), self.fdx, n)
        fid = self._view(self._lib.ref_fiddles(self._p), self.fdx, n)
        ring = self._view(self._lib.ref_input(self._p), self.td, 2 * n)
        cur = int(self._lib.ref_cursor(self._p))
        hist = np.roll(ring, -cur)        # ring[cursor] is the oldest sample -> time order
        return acc, fid, hist, cur


def best(dftsize, window="hann", latency=1.0, combo="f32f64"):
    """The strongest oracle available here: the reference build if present, else the port."""
    if have_reference(combo):
        return Reference(dftsize, window, latency, combo)
    return Port(dftsize, window, latency, combo)
