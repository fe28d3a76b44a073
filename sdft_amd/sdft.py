"""Host-side mirror of the reference's operator interface over the C-ABI.

``SDFT(dftsize, window, latency).sdft(x) / .isdft(dfts)`` has the shape of the reference's
Python class (/root/reference/python/src/sdft/sdft.py:30,76,122) and the call order of its C
test driver (/root/reference/test/test.c:49-93), but every call goes through
``libsdft_hip.so``: torch tensors on the GPU are passed as device pointers (nothing is copied),
numpy arrays take the library's staged host-pointer path.  torch is used for device memory
only.  There is no CPU implementation in this package.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from .capi import Api, OPS, STAGES, WINDOWS, SdftHipError

_NP_REAL = {"f32": np.float32, "f64": np.float64}
_NP_CPLX = {"f32": np.complex64, "f64": np.complex128}


def _torch():
    import torch
    return torch


def _is_tensor(a) -> bool:
    return type(a).__module__.startswith("torch")


class SDFT:
    """One analysis/synthesis plan = one stream (or a batch of independent channels).

    Parameters follow ``sdft_alloc_custom`` (reference sdft.h:413): ``dftsize`` bins, analysis
    ``window`` in {boxcar, hann, hamming, blackman}, synthesis ``latency`` in (0, 1].
    ``combo`` selects time/frequency domain scalar types (``SDFT_TD_*`` / ``SDFT_FD_*`` macros).
    ``channels`` > 1 allocates a batched plan (an addition over the reference).
    """

    def __init__(self, dftsize: int, window="hann", latency: float = 1.0, combo: str = "f32f64",
                 channels: int = 1, device=None, hooks: bool = False):
        # hooks: the plan lives in libsdft_hip_hooks.so (the same sources built with -DSDFT_HIP_TEST_HOOKS), whose set_option also knows
        # the keys that force every remaining fork of the host logic -- for the tests and the probes; see set_option below
        self.api = Api(combo, hooks=hooks)
        self._options = []                                   # (key, value) set so far, and the caller's stream: replayed when the plan moves
        self._stream = None
        self.combo = combo
        self.td = _NP_REAL[combo[:3]]
        self.fd = _NP_REAL[combo[3:]]
        self.fdx = _NP_CPLX[combo[3:]]
        self.dftsize = int(dftsize)
        self.channels = int(channels)
        self.window = WINDOWS[window] if isinstance(window, str) else int(window)
        self.latency = float(latency)
        if device is not None:
            if self.api.lib.sdft_hip_set_device(int(device)) != 0:
                self.api.check()
        self._p = self.api.alloc_batch(self.dftsize, self.window, self.latency, self.channels)
        self.device = int(self.api.get_option(self._p, b"device")) if self._p else -1
        if not self._p:
            err = self.api.last_error()
            self.api.lib.sdft_hip_clear_error()
            raise SdftHipError(f"sdft_alloc failed: {err}")

    # ---- lifetime -----------------------------------------------------------------------
    def close(self):
        if getattr(self, "_p", None):
            self.api.free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        self.api.clear()
        self.api.reset(self._p)
        self.api.check()

    def size(self) -> int:
        return int(self.api.size(self._p))

    # ---- options ------------------------------------------------------------------------
    def set_option(self, key: str, value: int):
        """sdft_hip_set_option.  A key the product library does not know may be a test hook (include/sdft/sdft_hip.h lists them): the plan
        then moves to libsdft_hip_hooks.so -- a new plan with the same parameters on the same device, the options set so far replayed
        and the stream state copied over (sdft_hip_get_state / set_state) -- so that the tests can force a route on any plan they make; a key
        neither build knows raises."""
        if self.api.set_option(self._p, key.encode(), int(value)) == 0:
            self._options.append((key, int(value)))
            return
        if not self.api.hooks:
            other = Api(self.combo, hooks=True)
            if self.device >= 0:
                other.lib.sdft_hip_set_device(self.device)
            q = other.alloc_batch(self.dftsize, self.window, self.latency, self.channels)
            if q and other.set_option(q, key.encode(), int(value)) == 0:
                for k, v in self._options:
                    other.set_option(q, k.encode(), v)
                acc, fid, hist, cursor = self.state()            # (synchronises; the plan may have been called already)
                self.api.free(self._p)
                self.api, self._p = other, q
                # (a plan nobody has called yet is not handed a state: an installed fid counts as the host's own, and the relay
                # form of the exact carries, which needs the canonical rotation, would never be taken)
                if cursor != 0 or np.any(acc) or np.any(hist):
                    self.set_state(acc, fid, hist, cursor)
                if self._stream is not None:
                    self.set_stream(self._stream)
                self._options.append((key, int(value)))
                return
            if q:
                other.free(q)
        raise SdftHipError(f"unknown option {key!r}")

    def get_option(self, key: str) -> int:
        return int(self.api.get_option(self._p, key.encode()))

    def set_stream(self, stream_handle: int):
        """Launch on a caller-owned HIP stream (e.g. ``torch.cuda.Stream().cuda_stream``)."""
        if self.api.set_stream(self._p, C.c_void_p(stream_handle)) != 0:
            self.api.check()
        self._stream = stream_handle

    def synchronize(self):
        if self.api.synchronize(self._p) != 0:
            self.api.check()

    def profile(self) -> dict:
        """-> {stage: (milliseconds, launches)} accumulated since the last call; needs option profile=1."""
        ms = (C.c_double * 4)()
        calls = (C.c_long * 4)()
        if self.api.get_profile(self._p, ms, calls) != 0:
            self.api.check()
        return {s: (ms[i], calls[i]) for i, s in enumerate(STAGES)}

    def state(self):
        """(acc, fid, hist, cursor) copied from the device; hist is in time order."""
        n, c = self.dftsize, self.channels
        acc = np.empty((c, n), dtype=self.fdx)
        fid = np.empty((c, n), dtype=self.fdx)
        hist = np.empty((c, 2 * n), dtype=self.td)
        cur = C.c_size_t(0)
        if self.api.get_state(self._p, acc.ctypes.data, fid.ctypes.data, hist.ctypes.data, C.byref(cur)) != 0:
            self.api.check()
        if c == 1:
            acc, fid, hist = acc[0], fid[0], hist[0]
        return acc, fid, hist, int(cur.value)

    def set_state(self, acc, fid, hist, cursor: int):
        """Checkpoint / resume: install a state obtained from :meth:`state` (of a plan with the same
        parameters, possibly on another GPU)."""
        acc = np.ascontiguousarray(acc, dtype=self.fdx); fid = np.ascontiguousarray(fid, dtype=self.fdx)
        hist = np.ascontiguousarray(hist, dtype=self.td)
        assert acc.size == self.channels * self.dftsize and hist.size == 2 * self.channels * self.dftsize
        if self.api.set_state(self._p, acc.ctypes.data, fid.ctypes.data, hist.ctypes.data, int(cursor)) != 0:
            self.api.check()
            raise SdftHipError("sdft_hip_set_state failed")

    # ---- analysis / synthesis ---------------------------------------------------------------
    def _check_tensor(self, t, what, dtype, shape=None):
        """Device tensors are handed to the library as raw pointers: they must live on the plan's GPU,
        be dense, and have exactly the element type and shape the C-ABI expects."""
        torch = _torch()
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError(f"{what} must be a CUDA tensor on the plan's device cuda:{self.device}, got {t.device}")
        if not t.is_contiguous():
            raise ValueError(f"{what} must be contiguous")
        if t.dtype != getattr(torch, np.dtype(dtype).name):
            raise ValueError(f"{what} must have dtype {np.dtype(dtype).name}, got {t.dtype}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"{what} must have shape {tuple(shape)}, got {tuple(t.shape)}")

    def _shape_x(self, shape):
        if self.channels == 1 and len(shape) == 1:
            return shape[0]
        if len(shape) == 2 and shape[0] == self.channels:
            return shape[1]
        raise ValueError(f"samples must have shape (n,) or ({self.channels}, n), got {tuple(shape)}")

    def sdft(self, x, out=None):
        """Analyse samples ``x`` -> DFT matrix of shape (n, dftsize) [(channels, n, dftsize) if batched].

        State persists across calls exactly like the reference's plan (endless streaming).
        """
        self.api.clear()
        if _is_tensor(x):
            torch = _torch()
            n = self._shape_x(x.shape)
            self._check_tensor(x, "samples", self.td)
            shape = (n, self.dftsize) if x.dim() == 1 else (self.channels, n, self.dftsize)
            if out is None:
                out = torch.empty(shape, dtype=getattr(torch, np.dtype(self.fdx).name), device=x.device)
            self._check_tensor(out, "out", self.fdx, shape)
            self.api.sdft_n(self._p, n, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()))
        else:
            x = np.ascontiguousarray(x, dtype=self.td)
            n = self._shape_x(x.shape)
            shape = (n, self.dftsize) if x.ndim == 1 else (self.channels, n, self.dftsize)
            if out is None:
                out = np.empty(shape, dtype=self.fdx)
            assert out.flags.c_contiguous and out.shape == shape and out.dtype == self.fdx
            self.api.sdft_n(self._p, n, C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data))
        self.api.check()
        return out

    def isdft(self, dfts, out=None):
        """Synthesise samples from a DFT matrix (n, dftsize) [(channels, n, dftsize)]."""
        self.api.clear()
        batched = (len(dfts.shape) == 3)
        n = dfts.shape[-2]
        assert dfts.shape[-1] == self.dftsize and (not batched or dfts.shape[0] == self.channels)
        yshape = (self.channels, n) if batched else (n,)
        if _is_tensor(dfts):
            torch = _torch()
            self._check_tensor(dfts, "dfts", self.fdx)
            if out is None:
                out = torch.empty(yshape, dtype=getattr(torch, np.dtype(self.td).name), device=dfts.device)
            self._check_tensor(out, "out", self.td, yshape)
            self.api.isdft_n(self._p, n, C.c_void_p(dfts.data_ptr()), C.c_void_p(out.data_ptr()))
        else:
            dfts = np.ascontiguousarray(dfts, dtype=self.fdx)
            if out is None:
                out = np.empty(yshape, dtype=self.td)
            assert out.flags.c_contiguous and out.shape == yshape and out.dtype == self.td
            self.api.isdft_n(self._p, n, C.c_void_p(dfts.ctypes.data), C.c_void_p(out.ctypes.data))
        self.api.check()
        return out


    def process(self, x, op="identity", gain=None, shift=0, out=None, dfts=None, hop=0, threshold=0.0, floor=0.0,
                exponent=1.0, scale=1.0, expr=None, expr_params=()):
        """Fused analysis -> spectral operation -> synthesis (``sdft_hip_process_n``): returns the
        processed samples; the DFT matrix is not materialised unless ``dfts`` (a CUDA tensor of shape
        (n, dftsize) [(channels, n, dftsize)]) asks for a copy of the processed spectrum.

        ``op``: "identity", "gain" (``gain`` = real array of dftsize factors), "cgain" (``gain`` = complex array),
        "shift" (``shift`` bins), "gain_rows" / "cgain_rows" (``gain`` = (rows, dftsize) array, row r for the call's samples
        [r*hop, (r+1)*hop), the last row for the rest), "gate" (``threshold``, ``floor``), "power" (``exponent``, ``scale``) or
        "expr" (``expr`` = HIP C++ statements on ``re``, ``im`` of bin ``k`` at sample ``t`` of channel ``ch`` with the
        parameters ``p[i]`` = ``expr_params``; compiled into the kernel at run time, see sdft_hip.h).
        """
        self.api.clear()
        kind = OPS[op] if isinstance(op, str) else int(op)
        params = None
        keep = None
        if kind in (OPS["gain_rows"], OPS["cgain_rows"]):
            gdt = self.fd if kind == OPS["gain_rows"] else self.fdx
            if _is_tensor(gain):
                assert gain.dim() == 2 and gain.shape[1] == self.dftsize
                self._check_tensor(gain, "gain", gdt)
                gptr, rows = gain.data_ptr(), int(gain.shape[0])
                keep_rows = gain
            else:
                keep_rows = np.ascontiguousarray(gain, dtype=gdt)
                assert keep_rows.ndim == 2 and keep_rows.shape[1] == self.dftsize
                gptr, rows = keep_rows.ctypes.data, int(keep_rows.shape[0])

            class _Table(C.Structure):
                _fields_ = [("gains", C.c_void_p), ("rows", C.c_size_t), ("hop", C.c_size_t)]
            keep = (_Table(gptr, rows, int(hop)), keep_rows)
            params = C.cast(C.byref(keep[0]), C.c_void_p)
        elif kind == OPS["expr"]:
            pv = np.ascontiguousarray(expr_params, dtype=self.fd).reshape(-1)
            text = C.c_char_p(str(expr).encode())

            class _Expr(C.Structure):
                _fields_ = [("expr", C.c_char_p), ("params", C.c_void_p), ("nparams", C.c_size_t)]
            keep = (_Expr(text, pv.ctypes.data if pv.size else None, int(pv.size)), pv, text)
            params = C.cast(C.byref(keep[0]), C.c_void_p)
        elif kind in (OPS["gate"], OPS["power"]):
            vals = (threshold, floor) if kind == OPS["gate"] else (exponent, scale)
            keep = np.asarray(vals, dtype=self.fd)
            params = C.c_void_p(keep.ctypes.data)
        elif kind in (OPS["gain"], OPS["cgain"]):
            gdt = self.fd if kind == OPS["gain"] else self.fdx
            if _is_tensor(gain):
                self._check_tensor(gain, "gain", gdt, (self.dftsize,))
                params = C.c_void_p(gain.data_ptr())
            else:
                keep = np.ascontiguousarray(gain, dtype=gdt)
                assert keep.shape == (self.dftsize,)
                params = C.c_void_p(keep.ctypes.data)
        elif kind == OPS["shift"]:
            keep = C.c_long(int(shift))
            params = C.cast(C.byref(keep), C.c_void_p)
        dptr = None
        if _is_tensor(x):
            torch = _torch()
            n = self._shape_x(x.shape)
            self._check_tensor(x, "samples", self.td)
            if out is None:
                out = torch.empty_like(x)
            self._check_tensor(out, "out", self.td, tuple(x.shape))
            if dfts is not None:
                self._check_tensor(dfts, "dfts", self.fdx, (n, self.dftsize) if x.dim() == 1 else (self.channels, n, self.dftsize))
                dptr = C.c_void_p(dfts.data_ptr())
            rc = self.api.process_n(self._p, n, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), kind, params, dptr)
        else:
            x = np.ascontiguousarray(x, dtype=self.td)
            n = self._shape_x(x.shape)
            if out is None:
                out = np.empty_like(x)
            assert out.flags.c_contiguous and out.shape == x.shape and out.dtype == self.td
            if dfts is not None:
                self._check_tensor(dfts, "dfts", self.fdx)
                dptr = C.c_void_p(dfts.data_ptr())
            rc = self.api.process_n(self._p, n, C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), kind, params, dptr)
        if rc != 0:
            self.api.check()
            raise SdftHipError("sdft_hip_process_n failed")
        self.api.check()
        return out


def plan_tables(dftsize: int, latency: float = 1.0, combo: str = "f32f64"):
    """Host-only: (tw, syn, wtab, weights) exactly as the library uploads them (no GPU needed)."""
    api = Api(combo)
    fd, fdx = _NP_REAL[combo[3:]], _NP_CPLX[combo[3:]]
    tw = np.empty(dftsize, dtype=fdx)
    syn = np.empty(dftsize, dtype=fdx)
    wtab = np.empty(2 * dftsize, dtype=fdx)
    w = np.empty(2, dtype=fd)
    api.plan_tables(dftsize, latency, tw.ctypes.data, syn.ctypes.data, wtab.ctypes.data, w.ctypes.data)
    return tw, syn, wtab, w


def check_expr(expr: str, arch: str | None = None) -> None:
    """Compiles the statements of an ``op="expr"`` operation without running them (``sdft_hip_check_expr``; needs no GPU).
    Raises :class:`SdftHipError` with the compiler's words when they do not compile."""
    from . import capi
    lib = capi.load()
    lib.sdft_hip_clear_error()
    if lib.sdft_hip_check_expr(str(expr).encode(), arch.encode() if arch else None) != 0:
        e = lib.sdft_hip_last_error()
        lib.sdft_hip_clear_error()
        raise SdftHipError(e.decode() if e else "the expression does not compile")
