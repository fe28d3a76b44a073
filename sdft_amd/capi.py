"""ctypes binding of the C-ABI in ``libsdft_hip.so`` (``include/sdft/sdft.h`` + ``sdft_hip.h``).

This is the same boundary a C host links against; nothing here computes anything.  There is no
CPU fallback: if the library is missing or no GPU is present, calls fail loudly.
"""

from __future__ import annotations

import ctypes as C
import os

from . import build as _build

COMBOS = _build.COMBOS
WINDOWS = {"boxcar": 0, "hann": 1, "hamming": 2, "blackman": 3}   # sdft.h:127-133 of the reference
STAGES = ("delta", "carry", "forward", "inverse")
OPS = {"identity": 0, "gain": 1, "shift": 2, "cgain": 3, "gain_rows": 4, "cgain_rows": 5, "gate": 6, "power": 7, "expr": 8}   # enum sdft_hip_op (sdft_hip.h)

# every typed entry point exported per (td, fd) combination:  name -> (restype, argtypes)
_TD = {"f32": C.c_float, "f64": C.c_double}
_FD = {"f32": C.c_float, "f64": C.c_double}


def typed_signatures(combo: str):
    td, fd = _TD[combo[:3]], _FD[combo[3:]]
    vp, sz = C.c_void_p, C.c_size_t
    return {
        # drop-in surface (reference sdft.h:413-687)
        "alloc": (vp, [sz]),
        "alloc_custom": (vp, [sz, C.c_int, C.c_double]),
        "free": (None, [vp]),
        "reset": (None, [vp]),
        "size": (sz, [vp]),
        "window": (C.c_int, [vp]),
        "latency": (C.c_double, [vp]),
        "sdft": (None, [vp, td, vp]),
        "sdft_n": (None, [vp, sz, vp, vp]),
        "sdft_nd": (None, [vp, sz, vp, vp]),
        "isdft": (td, [vp, vp]),
        "isdft_n": (None, [vp, sz, vp, vp]),
        "isdft_nd": (None, [vp, sz, vp, vp]),
        # additions
        "alloc_batch": (vp, [sz, C.c_int, C.c_double, sz]),
        "channels": (sz, [vp]),
        "set_stream": (C.c_int, [vp, vp]),
        "get_stream": (vp, [vp]),
        "synchronize": (C.c_int, [vp]),
        "time_hops": (C.c_double, [vp, sz, sz, vp, vp, vp]),
        "set_option": (C.c_int, [vp, C.c_char_p, C.c_long]),
        "get_option": (C.c_long, [vp, C.c_char_p]),
        "get_profile": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
        "get_state": (C.c_int, [vp, vp, vp, vp, C.POINTER(sz)]),
        "set_state": (C.c_int, [vp, vp, vp, vp, sz]),
        "plan_tables": (C.c_int, [sz, C.c_double, vp, vp, vp, vp]),
        "process_n": (C.c_int, [vp, sz, vp, vp, C.c_int, vp, vp]),
    }


UNTYPED = {
    "sdft_hip_last_error": (C.c_char_p, []),
    "sdft_hip_clear_error": (None, []),
    "sdft_hip_last_warning": (C.c_char_p, []),
    "sdft_hip_clear_warning": (None, []),
    "sdft_hip_device_count": (C.c_int, []),
    "sdft_hip_set_device": (C.c_int, [C.c_int]),
    "sdft_hip_get_device": (C.c_int, []),
    "sdft_hip_version": (C.c_char_p, []),
    "sdft_hip_selftest": (C.c_int, []),
    "sdft_hip_check_expr": (C.c_int, [C.c_char_p, C.c_char_p]),
    "sdft_hip_store_ceiling": (C.c_double, [C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_int]),
    "sdft_hip_load_ceiling": (C.c_double, [C.c_void_p, C.c_size_t, C.c_int]),
    "sdft_hip_load_rows_ceiling": (C.c_double, [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_int]),
    "sdft_hip_hold_cus": (C.c_int, [C.c_uint, C.c_double]),
    "sdft_hip_malloc_matrix": (C.c_void_p, [C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "sdft_hip_malloc_matrix_in_arena": (C.c_void_p, [C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]),
    "sdft_hip_free_matrix": (C.c_int, [C.c_void_p]),
    "sdft_hip_matrix_placement": (C.c_int, [C.c_void_p, C.c_void_p]),
}


class Placement(C.Structure):
    """sdft_hip_placement_t (include/sdft/sdft_hip.h)."""
    _fields_ = [("arena_bytes", C.c_size_t), ("window_offset", C.c_size_t), ("boundary_offset", C.c_size_t),
                ("pair_probes", C.c_int), ("window_probes", C.c_int),
                ("window_gbs", C.c_double), ("start_gbs", C.c_double), ("probe_ms", C.c_double), ("arenas_tried", C.c_int)]

    def as_dict(self) -> dict:
        return {"arena_bytes": int(self.arena_bytes), "window_offset": int(self.window_offset), "boundary_offset": int(self.boundary_offset),
                "pair_probes": int(self.pair_probes), "window_probes": int(self.window_probes),
                "window_gbs": round(float(self.window_gbs), 1), "start_gbs": round(float(self.start_gbs), 1), "probe_ms": round(float(self.probe_ms), 2),
                "arenas_tried": int(self.arenas_tried)}


class PlacedMatrix:
    """A DFT matrix in device memory placed by the library (sdft_hip_malloc_matrix_in_arena), seen as a torch tensor (a view: no copy;
    the allocation lives until free()).  Test and bench plumbing: a C host uses the two calls directly."""

    def __init__(self, shape, torch_dtype, arena_extra: int = 64 << 30):
        import math
        import torch
        lib = load()
        self.lib = lib
        self.nbytes = math.prod(shape) * torch.empty(0, dtype=torch_dtype).element_size()
        gbs = C.c_double(0.0)
        self.ptr = lib.sdft_hip_malloc_matrix_in_arena(self.nbytes, self.nbytes + arena_extra, C.byref(gbs))
        if not self.ptr:
            err = lib.sdft_hip_last_error()
            lib.sdft_hip_clear_error()
            raise SdftHipError((err or b"sdft_hip_malloc_matrix_in_arena failed").decode())
        info = Placement()
        lib.sdft_hip_matrix_placement(C.c_void_p(self.ptr), C.byref(info))
        self.info = info.as_dict()
        typestr = {torch.complex128: "<c16", torch.complex64: "<c8", torch.float64: "<f8", torch.float32: "<f4", torch.uint8: "|u1"}[torch_dtype]
        holder = type("DevicePointer", (), {"__cuda_array_interface__": {"shape": tuple(shape), "typestr": typestr, "data": (int(self.ptr), False),
                                                                          "version": 2, "strides": None}})()
        self.tensor = torch.as_tensor(holder, device=torch.device("cuda", torch.cuda.current_device()))
        assert self.tensor.data_ptr() == self.ptr

    def free(self):
        if getattr(self, "ptr", None):
            self.tensor = None
            self.lib.sdft_hip_free_matrix(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def symbol(name: str, combo: str) -> str:
    return f"sdft_hip_{name}_{combo}"


class SdftHipError(RuntimeError):
    pass


_lib = None


def library_path() -> str:
    return os.environ.get("SDFT_HIP_LIBRARY", _build.LIB)


def _bind_hip_runtime() -> str:
    """Make exactly one HIP runtime visible (RTLD_GLOBAL) before libsdft_hip.so is loaded.

    libsdft_hip.so carries no DT_NEEDED for libamdhip64 (see build.py).  If PyTorch is in the
    process its bundled runtime must be the one (a second runtime cannot open the GPU), otherwise
    the system ROCm runtime is used.
    """
    import sys
    cands = []
    if "torch" not in sys.modules and not os.environ.get("SDFT_HIP_NO_TORCH"):
        try:                        # PyTorch may be imported later by the same process: settle on its
            import torch  # noqa: F401   runtime now, or its CUDA init would find the GPU already taken
        except Exception:
            pass
    if "torch" in sys.modules:
        tl = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib")
        cands.append(os.path.join(tl, "libamdhip64.so"))
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    for c in cands:
        if os.path.sep in c and not os.path.exists(c):
            continue
        try:
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            return c
        except OSError:
            continue
    raise SdftHipError("no HIP runtime (libamdhip64.so) could be loaded")


hip_runtime = None


_lib_hooks = None


def load(build_if_missing: bool = True, hooks: bool = False) -> C.CDLL:
    """Load libsdft_hip.so (hooks: libsdft_hip_hooks.so, the same sources built with the path-forcing test options) and declare every
    prototype; raises if it cannot be had.  Both may live in one process: each has its own plans, error channel and kernels."""
    global _lib, _lib_hooks, hip_runtime
    if hooks and _lib_hooks is not None:
        return _lib_hooks
    if not hooks and _lib is not None:
        return _lib
    path = (_build.LIB_HOOKS if "SDFT_HIP_LIBRARY" not in os.environ else os.environ.get("SDFT_HIP_HOOKS_LIBRARY", _build.LIB_HOOKS)) if hooks else library_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise SdftHipError(f"{path} not found; run `python -m sdft_amd.build`")
        path = _build.build(hooks=hooks)
    hip_runtime = _bind_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in UNTYPED.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for combo in COMBOS:
        for name, (res, args) in typed_signatures(combo).items():
            fn = getattr(lib, symbol(name, combo))
            fn.restype, fn.argtypes = res, args
    if hooks:
        _lib_hooks = lib
    else:
        _lib = lib
    return lib


class Api:
    """The typed entry points of one (td, fd) combination as attributes: ``api.sdft_n(...)``."""

    def __init__(self, combo: str = "f32f64", hooks: bool = False):
        if combo not in COMBOS:
            raise ValueError(f"unknown type combination {combo!r}; expected one of {COMBOS}")
        self.combo = combo
        self.hooks = hooks
        self.lib = load(hooks=hooks)
        for name in typed_signatures(combo):
            setattr(self, name, getattr(self.lib, symbol(name, combo)))

    def last_error(self):
        e = self.lib.sdft_hip_last_error()
        return e.decode() if e else None

    def last_warning(self):
        """What a call that succeeded had to tell (a recovered time-out of the exact-carry kernels), or None; cleared by reading."""
        w = self.lib.sdft_hip_last_warning()
        if not w:
            return None
        text = w.decode()
        self.lib.sdft_hip_clear_warning()
        return text

    def clear(self):
        """Forget an error recorded by an earlier call on this thread (the reference's signatures return void: the error
        channel is the only way a failure shows, so every wrapped call starts from a clean slate)."""
        self.lib.sdft_hip_clear_error()

    def check(self):
        """Raise if the library recorded an error on this thread (the C API itself never aborts)."""
        e = self.last_error()
        if e:
            self.lib.sdft_hip_clear_error()
            raise SdftHipError(e)
