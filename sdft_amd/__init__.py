"""sdft_amd -- MI355X-native Sliding DFT engine behind the jurihock/sdft C API.

The product is ``sdft_amd/lib/libsdft_hip.so`` (C-ABI + hand-written HIP kernels for gfx950,
sources in ``sdft_amd/csrc``, header ``include/sdft/sdft.h``).  The Python modules are the
host-side mirror used by tests and benchmarks:

* :mod:`sdft_amd.build`   -- in-tree hipcc build
* :mod:`sdft_amd.capi`    -- ctypes prototypes of the C-ABI
* :mod:`sdft_amd.sdft`    -- ``SDFT(dftsize, window, latency).sdft/isdft`` over device pointers
* :mod:`sdft_amd.signals` -- deterministic synthetic inputs
* :mod:`sdft_amd.shard`   -- channel partitioning for one-process-per-GPU runs
"""

__version__ = "0.1.0"

def __getattr__(name):
    if name in ("SDFT", "plan_tables", "check_expr"):
        from . import sdft as _s
        return getattr(_s, name)
    if name in ("COMBOS", "WINDOWS", "SdftHipError"):
        from . import capi as _c
        return getattr(_c, name)
    raise AttributeError(name)
