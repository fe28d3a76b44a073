"""Builds libsdft_hip.so (the C-ABI + HIP kernels) and its test-hooks flavour libsdft_hip_hooks.so in-tree with hipcc for gfx950.

    python -m sdft_amd.build [--force] [--save-temps]

The library has no Python or torch dependency; it is what a C host links against
(INTEGRATION.md).  -ffp-contract=off is part of the numerical contract: the kernels must not
fuse a*b+c, otherwise float-FD results drift from the reference by more than the parity bar
(hard part 1).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libsdft_hip.so")
# the same sources with -DSDFT_HIP_TEST_HOOKS: sdft_hip_set_option then also knows the keys that force every remaining fork of the host
# logic (sdft_capi.inc) -- what the tests run against the reference route by route and the probes under scripts/ measure; no host links it
LIB_HOOKS = os.path.join(OUT_DIR, "libsdft_hip_hooks.so")
ARCH = "gfx950"
COMBOS = ("f32f64", "f32f32", "f64f64", "f64f32")
SOURCES = ["sdft_common.hip"] + [f"sdft_capi_{c}.hip" for c in COMBOS]
KERNEL_FILES = ["sdft_base.hpp", "sdft_carry_fast.hpp", "sdft_carry_exact.hpp", "sdft_forward.hpp", "sdft_forward_hop.hpp", "sdft_ops.hpp",
                "sdft_forward_rows.hpp", "sdft_fused.hpp", "sdft_inverse.hpp"]      # in include order (sdft_kernels.hpp)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc")))            # everything a translation unit may include
# -fno-slp-vectorize: on this VALU a packed f32 instruction costs what two plain ones cost and its operands have to be
# assembled by moves (profiles/r04_valu_issue_rates.txt); the vectoriser packs scalar float code all the same -- the generic
# FD float row kernel runs 28.3 -> 33.9 GB/s per CU without it (profiles/r04_kernels_beside_held_cus.txt).  Kernels that
# want packed operands write them as vector types (sdft_forward_rows_f32.hpp).
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsdft_hip.so cannot be built (no CPU fallback exists)")


def kernel_source() -> str:
    """sdft_kernels.hpp with its stage files inlined: what hiprtc compiles a host's statements into."""
    parts = []
    for name in KERNEL_FILES:
        with open(os.path.join(CSRC, name)) as fh:
            own = tuple(f'#include "{k}"' for k in KERNEL_FILES)         # (not "sdft_user_expr.inc": the host's statements)
            lines = [l for l in fh.read().split("\n") if not (l.startswith(own) or l.strip() == "#pragma once")]
        parts.append(f"// ---- {name} ----\n" + "\n".join(lines))
    return "\n".join(parts)


def _source_hash() -> str:
    """Identity of what the library is built from: the CONTENTS of every source (a snapshot copied to another machine does
    not keep modification times in any useful order: the GPU box used to rebuild a library that was up to date)."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + [os.path.abspath(__file__)]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        h.update(os.path.basename(f).encode() + b"\0")             # (the name, not the path: the tree may live anywhere)
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


STAMP = os.path.join(OUT_DIR, "libsdft_hip.sources.sha256")


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(LIB_HOOKS) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _source_hash()


def build(force: bool = False, save_temps: bool = False, verbose: bool = False,
          extra_flags=tuple(os.environ.get("SDFT_HIP_EXTRA_FLAGS", "").split()), hooks: bool = False) -> str:
    """Compile (if stale) both flavours and return the path of libsdft_hip.so (hooks: of libsdft_hip_hooks.so)."""
    force = force or bool(extra_flags)                    # development flags: never hand back a library built without them
    want = LIB_HOOKS if hooks else LIB
    if not force and not _stale():
        return want
    os.makedirs(OUT_DIR, exist_ok=True)
    # one builder at a time (every rank of a torchrun job may arrive here at once): the others wait
    # on the lock and then find a fresh library; the link goes to a temporary name and is renamed
    # into place, so no process can ever map a half-written libsdft_hip.so
    import fcntl
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return want
        _build_locked(save_temps, verbose, extra_flags)
        return want


def _build_locked(save_temps, verbose, extra_flags) -> str:
    obj_dir = os.path.join(OUT_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()
    # the text of the kernels, for the run-time compilation of a host's own spectral operation (sdft_hip_process_expr_n):
    # the stage files in include order with their quoted includes removed (one self-contained source), as a raw string
    # literal in pieces (compilers bound the length of one literal), included by sdft_common.hip
    text = kernel_source()
    assert ')SDFTSRC"' not in text
    pieces = [text[i:i + 8000] for i in range(0, len(text), 8000)]
    with open(os.path.join(obj_dir, "sdft_kernels_src.inc"), "w") as fh:
        fh.write("\n".join(f'R"SDFTSRC({piece})SDFTSRC"' for piece in pieces) + "\n")

    def compile_one(job) -> str:
        src, hooks = job
        obj = os.path.join(obj_dir, src.replace(".hip", ".hooks.o" if hooks else ".o"))
        cmd = [cc, *FLAGS, *extra_flags, *(["-DSDFT_HIP_TEST_HOOKS"] if hooks else []), f"-I{obj_dir}", "-c", os.path.join(CSRC, src), "-o", obj]
        if save_temps:
            cmd.insert(1, "-save-temps=obj")
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=obj_dir)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    # (sdft_common.hip holds nothing a hook changes: one object serves both libraries)
    jobs = [(src, False) for src in SOURCES] + [(src, True) for src in SOURCES[1:]]
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, jobs))
    plain, hooked = objs[:len(SOURCES)], [objs[0]] + objs[len(SOURCES):]
    # -no-hip-rt: no DT_NEEDED on a particular libamdhip64.  A process must hold exactly one HIP
    # runtime (two cannot both open the GPU); PyTorch wheels bundle their own under a different
    # soname than /opt/rocm's.  The host decides: a C program links -lamdhip64 itself
    # (INTEGRATION.md), capi.load() binds to the runtime already loaded in the interpreter.
    for lib, members in ((LIB, plain), (LIB_HOOKS, hooked)):
        tmp = lib + f".tmp{os.getpid()}"
        r = subprocess.run([cc, "-shared", "-fPIC", "-no-hip-rt", f"--offload-arch={ARCH}", *members, "-o", tmp, "-lm", "-ldl"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        os.replace(tmp, lib)
    if not extra_flags:                                   # (a development build is never mistaken for the product)
        with open(STAMP, "w") as fh:
            fh.write(_source_hash() + "\n")
    elif os.path.exists(STAMP):
        os.remove(STAMP)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True)
    print(path)
