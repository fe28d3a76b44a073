"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol the
headers declare, and its host-only logic (plan tables) equals the oracle's.  No GPU compute."""

import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def declared_symbols():
    """Names under SDFT_HIP_SYMBOL(...) in the headers (typed) + plain sdft_hip_* prototypes (untyped)."""
    typed, untyped = set(), set()
    for h in ("sdft/sdft.h", "sdft/sdft_hip.h"):
        text = open(os.path.join(INC, h)).read()
        typed |= set(re.findall(r"SDFT_HIP_SYMBOL\((\w+)\);", text))
        for m in re.finditer(r"^\s*(?:const\s+char\*|int|void)\s+(sdft_hip_\w+)\s*\([^;]*\);", text, re.M):
            if "SDFT_HIP_SYMBOL" not in m.group(0):
                untyped.add(m.group(1))
    return typed, untyped


def test_library_exports_every_declared_symbol(hip_library):
    from sdft_amd import capi
    lib = capi.load()
    typed, untyped = declared_symbols()
    assert {"alloc", "alloc_custom", "free", "reset", "size", "window", "latency", "sdft", "sdft_n", "sdft_nd",
            "isdft", "isdft_n", "isdft_nd"} <= typed                       # the reference surface, sdft.h:39-58
    assert {"sdft_hip_last_error", "sdft_hip_device_count"} <= untyped
    for combo in capi.COMBOS:
        for name in typed:
            assert hasattr(lib, capi.symbol(name, combo)), (name, combo)
    for name in untyped:
        assert hasattr(lib, name), name
    # and the binding declares exactly the typed set
    assert set(capi.typed_signatures("f32f64")) == typed


def test_library_has_no_hard_hip_runtime_dependency(hip_library):
    out = subprocess.run(["readelf", "-d", hip_library], capture_output=True, text=True).stdout
    assert "libamdhip64" not in out


@pytest.mark.parametrize("combo", O.COMBOS)
def test_host_plan_tables_bit_identical_to_oracle(hip_library, combo):
    from sdft_amd.sdft import plan_tables
    for m in (1, 2, 3, 64, 1000, 1024, 4096):
        for latency in (1.0, 0.5, 0.25, 0.77):
            tw, syn, wtab, w = plan_tables(m, latency, combo)
            otw, osyn, ow = O.Port(m, "hann", latency, combo).tables()
            assert np.array_equal(tw, otw) and np.array_equal(syn, osyn) and np.array_equal(w, ow), (m, latency)
            # closed-form rotation table: first half is the twiddle table, all entries on the unit circle
            assert np.array_equal(wtab[:m], tw)
            j = np.arange(2 * m)
            assert np.allclose(wtab, np.exp(-1j * np.pi * j / m), atol=1e-6 if combo.endswith("f32") else 1e-15)


def test_null_plan_getters_without_gpu(hip_library):
    """Reference semantics for NULL plans (sdft.h:466-554) need no device."""
    from sdft_amd import capi
    api = capi.Api("f32f64")
    assert api.size(None) == 0 and api.window(None) == 0 and api.latency(None) == 0.0
    api.free(None)
    api.reset(None)


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text("#include <sdft/sdft.h>\nint main(void){ sdft_t* p = sdft_alloc(8); sdft_free(p); return (int)sizeof(sdft_fdx_t); }\n")
    for flags in ([], ["-DSDFT_FD_FLOAT"], ["-DSDFT_TD_DOUBLE", "-DSDFT_NO_COMPLEX_H"]):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, *flags, "-c", str(src), "-o", str(tmp_path / "t.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        nm = subprocess.run(["nm", str(tmp_path / "t.o")], capture_output=True, text=True).stdout
        suffix = ("f64" if "-DSDFT_TD_DOUBLE" in flags else "f32") + ("f32" if "-DSDFT_FD_FLOAT" in flags else "f64")
        assert f"sdft_hip_alloc_{suffix}" in nm
    r = subprocess.run(["g++", "-x", "c++", "-fsyntax-only", "-I", INC, os.path.join(INC, "sdft", "sdft.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_cpp_facade_compiles_for_all_type_pairs(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text("#include <sdft/sdft.h>\n"
                   "template <class T, class F> int use() { sdft::SDFT<T, F> s(8, sdft::Window::Blackman, 0.5); std::complex<F> d[8]; T y;"
                   " s.sdft(T(1), d); y = s.isdft(d); s.reset(); return (int)s.size() + (int)y; }\n"
                   "int main() { return use<float, double>() + use<float, float>() + use<double, double>() + use<double, float>(); }\n")
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(INC, "cpp"), "-c", str(src), "-o", str(tmp_path / "t.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", str(tmp_path / "t.o")], capture_output=True, text=True).stdout
    for suf in ("f32f64", "f32f32", "f64f64", "f64f32"):
        assert f"sdft_hip_sdft_n_{suf}" in nm or f"sdft_hip_sdft_{suf}" in nm


FUSED_OPS = r"\b(v_fma_f32|v_fmac_f32|v_pk_fma_f32|v_fma_f64|v_fmac_f64|v_mad_f32|v_mac_f32|v_fma_legacy_f32|v_mad_legacy_f32|v_fma_mix\w*|v_dot\w*)\b"


def disassemble(combo, hip_library):
    """{demangled kernel name: [instruction lines]} of one translation unit's gfx950 code object."""
    import shutil
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(os.path.dirname(hip_library), "obj", f"sdft_capi_{combo}.o")
    if not (os.path.exists(objdump) and os.path.exists(obj) and shutil.which("c++filt")):
        pytest.skip("llvm-objdump / c++filt / object files not available")
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, "dev.o")
        shutil.copy(obj, local)
        subprocess.run([objdump, "--offloading", local], capture_output=True, text=True, cwd=td)
        cos = [f for f in os.listdir(td) if "amdgcn" in f and "gfx950" in f]
        if not cos:
            pytest.skip("could not extract the gfx950 code object")
        dis = subprocess.run([objdump, "-d", os.path.join(td, cos[0])], capture_output=True, text=True).stdout
    kernels, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.strip():
            kernels[name].append(line.split("//")[0].strip())
    mangled = list(kernels)
    plain = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(sdfthip::.*$", "", d).replace("void sdfthip::", ""): kernels[k] for k, d in zip(mangled, plain)}


@pytest.mark.parametrize("combo", ("f32f64", "f32f32", "f64f64", "f64f32"))
def test_kernels_contain_no_fused_multiply_add(hip_library, combo):
    """Parity depends on unfused a*b+c in the recurrence, the window and the synthesis terms (SURVEY.md
    section 7): every kernel of every (TD, FD) translation unit is disassembled and must contain NO fused
    floating point instruction at all -- except the places that ask for them by name: the FUSED
    instantiations of forward_rows_kernel and process_rows_kernel (chunk-parallel FD double path only) and
    chunk_sum_kernel (feeds carries whose summation order differs from the reference anyway)."""
    kernels = disassemble(combo, hip_library)
    assert any(k.startswith("forward_rows_kernel") for k in kernels) and any(k.startswith("inverse_exact_kernel") for k in kernels)
    checked = 0
    for name, body in kernels.items():
        args = re.search(r"<(.*)>", name)
        params = [a.strip() for a in args.group(1).split(",")] if args else []
        deliberately_fused = name.startswith("chunk_sum_kernel") or \
            (name.startswith(("forward_rows_kernel", "process_rows_kernel")) and params[3] == "true")
        # round 3: the magnitude power law of sdft_hip_process_n evaluates |X|^(p-1) by polynomials in fused multiply-adds
        # (pow_positive); it is inlined into the kernels that apply spectral operations to windowed bins -- the fused-synthesis
        # instantiations of the row-group kernel (SYN != 0), the OPS instantiations of the synthesis kernels and
        # scale_rows_kernel.  Their reference arithmetic is the same source as their SYN = 0 / OPS = false twins', which stay
        # under the zero-FMA rule below (and the bit-exact GPU tests cover the fused ones: tests/test_gpu_process.py).
        carries_pow = (name.startswith("forward_rows_kernel") and len(params) > 5 and params[5] != "0") or \
            (name.startswith(("inverse_kernel", "inverse_row_kernel")) and params[3] == "true") or \
            (name.startswith("inverse_exact_kernel") and params[5] == "true") or \
            name.startswith("scale_rows_kernel")
        if carries_pow and not deliberately_fused:
            continue
        fused = [l for l in body if re.search(FUSED_OPS, l)]
        if deliberately_fused:
            continue
        assert not fused, (combo, name, fused[:4])
        checked += 1
        if name.startswith(("forward_kernel", "forward_hop_kernel", "forward_hop2_kernel", "forward_rows_kernel", "carry_exact_kernel")):
            assert any(re.search(r"\bv_(pk_)?mul_f(32|64)(_e32|_e64|_dpp|_sdwa)?\b", l) for l in body), name       # the arithmetic is really there
    assert checked >= 40


@pytest.mark.parametrize("combo", ("f32f64", "f32f32", "f64f64", "f64f32"))
def test_the_kernels_of_the_two_reference_calls_do_not_spill(hip_library, combo):
    """The row-group analysis kernel sits at the 128 registers a 16-wave workgroup can have; one more value alive across its
    prologue and the compiler starts spilling (round 4: the self-carried form for double samples spilled 48 registers after an
    unrelated change, sdft_sdft_n at the reference's bench.cpp shape went from 147 to 170 us and only the bench noticed).
    No scratch instruction in the kernels of sdft_sdft_n / sdft_isdft_n: the analysis kernels without fused synthesis (SYN = 0),
    the bin-pair kernel, the state kernel, the synthesis kernels (exact order, tree sum, whole rows in step: the one-bin form is held to 64 registers), the relay."""
    kernels = disassemble(combo, hip_library)
    checked = 0
    for name, body in kernels.items():
        args = re.search(r"<(.*)>", name)
        params = [a.strip() for a in args.group(1).split(",")] if args else []
        hot = (name.startswith("forward_rows_kernel") and len(params) > 5 and params[5] == "0") or \
            name.startswith(("forward_rows_f32_kernel", "self_state_kernel", "inverse_exact_kernel", "inverse_rows1_kernel", "inverse_rows2_kernel", "inverse_kernel",
                             "carry_relay_kernel", "forward_hop2_kernel"))
        if not hot:
            continue
        spills = [l for l in body if "scratch_" in l]
        assert not spills, (combo, name, len(spills), spills[:2])
        checked += 1
    assert checked >= 20, checked


def test_hand_written_sequences_are_in_place(hip_library):
    """The FD float exact passes are spelled out in ISA (pinned registers, DPP lane-pair multiplies);
    a compiler bump that drops or rewrites them must not pass silently."""
    for combo in ("f32f32", "f64f32"):
        kernels = disassemble(combo, hip_library)
        serial = "\n".join(kernels["carry_exact_kernel<float>"])
        assert len(re.findall(r"v_mul_f32_dpp v40, v40, v\d+ quad_perm:\[1,0,3,2\]", serial)) >= 32      # 32 steps per trip
        assert len(re.findall(r"v_pk_add_f32 v\[40:41\], v\[42:43\], v\[40:41\]", serial)) >= 32
        assert len(re.findall(r"s_load_dwordx16 s\[(64:79|80:95)\]", serial)) >= 3
        for L in (8, 16, 32, 64, 128):                           # every block length of the relay form
            relay = "\n".join(kernels[f"carry_relay_kernel<float, {L}, false>"])
            assert len(re.findall(r"v_mul_f32_dpp v\d+, v\d+, v\d+ quad_perm:\[1,0,3,2\]", relay)) >= L
            assert "v_pk_mul_f32" not in relay and "v_pk_add_f32" not in relay                 # what the spelling-out prevents


def test_relay_form_is_what_it_claims(hip_library):
    """carry_relay_kernel keeps a block's products in registers and the chain free of everything but additions: the
    products pick their difference with the DPP row broadcast (no scalar operand, no v_readlane), the chain is L dependent
    additions back to back, the wait for the token is one ds_read per poll -- and there is no LDS traffic for products."""
    k = disassemble("f32f32", hip_library)
    body = k["carry_relay_kernel<float, 128, false>"]
    text = "\n".join(body)
    assert len(re.findall(r"v_mul_f32_dpp v\d+, v\d+, v\d+ row_newbcast:\d+", text)) >= 128
    assert len(re.findall(r"v_mul_f32_dpp v\d+, v\d+, v\d+ quad_perm:\[1,0,3,2\]", text)) >= 128
    assert "v_pk_mul_f32" not in text and "v_pk_add_f32" not in text and "ds_read_b128" not in text and "ds_write_b128" not in text
    # the chain: a run of at least 128 consecutive dependent v_add_f32 on one accumulator
    best = run = 0
    for line in body:
        run = run + 1 if re.match(r"v_add_f32_e32 v\d+, v\d+, v\d+", line) else 0
        best = max(best, run)
    assert best >= 128, best
    assert len(re.findall(r"ds_read_b64 v\[4:5\]", text)) >= 1 and "flat_load" not in text
    d = disassemble("f32f64", hip_library)
    dbl = "\n".join(d["carry_relay_kernel<double, 64, false>"])
    assert len(re.findall(r"v_mov_b32_dpp v1[23], v\d+ row_newbcast:\d+", dbl)) >= 128 and "scratch_" not in dbl


def test_expression_operation_compiles_without_a_gpu(hip_library):
    """sdft_hip_op_expr: the library carries the text of its kernels and compiles a host's statements into them at run time
    (hiprtc).  No GPU is needed to compile: good statements give a code object for the two-pass kernel (both bin types) and
    the fused kernel, bad ones the compiler's words through sdft_hip_last_error()."""
    import ctypes
    try:
        ctypes.CDLL("libhiprtc.so")
    except OSError:
        try:
            ctypes.CDLL("/opt/rocm/lib/libhiprtc.so")
        except OSError:
            pytest.skip("no libhiprtc.so here")
    from sdft_amd import capi
    lib = capi.load()
    lib.sdft_hip_clear_error()
    good = b"const sdft_fd_t m2 = re * re + im * im; const sdft_fd_t g = m2 / (m2 + p[0] * (1 + k)) * cos((sdft_fd_t)t * p[1]) * (1 + ch); re *= g; im *= g;"
    assert lib.sdft_hip_check_expr(good, None) == 0, lib.sdft_hip_last_error()
    assert lib.sdft_hip_check_expr(good, b"gfx942") == 0, lib.sdft_hip_last_error()       # the target is the device's, whatever it is
    assert lib.sdft_hip_check_expr(b"re = no_such_thing;", None) == -1
    text = lib.sdft_hip_last_error().decode()
    assert "does not compile" in text and "no_such_thing" in text and "sdft_user_expr.inc:1" in text
    lib.sdft_hip_clear_error()
    assert lib.sdft_hip_check_expr(b"", None) == -1 and lib.sdft_hip_check_expr(None, None) == -1
    lib.sdft_hip_clear_error()
    import sdft_amd
    sdft_amd.check_expr("re = im; im = 0;")
    with pytest.raises(sdft_amd.SdftHipError):
        sdft_amd.check_expr("this is not C++")
