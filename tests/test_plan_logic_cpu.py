"""Host-side logic of the engine on the CPU under sanitizers (round-4 review: "none of the product's host
C++ runs under ASan/UBSan/TSan").  sdft_plan_logic.hpp holds every decision of the host side that needs no HIP call (launch
geometry, time chunking, relay block length, which calls leave the plan's stream, hop parts, rows per wave, the synchronous
wait, the slot ring of the host copies); sdft_copy_engine.hpp the worker pool and the pipelined copies through pinned slots,
with the device as a policy.  Both are compiled here by g++ -- no hipcc, no GPU -- and run with their property tests."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdft_amd", "csrc")


def build_and_run(tmp_path, source, sanitize, args=(), timeout=600):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++ on this host")
    exe = str(tmp_path / (os.path.splitext(source)[0] + "_" + sanitize.replace(",", "_")))
    cmd = [gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={sanitize}", "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-Werror",
           f"-I{CSRC}", os.path.join(ROOT, "tests", "cpp", source), "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def test_plan_logic_properties_under_asan_ubsan(tmp_path):
    out = build_and_run(tmp_path, "plan_logic_test.cpp", "address,undefined")
    assert "all properties hold" in out


def test_copy_engine_under_tsan(tmp_path):
    """Worker pool + slot ring against a mock device that executes DMAs on a thread of its own: a slot touched by a worker
    while a DMA still reads or writes it is a data race ThreadSanitizer reports."""
    out = build_and_run(tmp_path, "copy_engine_test.cpp", "thread", args=(12,))
    assert "all copies arrived" in out


def test_run_time_compilation_cache_under_tsan(tmp_path):
    """sdft_keyed_once.hpp (the cache of run-time-compiled kernels: one compilation per key, outside the lock, same-key callers
    wait for it) with sixteen threads on eight keys."""
    out = build_and_run(tmp_path, "keyed_once_test.cpp", "thread")
    assert "every key made once" in out


def test_copy_engine_under_asan_ubsan(tmp_path):
    out = build_and_run(tmp_path, "copy_engine_test.cpp", "address,undefined", args=(60,))
    assert "all copies arrived" in out


def test_logic_header_has_no_hip_dependency():
    """The point of the split: these two headers compile without the HIP toolchain (the compile steps above prove it for the
    tests' include set; this pins the source text)."""
    for name in ("sdft_plan_logic.hpp", "sdft_copy_engine.hpp", "sdft_keyed_once.hpp"):
        text = open(os.path.join(CSRC, name)).read()
        assert "#include <hip" not in text and "hipStream" not in text and "__global__" not in text, name
    plan = open(os.path.join(CSRC, "sdft_plan.hpp")).read()
    assert '#include "sdft_plan_logic.hpp"' in plan
