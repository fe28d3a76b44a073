"""The CPU-side checker under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5:
sanitizers run on the CPU build only; GPU sanitizers are not available on this pool)."""

import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
#include <stdio.h>
#include <stdlib.h>
typedef struct oracle_plan oracle_plan;
typedef struct { ORACLE_FD re, im; } cx_t;
oracle_plan* oracle_new(size_t, int, double);
void oracle_free(oracle_plan*);
void oracle_reset(oracle_plan*);
void oracle_sdft_n(oracle_plan*, size_t, const ORACLE_TD*, cx_t*);
void oracle_isdft_n(const oracle_plan*, size_t, const cx_t*, ORACLE_TD*);
void oracle_digest_n(oracle_plan*, size_t, const ORACLE_TD*, double*, ORACLE_TD*);
int main(void)
{
  const size_t sizes[] = {1, 2, 3, 5, 64, 100};
  for (int w = 0; w < 4; ++w)
    for (size_t si = 0; si < sizeof(sizes) / sizeof(sizes[0]); ++si)
    {
      const size_t m = sizes[si], n = 5 * m + 3;
      oracle_plan* p = oracle_new(m, w, w % 2 ? 1.0 : 0.5);
      ORACLE_TD* x = malloc(n * sizeof(*x)); ORACLE_TD* y = malloc(n * sizeof(*y));
      cx_t* d = malloc(n * m * sizeof(*d)); double* dig = malloc(n * 4 * sizeof(double));
      for (size_t i = 0; i < n; ++i) x[i] = (ORACLE_TD)((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0);
      oracle_sdft_n(p, n, x, d); oracle_isdft_n(p, n, d, y);
      oracle_reset(p); oracle_digest_n(p, n, x, dig, y);
      free(dig); free(d); free(y); free(x); oracle_free(p);
    }
  puts("SANITIZED-OK");
  return 0;
}
"""


@pytest.mark.parametrize("td,fd", [("float", "double"), ("float", "float"), ("double", "double")])
def test_oracle_clean_under_asan_ubsan(tmp_path, td, fd):
    drv = tmp_path / "drv.c"
    drv.write_text(DRIVER)
    exe = tmp_path / "drv"
    cmd = ["gcc", "-std=gnu99", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           f"-DORACLE_TD={td}", f"-DORACLE_FD={fd}", os.path.join(ROOT, "oracle", "sdft_oracle.c"), str(drv), "-o", str(exe), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr.lower():
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, (r.stdout, r.stderr[-2000:])
