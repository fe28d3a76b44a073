/*
 * host_process.c -- a plain C host that replaces the reference's
 *     sdft_sdft_n -> own loop over the (n, N) matrix -> sdft_isdft_n          (README.md:42-47)
 * by the fused call sdft_hip_process_n (include/sdft/sdft_hip.h) and, for comparison, runs the three
 * steps through the drop-in functions on a second plan.  Built by tests/test_gpu_chost.py with
 *   gcc -std=c99 -Iinclude [-DSDFT_FD_FLOAT] host_process.c -lsdft_hip -lamdhip64 -lm
 *
 * usage: host_process <dftsize> <hopsize> <op: 0 identity | 1 gain | 2 shift | 3 complex gain> <x.raw> <y_fused.raw> <y_threestep.raw>
 */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SDFT_NO_COMPLEX_H
#include <sdft/sdft.h>

int main(int argc, char* argv[])
{
  if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
  const size_t m = (size_t)atol(argv[1]);
  const size_t hop = (size_t)atol(argv[2]);
  const int op = atoi(argv[3]);

  FILE* f = fopen(argv[4], "rb");
  if (!f) return 3;
  fseek(f, 0, SEEK_END);
  size_t n = (size_t)ftell(f) / sizeof(sdft_td_t);
  fseek(f, 0, SEEK_SET);
  sdft_td_t* x = (sdft_td_t*)malloc(n * sizeof(sdft_td_t));
  if (fread(x, sizeof(sdft_td_t), n, f) != n) return 3;
  fclose(f);
  n = (n / hop) * hop;

  sdft_fd_t* gain = (sdft_fd_t*)malloc(m * sizeof(sdft_fd_t));
  for (size_t k = 0; k < m; ++k) gain[k] = (sdft_fd_t)(1.0 / (1.0 + (double)k / 64.0));
  const long shift = 3;
  /* complex factors: the real ones with the phase of a two-sample delay at every other bin (exact in binary) */
  sdft_fdx_t* cgain = (sdft_fdx_t*)malloc(m * sizeof(sdft_fdx_t));
  for (size_t k = 0; k < m; ++k)
  {
    cgain[k].r = (k % 4 == 0) ? gain[k] : (k % 4 == 2 ? -gain[k] : 0);
    cgain[k].i = (k % 4 == 1) ? gain[k] : (k % 4 == 3 ? -gain[k] : 0);
  }
  const void* params = op == 1 ? (const void*)gain : (op == 2 ? (const void*)&shift : (op == 3 ? (const void*)cgain : NULL));

  sdft_t* fused = sdft_alloc_custom(m, sdft_window_hann, 1);
  sdft_t* plain = sdft_alloc_custom(m, sdft_window_hann, 1);
  if (!fused || !plain) { fprintf(stderr, "alloc failed: %s\n", sdft_hip_last_error()); return 5; }

  sdft_td_t* y1 = (sdft_td_t*)malloc(n * sizeof(sdft_td_t));
  sdft_td_t* y2 = (sdft_td_t*)malloc(n * sizeof(sdft_td_t));
  sdft_fdx_t* dfts = (sdft_fdx_t*)malloc(hop * m * sizeof(sdft_fdx_t));
  sdft_fdx_t* tmp = (sdft_fdx_t*)malloc(m * sizeof(sdft_fdx_t));

  for (size_t i = 0; i < n; i += hop)
  {
    /* one call, no matrix */
    if (sdft_hip_process_n(fused, hop, x + i, y1 + i, op, params, NULL) != 0)
    { fprintf(stderr, "process_n: %s\n", sdft_hip_last_error()); return 6; }

    /* the reference's three steps */
    sdft_sdft_n(plain, hop, x + i, dfts);
    for (size_t t = 0; t < hop; ++t)
    {
      sdft_fdx_t* row = dfts + t * m;
      if (op == 1) for (size_t k = 0; k < m; ++k) { row[k].r *= gain[k]; row[k].i *= gain[k]; }
      if (op == 3)
        for (size_t k = 0; k < m; ++k)
        {
          const sdft_fdx_t v = row[k], g = cgain[k];
          row[k].r = v.r * g.r - v.i * g.i;
          row[k].i = v.r * g.i + v.i * g.r;
        }
      if (op == 2)
      {
        for (size_t k = 0; k < m; ++k)
        {
          const long j = (long)k - shift;
          if (j >= 0 && j < (long)m) tmp[k] = row[j]; else { tmp[k].r = 0; tmp[k].i = 0; }
        }
        memcpy(row, tmp, m * sizeof(sdft_fdx_t));
      }
    }
    sdft_isdft_n(plain, hop, dfts, y2 + i);
  }
  if (sdft_hip_last_error()) { fprintf(stderr, "error: %s\n", sdft_hip_last_error()); return 7; }

  f = fopen(argv[5], "wb"); fwrite(y1, sizeof(sdft_td_t), n, f); fclose(f);
  f = fopen(argv[6], "wb"); fwrite(y2, sizeof(sdft_td_t), n, f); fclose(f);
  free(tmp); free(dfts); free(y2); free(y1); free(cgain); free(gain); free(x);
  sdft_free(fused); sdft_free(plain);
  printf("C-PROCESS ok n=%zu hops=%zu\n", n, n / hop);
  return 0;
}
