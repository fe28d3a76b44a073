/*
 * host_stream.c -- a plain C host in the calling pattern of the reference's test driver
 * (/root/reference/test/test.c:49-93): alloc_custom -> hop loop { sdft_sdft_n; sdft_isdft_n;
 * keep the first DFT row of the hop } -> free.  Built by tests/test_gpu_chost.py with
 *   gcc -std=c99 -Iinclude [-DSDFT_FD_FLOAT ...] host_stream.c -lsdft_hip -lamdhip64 -lm
 * Input samples come from a raw file, outputs go to raw files, the test compares them with the
 * oracle.  Also exercises single-sample calls, the row-pointer variants and the getters.
 *
 * usage: host_stream <dftsize> <hopsize> <window> <latency> <x.raw> <y.raw> <dft.raw>
 */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sdft/sdft.h>

static sdft_window_t parse_window(const char* s)
{
  if (!strcmp(s, "hann")) return sdft_window_hann;
  if (!strcmp(s, "hamming")) return sdft_window_hamming;
  if (!strcmp(s, "blackman")) return sdft_window_blackman;
  return sdft_window_boxcar;
}

int main(int argc, char* argv[])
{
  if (argc < 8) { fprintf(stderr, "usage\n"); return 2; }
  const size_t dftsize = (size_t)atol(argv[1]);
  const size_t hop = (size_t)atol(argv[2]);
  const sdft_window_t window = parse_window(argv[3]);
  const double latency = atof(argv[4]);

  FILE* f = fopen(argv[5], "rb");
  if (!f) return 3;
  fseek(f, 0, SEEK_END);
  size_t n = (size_t)ftell(f) / sizeof(sdft_td_t);
  fseek(f, 0, SEEK_SET);
  sdft_td_t* x = (sdft_td_t*)malloc(n * sizeof(sdft_td_t));
  if (fread(x, sizeof(sdft_td_t), n, f) != n) return 3;
  fclose(f);
  n = (n / hop) * hop;

  /* NULL tolerance of the getters / free (reference sdft.h:466-554) */
  if (sdft_size(NULL) != 0 || sdft_window(NULL) != sdft_window_boxcar || sdft_latency(NULL) != 0) return 4;
  sdft_free(NULL);

  sdft_t* sdft = sdft_alloc_custom(dftsize, window, latency);
  if (!sdft) { fprintf(stderr, "alloc failed: %s\n", sdft_hip_last_error()); return 5; }
  if (sdft_size(sdft) != dftsize || sdft_window(sdft) != window || sdft_latency(sdft) != latency) return 6;

  sdft_td_t* y = (sdft_td_t*)malloc(n * sizeof(sdft_td_t));
  sdft_fdx_t* buffer = (sdft_fdx_t*)malloc(hop * dftsize * sizeof(sdft_fdx_t));
  sdft_fdx_t* dfts = (sdft_fdx_t*)malloc((n / hop) * dftsize * sizeof(sdft_fdx_t));
  sdft_fdx_t** rows = (sdft_fdx_t**)malloc(hop * sizeof(sdft_fdx_t*));
  for (size_t r = 0; r < hop; ++r) rows[r] = buffer + r * dftsize;

  for (size_t i = 0, j = 0; i < n; i += hop, ++j)
  {
    switch (j % 3)
    {
      case 0:                                   /* dense matrix calls */
        sdft_sdft_n(sdft, hop, x + i, buffer);
        sdft_isdft_n(sdft, hop, buffer, y + i);
        break;
      case 1:                                   /* row-pointer calls */
        sdft_sdft_nd(sdft, hop, x + i, rows);
        sdft_isdft_nd(sdft, hop, (const sdft_fdx_t**)rows, y + i);
        break;
      default:                                  /* sample-by-sample calls */
        for (size_t t = 0; t < hop; ++t)
        {
          sdft_sdft(sdft, x[i + t], buffer + t * dftsize);
          y[i + t] = sdft_isdft(sdft, buffer + t * dftsize);
        }
        break;
    }
    memcpy(dfts + j * dftsize, buffer, dftsize * sizeof(sdft_fdx_t));
  }
  if (sdft_hip_last_error()) { fprintf(stderr, "error: %s\n", sdft_hip_last_error()); return 7; }

  f = fopen(argv[6], "wb"); fwrite(y, sizeof(sdft_td_t), n, f); fclose(f);
  f = fopen(argv[7], "wb"); fwrite(dfts, sizeof(sdft_fdx_t), (n / hop) * dftsize, f); fclose(f);

  /* reset brings the plan back to t = 0 */
  sdft_reset(sdft);
  sdft_sdft_n(sdft, hop, x, buffer);
  if (memcmp(buffer, dfts, dftsize * sizeof(sdft_fdx_t)) != 0) { fprintf(stderr, "reset mismatch\n"); return 8; }

  free(rows); free(dfts); free(buffer); free(y); free(x);
  sdft_free(sdft);
  printf("C-HOST ok n=%zu hops=%zu\n", n, n / hop);
  return 0;
}
