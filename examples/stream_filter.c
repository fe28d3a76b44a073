/*
 * stream_filter.c -- WAV in, filtered WAV out: the reference's hop-wise streaming driver (test/test.c:69-83)
 * whose three steps per hop -- sdft_sdft_n, a loop of the host over the (hop, dftsize) matrix, sdft_isdft_n --
 * are one call of sdft_hip_process_n with a gain per bin.  The matrix is never formed; a hop costs one kernel
 * launch.
 *
 *   stream_filter <dftsize> <hopsize> <cutoff Hz> <in.wav> <out.wav> [code]
 *   e.g.           1000 100 2000 test.wav lowpassed.wav
 *
 * The mask is a raised-cosine low-pass: 1 below the cut-off, a half-octave roll-off, 0 above.  With the word `code` as the
 * last argument the host does not tabulate the mask: it hands its loop body in as statements (sdft_hip_op_expr), which
 * are compiled into the kernel at run time -- the general form of "the host's loop over the matrix", on the GPU.
 */

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <sdft/sdft.h>

#include "wav.h"

int main(int argc, char** argv)
{
  if (argc != 6 && argc != 7)
  {
    fprintf(stderr, "usage: %s dftsize hopsize cutoff_hz in.wav out.wav [code]\n", argv[0]);
    return 2;
  }
  const int as_code = argc == 7;
  const size_t bins = strtoul(argv[1], NULL, 10), hop = strtoul(argv[2], NULL, 10);
  const double cutoff = strtod(argv[3], NULL);
  if (bins == 0 || hop == 0 || !(cutoff > 0)) return 2;

  float* samples = NULL;
  size_t frames = 0, rate = 0;
  if (!wav_read_mono(argv[4], &samples, &frames, &rate))
  {
    fprintf(stderr, "%s: not a PCM/float WAV file\n", argv[4]);
    return 1;
  }
  const size_t hops = frames / hop;

  /* bin k of an SDFT of `bins` bins sits at k * rate / (2 * bins) Hz */
  sdft_fd_t* mask = (sdft_fd_t*)malloc(bins * sizeof(sdft_fd_t));
  for (size_t k = 0; k < bins; ++k)
  {
    const double hz = (double)k * (double)rate / (2.0 * (double)bins);
    const double edge = cutoff * 1.4142135623730951;
    mask[k] = (sdft_fd_t)(hz <= cutoff ? 1.0 : hz >= edge ? 0.0 : 0.5 * (1.0 + cos(3.141592653589793 * (hz - cutoff) / (edge - cutoff))));
  }

  sdft_t* plan = sdft_alloc_custom(bins, sdft_window_hann, 1);
  if (plan == NULL)
  {
    fprintf(stderr, "sdft_alloc_custom: %s\n", sdft_hip_last_error());
    return 1;
  }
  float* filtered = (float*)calloc(hops * hop + 1, sizeof(float));
  /* the same mask as statements on (re, im) of bin k: p[0] = Hz per bin, p[1] = cut-off, p[2] = end of the roll-off */
  const sdft_fd_t prm[3] = { (sdft_fd_t)((double)rate / (2.0 * (double)bins)), (sdft_fd_t)cutoff, (sdft_fd_t)(cutoff * 1.4142135623730951) };
  const sdft_hip_expr_t code = { "const sdft_fd_t hz = (sdft_fd_t)k * p[0];"
                                 "const sdft_fd_t g = hz <= p[1] ? 1 : hz >= p[2] ? 0 : 0.5 * (1 + cos(3.141592653589793 * (hz - p[1]) / (p[2] - p[1])));"
                                 "re *= g; im *= g;", prm, 3 };
  int ok = 1;
  for (size_t h = 0; h < hops && ok; ++h)
    ok = (as_code ? sdft_hip_process_n(plan, hop, samples + h * hop, filtered + h * hop, sdft_hip_op_expr, &code, NULL)
                  : sdft_hip_process_n(plan, hop, samples + h * hop, filtered + h * hop, sdft_hip_op_gain, mask, NULL)) == 0;
  if (!ok) fprintf(stderr, "sdft_hip_process_n: %s\n", sdft_hip_last_error());

  ok = ok && wav_write_mono_f32(argv[5], filtered, hops * hop, rate);
  printf("C\t%s %zu %zuHz -> %s (low-pass %.0f Hz, %zu hops of %zu)\n", argv[4], frames, rate, argv[5], cutoff, hops, hop);
  sdft_free(plan);
  free(filtered); free(mask); free(samples);
  return ok ? 0 : 1;
}
