/*
 * sdft_wav.c -- WAV in, WAV + DFT dump out: hop-wise analysis and resynthesis with persistent plan
 * state on the GPU.  Same command line and same outputs as the reference's end-to-end test driver
 * (positional arguments of test/test.c:41-47; first DFT row of every hop dumped as raw interleaved
 * complex doubles like test/dump.h), so its comparison script can be pointed at these files:
 *
 *   sdft_wav <dftsize> <hopsize> <boxcar|hann|hamming|blackman> <latency> <in.wav> <out.wav> <out.dft>
 *   e.g.     1000 100 hann 1 test.wav test.hip.wav test.hip.dft        (reference test/main.sh:3-6)
 */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sdft/sdft.h>

#include "wav.h"

struct job
{
  size_t bins, hop;
  sdft_window_t window;
  double latency;
  const char *wav_in, *wav_out, *dft_out;
};

static int parse(int argc, char** argv, struct job* j)
{
  static const struct { const char* name; sdft_window_t id; } windows[] = {
    { "boxcar", sdft_window_boxcar }, { "hann", sdft_window_hann },
    { "hamming", sdft_window_hamming }, { "blackman", sdft_window_blackman } };
  if (argc != 8) return 0;
  j->bins = strtoul(argv[1], NULL, 10);
  j->hop = strtoul(argv[2], NULL, 10);
  j->window = sdft_window_boxcar;                      /* unknown names fall back to boxcar */
  for (size_t w = 0; w < sizeof(windows) / sizeof(windows[0]); ++w)
    if (strcmp(argv[3], windows[w].name) == 0) j->window = windows[w].id;
  j->latency = strtod(argv[4], NULL);
  j->wav_in = argv[5]; j->wav_out = argv[6]; j->dft_out = argv[7];
  return j->bins > 0 && j->hop > 0;
}

/* analyse + resynthesise `hops` hops; spectra of one hop live in `scratch` (hop x bins) */
static void stream(sdft_t* plan, const struct job* j, const float* in, float* out, size_t hops,
                   sdft_fdx_t* scratch, FILE* dump)
{
  for (size_t h = 0; h < hops; ++h)
  {
    const size_t at = h * j->hop;
    sdft_sdft_n(plan, j->hop, in + at, scratch);
    sdft_isdft_n(plan, j->hop, scratch, out + at);
    if (dump) fwrite(scratch, sizeof(sdft_fdx_t), j->bins, dump);      /* row 0 of the hop */
  }
}

int main(int argc, char** argv)
{
  struct job j;
  if (!parse(argc, argv, &j))
  {
    fprintf(stderr, "usage: %s dftsize hopsize window latency in.wav out.wav out.dft\n", argv[0]);
    return 2;
  }
  float* samples = NULL;
  size_t frames = 0, rate = 0;
  if (!wav_read_mono(j.wav_in, &samples, &frames, &rate))
  {
    fprintf(stderr, "%s: not a PCM/float WAV file\n", j.wav_in);
    return 1;
  }
  const size_t hops = frames / j.hop;                  /* whole hops only */
  printf("C\t%s %zu %zuHz\n", j.wav_in, frames, rate);

  sdft_t* plan = sdft_alloc_custom(j.bins, j.window, j.latency);
  if (plan == NULL)
  {
    fprintf(stderr, "sdft_alloc_custom: %s\n", sdft_hip_last_error());
    free(samples);
    return 1;
  }
  float* resynth = (float*)calloc(hops * j.hop + 1, sizeof(float));
  sdft_fdx_t* scratch = (sdft_fdx_t*)malloc(j.hop * j.bins * sizeof(sdft_fdx_t));
  FILE* dump = fopen(j.dft_out, "wb");

  stream(plan, &j, samples, resynth, hops, scratch, dump);

  if (dump) fclose(dump);
  const int ok = wav_write_mono_f32(j.wav_out, resynth, hops * j.hop, rate) && sdft_hip_last_error() == NULL;
  sdft_free(plan);
  free(scratch); free(resynth); free(samples);
  return ok ? 0 : 1;
}
