/*
 * sdft_wav.c -- the reference's end-to-end test driver (test/test.c:34-96) as a tool on top of
 * libsdft_hip.so: read a WAV file, analyse and resynthesise it hop by hop with persistent state,
 * write the synthesised WAV and a dump of the first DFT row of every hop (raw interleaved complex
 * doubles, like test/dump.h:12-28).
 *
 * usage: sdft_wav <dftsize> <hopsize> <window> <latency> <src.wav> <out.wav> <out.dft>
 *        (the reference's run: 1000 100 hann 1 test.wav test.c.wav test.c.dft -- test/main.sh:3-6,20)
 */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sdft/sdft.h>

#include "wav.h"

static sdft_window_t getwindow(const char* w)
{
  if (!strcmp(w, "hann")) return sdft_window_hann;
  if (!strcmp(w, "hamming")) return sdft_window_hamming;
  if (!strcmp(w, "blackman")) return sdft_window_blackman;
  return sdft_window_boxcar;
}

int main(int argc, char* argv[])
{
  if (argc < 8) { fprintf(stderr, "usage: %s dftsize hopsize window latency src.wav out.wav out.dft\n", argv[0]); return 1; }
  const size_t dftsize = (size_t)atol(argv[1]), hopsize = (size_t)atol(argv[2]);
  float* input; size_t size, sr;
  if (!wav_read_mono(argv[5], &input, &size, &sr)) { fprintf(stderr, "cannot read %s\n", argv[5]); return 1; }
  printf("C\t%s %zu %zuHz\n", argv[5], size, sr);
  size = (size / hopsize) * hopsize;

  sdft_t* sdft = sdft_alloc_custom(dftsize, getwindow(argv[3]), atof(argv[4]));
  if (!sdft) { fprintf(stderr, "no plan: %s\n", sdft_hip_last_error()); return 1; }

  float* output = (float*)malloc((size ? size : 1) * sizeof(float));
  sdft_fdx_t* buffer = (sdft_fdx_t*)malloc(hopsize * dftsize * sizeof(sdft_fdx_t));
  sdft_fdx_t* dfts = (sdft_fdx_t*)malloc((size / hopsize + 1) * dftsize * sizeof(sdft_fdx_t));
  for (size_t i = 0, j = 0; i < size; i += hopsize, ++j)
  {
    sdft_sdft_n(sdft, hopsize, input + i, buffer);
    sdft_isdft_n(sdft, hopsize, buffer, output + i);
    memcpy(dfts + j * dftsize, buffer, dftsize * sizeof(sdft_fdx_t));
  }
  wav_write_mono_f32(argv[6], output, size, sr);
  FILE* f = fopen(argv[7], "wb");
  if (f) { fwrite(dfts, sizeof(sdft_fdx_t), (size / hopsize) * dftsize, f); fclose(f); }

  free(dfts); free(buffer); free(output); free(input);
  sdft_free(sdft);
  return 0;
}
