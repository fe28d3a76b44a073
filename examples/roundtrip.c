/*
 * roundtrip.c -- the reference's README usage (README.md:30-50) against libsdft_hip.so: analyse one
 * second of a 1 kHz sine at 48 kHz, synthesise it back, report the round-trip SNR after compensating
 * the (dftsize-1)*latency samples of delay (reference python/examples/latency.py:30).
 *
 *   make -C examples && ./examples/roundtrip
 */

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <sdft/sdft.h>

int main(void)
{
  const size_t n = 48000, m = 1024;
  float* x = (float*)malloc(n * sizeof(float));
  float* y = (float*)malloc(n * sizeof(float));
  double complex* dfts = (double complex*)malloc(n * m * sizeof(double complex));   /* (n, m) matrix, host memory */
  for (size_t i = 0; i < n; ++i) x[i] = (float)sin(2.0 * 3.14159265358979323846 * 1000.0 * (double)i / 48000.0);

  sdft_t* sdft = sdft_alloc_custom(m, sdft_window_hann, 1);
  if (!sdft) { fprintf(stderr, "no plan: %s\n", sdft_hip_last_error()); return 1; }

  sdft_sdft_n(sdft, n, x, dfts);      /* analysis  */
  sdft_isdft_n(sdft, n, dfts, y);     /* synthesis */

  const size_t lag = m - 1;
  double sig = 0, err = 0;
  for (size_t i = 2 * m; i + lag < n; ++i)
  {
    const double e = (double)y[i + lag] - (double)x[i];
    sig += (double)x[i] * x[i]; err += e * e;
  }
  printf("|X[43]| at t=%zu: %.6f  (1 kHz = bin 42.7 of 1024 bins spanning 0..24 kHz)\n", n - 1, cabs(dfts[(n - 1) * m + 43]));
  printf("round-trip SNR: %.1f dB over %zu samples, delay %zu samples\n", 10.0 * log10(sig / err), n, lag);

  sdft_free(sdft);
  free(dfts); free(y); free(x);
  return 0;
}
