/*
 * multichannel_rccl.c -- BASELINE config 5 from a C host: independent channels sharded over the
 * GPUs of one node, one batched plan per GPU, no data-path collective.  RCCL (over xGMI) is used
 * only as a barrier that brackets the timed region: a 1-element ncclAllReduce in a group across
 * all devices (single process, ncclCommInitAll), as SURVEY.md section 8e describes.
 *
 *   multichannel_rccl [channels_per_gpu=64] [n=48000] [dftsize=1024] [steps=5] [gpus=all]
 *
 * Hosts without RCCL build with -DSDFT_NO_RCCL: the start/stop line then is a plain loop of
 * hipStreamSynchronize over the devices (one host thread owns every stream, so nothing else is
 * needed; RCCL only adds that the GPUs also rendezvous among themselves).
 *
 * One host thread drives every GPU: plans run in async mode on their own streams, so the calls
 * return after enqueueing and all devices work concurrently.  Prints the aggregate Msamples/s and,
 * for verification, a checksum of the synthesis of channel 0 on device 0.
 */

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#if !defined(SDFT_NO_RCCL)
#include <rccl/rccl.h>
#endif

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define SDFT_NO_COMPLEX_H
#include <sdft/sdft.h>

#define MAXDEV 16
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#if !defined(SDFT_NO_RCCL)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#endif

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* channel c of C: linear sweep 0 -> (sr/2)(1 - c/2C), start phase 2*pi*c/C */
static void sweep(float* x, size_t n, size_t c, size_t C)
{
  const double sr = 48000.0, fend = 0.5 * sr * (1.0 - (double)c / (2.0 * C));
  double phi = 2.0 * 3.14159265358979323846 * (double)c / (double)C;
  for (size_t i = 0; i < n; ++i)
  {
    phi += 2.0 * 3.14159265358979323846 * ((double)i / n * fend) / sr;
    x[i] = (float)sin(phi);
  }
}

int main(int argc, char** argv)
{
  const size_t per_gpu = argc > 1 ? strtoul(argv[1], NULL, 10) : 64;
  const size_t n = argc > 2 ? strtoul(argv[2], NULL, 10) : 48000;
  const size_t m = argc > 3 ? strtoul(argv[3], NULL, 10) : 1024;
  const int steps = argc > 4 ? atoi(argv[4]) : 5;
  int ndev = sdft_hip_device_count();
  if (argc > 5 && atoi(argv[5]) > 0 && atoi(argv[5]) < ndev) ndev = atoi(argv[5]);
  if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 1; }
  if (ndev > MAXDEV) ndev = MAXDEV;
  const size_t channels = per_gpu * (size_t)ndev;

  sdft_t* plan[MAXDEV]; hipStream_t stream[MAXDEV];
#if !defined(SDFT_NO_RCCL)
  ncclComm_t comm[MAXDEV];
#endif
  float *x[MAXDEV], *flag[MAXDEV]; sdft_fdx_t* dfts[MAXDEV]; int placed[MAXDEV];
  int devs[MAXDEV];
  float* host = (float*)malloc(per_gpu * n * sizeof(float));

  for (int d = 0; d < ndev; ++d)
  {
    devs[d] = d;
    CHECK_HIP(hipSetDevice(d));
    CHECK_HIP(hipStreamCreate(&stream[d]));
    plan[d] = sdft_hip_alloc_batch(m, sdft_window_hann, 1.0, per_gpu);
    if (!plan[d]) { fprintf(stderr, "device %d: %s\n", d, sdft_hip_last_error()); return 1; }
    sdft_hip_set_stream(plan[d], stream[d]);
    sdft_hip_set_option(plan[d], "async", 1);
    CHECK_HIP(hipMalloc((void**)&x[d], per_gpu * n * sizeof(float)));
    /* the matrix: placed by the library inside matrix + 64 GiB (the kind of device memory decides how fast it can be written: sdft_hip.h);
       a GPU without that much free memory takes a plain allocation */
    {
      const size_t bytes = per_gpu * n * m * sizeof(sdft_fdx_t);
      double gbs = 0.0;
      dfts[d] = (sdft_fdx_t*)sdft_hip_malloc_matrix_in_arena(bytes, bytes + ((size_t)64 << 30), &gbs);
      placed[d] = dfts[d] != NULL;
      if (!placed[d]) { sdft_hip_clear_error(); CHECK_HIP(hipMalloc((void**)&dfts[d], bytes)); }
      else fprintf(stderr, "device %d: matrix placed, store-only probe %.0f GB/s\n", d, gbs);
    }
    CHECK_HIP(hipMalloc((void**)&flag[d], sizeof(float)));
    CHECK_HIP(hipMemset(flag[d], 0, sizeof(float)));
    for (size_t c = 0; c < per_gpu; ++c) sweep(host + c * n, n, (size_t)d * per_gpu + c, channels);
    CHECK_HIP(hipMemcpy(x[d], host, per_gpu * n * sizeof(float), hipMemcpyHostToDevice));
  }
#if !defined(SDFT_NO_RCCL)
  CHECK_NCCL(ncclCommInitAll(comm, ndev, devs));

#define BARRIER() do {                                                                            \
    CHECK_NCCL(ncclGroupStart());                                                                 \
    for (int d_ = 0; d_ < ndev; ++d_)                                                             \
      CHECK_NCCL(ncclAllReduce(flag[d_], flag[d_], 1, ncclFloat, ncclSum, comm[d_], stream[d_])); \
    CHECK_NCCL(ncclGroupEnd());                                                                   \
    for (int d_ = 0; d_ < ndev; ++d_) { CHECK_HIP(hipSetDevice(d_)); CHECK_HIP(hipStreamSynchronize(stream[d_])); } \
  } while (0)
#else
  (void)devs;
#define BARRIER() do {                                                                            \
    for (int d_ = 0; d_ < ndev; ++d_) { CHECK_HIP(hipSetDevice(d_)); CHECK_HIP(hipStreamSynchronize(stream[d_])); } \
  } while (0)
#endif

  for (int d = 0; d < ndev; ++d) { CHECK_HIP(hipSetDevice(d)); sdft_sdft_n(plan[d], n, x[d], dfts[d]); }   /* warm-up */
  BARRIER();
  const double t0 = now();
  for (int s = 0; s < steps; ++s)
    for (int d = 0; d < ndev; ++d) { CHECK_HIP(hipSetDevice(d)); sdft_sdft_n(plan[d], n, x[d], dfts[d]); }
  BARRIER();
  const double dt = now() - t0;
  if (sdft_hip_last_error()) { fprintf(stderr, "error: %s\n", sdft_hip_last_error()); return 1; }

  /* verification aid: synthesise channel 0 of device 0 from the last matrix */
  CHECK_HIP(hipSetDevice(0));
  sdft_t* one = sdft_alloc_custom(m, sdft_window_hann, 1.0);
  float* y = (float*)malloc(n * sizeof(float));
  sdft_isdft_n(one, n, dfts[0], y);          /* device matrix in, host samples out */
  double checksum = 0;
  for (size_t i = 0; i < n; ++i) checksum += (double)y[i] * (double)((i % 7) + 1);
  sdft_free(one);

#if defined(SDFT_NO_RCCL)
  printf("(no RCCL: host-side start line)  ");
#endif
  printf("gpus=%d channels=%zu n=%zu dftsize=%zu steps=%d  %.3f ms/step  %.1f Msamples/s aggregate  checksum=%.9e\n",
         ndev, channels, n, m, steps, dt / steps * 1e3, (double)channels * n * steps / dt / 1e6, checksum);

  for (int d = 0; d < ndev; ++d)
  {
    CHECK_HIP(hipSetDevice(d));
    sdft_free(plan[d]);
#if !defined(SDFT_NO_RCCL)
    ncclCommDestroy(comm[d]);
#endif
    (void)hipFree(x[d]); (void)hipFree(flag[d]);
    if (placed[d]) (void)sdft_hip_free_matrix(dfts[d]); else (void)hipFree(dfts[d]);
    (void)hipStreamDestroy(stream[d]);
  }
  free(y); free(host);
  return 0;
}
