/*
 * sdft/sdft.h -- drop-in declarations for the MI355X-native Sliding DFT engine (libsdft_hip.so).
 *
 * A C host that was written against jurihock/sdft's header-only C library keeps its source:
 *
 *     #include <sdft/sdft.h>
 *     sdft_t* sdft = sdft_alloc(1024);
 *     sdft_sdft_n(sdft, n, x, dfts);      // analysis  -> (n, 1024) complex matrix
 *     sdft_isdft_n(sdft, n, dfts, y);     // synthesis -> n samples
 *     sdft_free(sdft);
 *
 * but links `-lsdft_hip -lamdhip64` instead of compiling the algorithm into its own translation
 * unit.  The functions below have the names, argument meaning and (absence of) error behaviour of
 * the reference (file:line citations are into the reference's c/src/sdft/sdft.h); they run as
 * hand-written HIP kernels on gfx950.  This header contains declarations only.
 *
 * Scalar types are selected exactly like in the reference (sdft.h:21-37):
 *
 *     #define SDFT_TD_FLOAT   (default)   | SDFT_TD_DOUBLE
 *     #define SDFT_FD_DOUBLE  (default)   | SDFT_FD_FLOAT
 *     #define SDFT_NO_COMPLEX_H           -> struct { r, i } complex instead of <complex.h>
 *
 * C has no overloading, so the public names are mapped onto one of four exported symbol sets
 * (suffix _f32f64, _f32f32, _f64f64, _f64f32 = <time domain><frequency domain>).
 * SDFT_TD_LONG_DOUBLE / SDFT_FD_LONG_DOUBLE have no GPU representation and are rejected.
 *
 * Pointers: `samples` / `dft(s)` may be host pointers (the library stages them over PCIe and
 * returns when the output is complete, like the reference) or device pointers (hipMalloc; no
 * copies -- this is the path the throughput figures are quoted on).  See sdft_hip.h for the
 * additions (batched channels, streams, error text).
 */

#ifndef SDFT_HIP_SDFT_H
#define SDFT_HIP_SDFT_H

#include <stddef.h>

#if defined(SDFT_TD_LONG_DOUBLE) || defined(SDFT_FD_LONG_DOUBLE)
#error "sdft-hip: long double has no GPU representation; use SDFT_TD_DOUBLE / SDFT_FD_DOUBLE"
#endif

#if !defined(SDFT_NO_COMPLEX_H) && !defined(__cplusplus)
#include <complex.h>
#endif

#if defined(__cplusplus)
extern "C" {
#endif

/* ---- types (reference sdft.h:78-135) ----------------------------------------------------- */

typedef size_t sdft_size_t;
typedef float sdft_float_t;
typedef double sdft_double_t;

#if defined(SDFT_NO_COMPLEX_H) || defined(__cplusplus)
  /* two consecutive scalars: layout-compatible with C99 complex and std::complex */
  struct sdft_float_complex { float r, i; };
  struct sdft_double_complex { double r, i; };
  typedef struct sdft_float_complex sdft_float_complex_t;
  typedef struct sdft_double_complex sdft_double_complex_t;
#else
  typedef float complex sdft_float_complex_t;
  typedef double complex sdft_double_complex_t;
#endif

#if defined(SDFT_TD_DOUBLE)
  typedef sdft_double_t sdft_td_t;
  #define SDFT_HIP_TD_TAG f64
#else
  #if !defined(SDFT_TD_FLOAT)
  #define SDFT_TD_FLOAT
  #endif
  typedef sdft_float_t sdft_td_t;
  #define SDFT_HIP_TD_TAG f32
#endif

#if defined(SDFT_FD_FLOAT)
  typedef sdft_float_t sdft_fd_t;
  typedef sdft_float_complex_t sdft_fdx_t;
  #define SDFT_HIP_FD_TAG f32
#else
  #if !defined(SDFT_FD_DOUBLE)
  #define SDFT_FD_DOUBLE
  #endif
  typedef sdft_double_t sdft_fd_t;
  typedef sdft_double_complex_t sdft_fdx_t;
  #define SDFT_HIP_FD_TAG f64
#endif

enum sdft_window            /* reference sdft.h:127-133 */
{
  sdft_window_boxcar,
  sdft_window_hann,
  sdft_window_hamming,
  sdft_window_blackman
};
typedef enum sdft_window sdft_window_t;

/* The plan is opaque here: it owns device buffers (tables, delay line, accumulators, a stream).
   The reference exposes the struct (sdft.h:145-182) but its documented usage only passes the
   pointer around. */
typedef struct sdft_plan sdft_t;

/* ---- symbol selection --------------------------------------------------------------------
   The C identifiers stay exactly the reference's; only the linker symbol carries the type
   suffix (GNU asm label; gcc, clang and hipcc all support it).  No macro touches the names, so
   `enum sdft_window` and the function `sdft_window` keep coexisting like in the reference. */

#define SDFT_HIP_STR_(x) #x
#define SDFT_HIP_STR(x) SDFT_HIP_STR_(x)
#if defined(__GNUC__) || defined(__clang__)
#define SDFT_HIP_SYMBOL(name) \
  __asm__("sdft_hip_" #name "_" SDFT_HIP_STR(SDFT_HIP_TD_TAG) SDFT_HIP_STR(SDFT_HIP_FD_TAG))
#else
#error "sdft-hip: this header needs GNU asm labels (gcc, clang, hipcc)"
#endif

/* ---- the drop-in surface ----------------------------------------------------------------- */

/* Halo cells on either side of a spectrum for the window convolution (replaces the constant
   defined at sdft.h:184; here it lives in the library, untyped). */
extern const sdft_size_t sdft_convolution_kernel_size;

/* Allocates a plan with `dftsize` bins, Hann window, latency 1.   (replaces sdft.h:457)
   Returns NULL if the GPU cannot be set up (sdft_hip_last_error() tells why). */
sdft_t* sdft_alloc(const sdft_size_t dftsize) SDFT_HIP_SYMBOL(alloc);

/* Allocates a plan: analysis window (boxcar, hann, hamming, blackman) and synthesis latency
   factor in (0, 1].                                                  (replaces sdft.h:413) */
sdft_t* sdft_alloc_custom(const sdft_size_t dftsize, const sdft_window_t window, const sdft_double_t latency) SDFT_HIP_SYMBOL(alloc_custom);

/* Releases the plan; NULL is ignored.                                (replaces sdft.h:466) */
void sdft_free(sdft_t* sdft) SDFT_HIP_SYMBOL(free);

/* Back to the initial state: delay line and accumulators zero, cursor 0.   (replaces sdft.h:517) */
void sdft_reset(sdft_t* sdft) SDFT_HIP_SYMBOL(reset);

/* Getters; NULL plan -> 0 / boxcar / 0.                  (replace sdft.h:535, :543, :551) */
sdft_size_t sdft_size(const sdft_t* sdft) SDFT_HIP_SYMBOL(size);
sdft_window_t sdft_window(const sdft_t* sdft) SDFT_HIP_SYMBOL(window);
sdft_double_t sdft_latency(const sdft_t* sdft) SDFT_HIP_SYMBOL(latency);

/* Analyses one sample into an already allocated DFT vector of shape (dftsize).
                                                                      (replaces sdft.h:562) */
void sdft_sdft(sdft_t* sdft, const sdft_td_t sample, sdft_fdx_t* const dft) SDFT_HIP_SYMBOL(sdft);

/* Analyses `nsamples` samples into an already allocated row-major DFT matrix of shape
   (nsamples, dftsize).  State persists across calls.                 (replaces sdft.h:607) */
void sdft_sdft_n(sdft_t* sdft, const sdft_size_t nsamples, const sdft_td_t* samples, sdft_fdx_t* const dfts) SDFT_HIP_SYMBOL(sdft_n);

/* Same with an array of `nsamples` row pointers, each of shape (dftsize).
                                                                      (replaces sdft.h:622) */
void sdft_sdft_nd(sdft_t* sdft, const sdft_size_t nsamples, const sdft_td_t* samples, sdft_fdx_t** const dfts) SDFT_HIP_SYMBOL(sdft_nd);

/* Synthesises one sample from a DFT vector.                          (replaces sdft.h:635) */
sdft_td_t sdft_isdft(sdft_t* sdft, const sdft_fdx_t* dft) SDFT_HIP_SYMBOL(isdft);

/* Synthesises `nsamples` samples from a DFT matrix (nsamples, dftsize).  Does not modify the
   plan.                                                              (replaces sdft.h:666) */
void sdft_isdft_n(sdft_t* sdft, const sdft_size_t nsamples, const sdft_fdx_t* dfts, sdft_td_t* const samples) SDFT_HIP_SYMBOL(isdft_n);

/* Same with an array of row pointers.                                (replaces sdft.h:681) */
void sdft_isdft_nd(sdft_t* sdft, const sdft_size_t nsamples, const sdft_fdx_t** dfts, sdft_td_t* const samples) SDFT_HIP_SYMBOL(isdft_nd);

#if defined(__cplusplus)
}
#endif

#include <sdft/sdft_hip.h>

#endif /* SDFT_HIP_SDFT_H */
