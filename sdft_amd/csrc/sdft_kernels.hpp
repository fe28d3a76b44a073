// sdft_kernels.hpp -- hand-written HIP kernels for gfx950 (CDNA4) implementing the modulated
// Sliding DFT hot path.  Citations are into /root/reference/c/src/sdft/sdft.h.
//
// Kernels:
//   delta_kernel          K0   differences x[t] - x[t-2N] in TD precision + delay line update
//   chunk_fft_kernel      K1a  per-chunk partial sums of the accumulator as an in-LDS 2N-point FFT
//   chunk_fft_mixed_kernel K1a the same for 2N = product of 2, 3, 4, 5 (Stockham, two LDS buffers)
//   chunk_sum_kernel      K1a  the same sums directly (any other N)
//   carry_scan_kernel     K1b  exclusive scan over chunks -> carry-in of every time chunk
//   carry_exact_kernel    K1a' serial pass with the reference's rounding sequence (FD float)
//   fid_seed_kernel, carry_ring_kernel, carry_chain_kernel
//                         K1a' exact carries, chain form: rotations regenerated from a seed table by producer
//                              waves, one dependent addition per step on a consumer wave (LDS ring / rounds)
//   forward_rows_kernel   K1   one workgroup per (chunk, row): LDS edge exchange, lockstep row stores;
//                              SYN != 0: fused synthesis (terms in LDS, tree sum or the reference's order)
//   forward_kernel        K1   independent waves with halo lanes (any N, row-pointer outputs)
//   forward_hop_kernel    K1h  calls of one time chunk: differences + analysis in one launch
//   inverse_exact_kernel  K2   synthesis, bins summed in the reference's order (LDS transpose)
//   inverse_row_kernel    K2   the same for few rows: one wave per row
//   inverse_kernel        K2   synthesis, wave-parallel tree sum (measurement alternative)
//   fold_coeff_kernel, process_rows_kernel, process_hop_kernel
//                         K3   fused analysis -> operation -> synthesis, folded form: one coefficient per bin
//                              instead of window + operation + synthesis term; long calls / calls of one chunk
//
// Common decomposition: lanes <-> frequency bins (one complex bin per lane for 16-byte bins, two
// adjacent bins per lane for 8-byte bins, so a lane always stores 16 B), the sample loop is carried
// inside the kernel, the grid is bins x time chunks x channels.  The per-sample input difference
// is wave-uniform and arrives over the scalar unit (s_load through the constant address space);
// twiddles and state live in VGPRs for a whole chunk; window neighbours come from DPP whole-wave
// shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1).
//
// Arithmetic follows the reference's struct-complex formulas operation by operation and the
// translation units are compiled with -ffp-contract=off: given the same carry-in a wave
// reproduces the reference bit for bit (the FUSED instantiations of forward_rows_kernel and
// process_rows_kernel are the deliberate exceptions, used only where the carry-in already differs
// in summation order).

#pragma once

// (this file is also compiled at run time, by hiprtc, for sdft_hip_process_n with sdft_hip_op_expr: the library carries its text, and the
// run-time compiler brings its own HIP declarations)
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#else
typedef unsigned long uintptr_t;
#endif

#pragma clang fp contract(off)

namespace sdfthip {

#define SDFT_HD __host__ __device__ __forceinline__
#define SDFT_D  __device__ __forceinline__

template <typename T> struct cx { T re, im; };

// sdft.h:265-331 (SDFT_NO_COMPLEX_H formulas)
template <typename T> SDFT_HD cx<T> cmake(T re, T im) { cx<T> z; z.re = re; z.im = im; return z; }
template <typename T> SDFT_HD cx<T> cadd(cx<T> a, cx<T> b) { return cmake<T>(a.re + b.re, a.im + b.im); }
template <typename T> SDFT_HD cx<T> csub(cx<T> a, cx<T> b) { return cmake<T>(a.re - b.re, a.im - b.im); }
template <typename T> SDFT_HD cx<T> cmul(cx<T> a, cx<T> b) { return cmake<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
template <typename T> SDFT_HD cx<T> cscale(cx<T> a, T s) { return cmake<T>(a.re * s, a.im * s); }
template <typename T> SDFT_HD cx<T> cconj(cx<T> a) { return cmake<T>(a.re, -a.im); }

enum : int { WIN_BOXCAR = 0, WIN_HANN = 1, WIN_HAMMING = 2, WIN_BLACKMAN = 3 };   // sdft.h:127-133

template <int WIN> struct win_halo { static constexpr int value = (WIN == WIN_BLACKMAN) ? 2 : (WIN == WIN_BOXCAR ? 0 : 1); };

// Wave-uniform read-only streams (the per-sample differences) are read through the constant
// address space so that the compiler keeps them on the scalar unit (s_load via the scalar cache)
// even though the kernel also stores to global memory.  Legal because no kernel writes a buffer
// it reads this way.
#define SDFT_CONSTANT __attribute__((address_space(4)))
template <typename T> SDFT_D const SDFT_CONSTANT T* as_uniform(const T* p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const SDFT_CONSTANT T*)p;
#pragma clang diagnostic pop
}

constexpr int kWave = 64;
constexpr int kBlock = 256;              // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kGroup = 8;                // samples per scalar-load burst in the time loop
constexpr int kHopMax = 512;             // calls of one time chunk are shorter than this (Plan::choose_chunks)

// ------------------------------------------------------------------------------------------
// cross-lane neighbour fetch: lane i <- lane i-1 (from_below) / lane i+1 (from_above).
// gfx950 is a GFX9-family ISA and still has the whole-wave DPP shifts.
// ------------------------------------------------------------------------------------------
#if defined(SDFT_NEIGHBOUR_BPERMUTE)
SDFT_D int lane_from_below(int v) { return __shfl_up(v, 1, 64); }
SDFT_D int lane_from_above(int v) { return __shfl_down(v, 1, 64); }
#else
SDFT_D int lane_from_below(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); }
SDFT_D int lane_from_above(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false); }
#endif

// Variants with an explicit fill value: a whole-wave shift leaves lane 0 (from_below) / lane 63
// (from_above) without a source lane; with bound_ctrl off that lane keeps `old`.  The row-group
// kernel passes the neighbouring wave's edge bin there, so crossing a wave boundary costs no
// select.
SDFT_D int lane_from_below_fill(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false); }
SDFT_D int lane_from_above_fill(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false); }
SDFT_D float from_below_fill(float old, float v) { return __int_as_float(lane_from_below_fill(__float_as_int(old), __float_as_int(v))); }
SDFT_D float from_above_fill(float old, float v) { return __int_as_float(lane_from_above_fill(__float_as_int(old), __float_as_int(v))); }
SDFT_D double from_below_fill(double old, double v)
{
  const int lo = lane_from_below_fill(__double2loint(old), __double2loint(v));
  const int hi = lane_from_below_fill(__double2hiint(old), __double2hiint(v));
  return __hiloint2double(hi, lo);
}
SDFT_D double from_above_fill(double old, double v)
{
  const int lo = lane_from_above_fill(__double2loint(old), __double2loint(v));
  const int hi = lane_from_above_fill(__double2hiint(old), __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// bound_ctrl forms: the lane without a source lane receives 0 and no `old` value has to be set up
// (saves one v_mov per shifted dword); for callers that never use what lane 0 / lane 63 receive
SDFT_D int lane_from_below_z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
SDFT_D int lane_from_above_z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
SDFT_D float from_below_z(float v) { return __int_as_float(lane_from_below_z(__float_as_int(v))); }
SDFT_D float from_above_z(float v) { return __int_as_float(lane_from_above_z(__float_as_int(v))); }
SDFT_D double from_below_z(double v)
{
  const int lo = lane_from_below_z(__double2loint(v)), hi = lane_from_below_z(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
SDFT_D double from_above_z(double v)
{
  const int lo = lane_from_above_z(__double2loint(v)), hi = lane_from_above_z(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

SDFT_D float from_below(float v) { return __int_as_float(lane_from_below(__float_as_int(v))); }
SDFT_D float from_above(float v) { return __int_as_float(lane_from_above(__float_as_int(v))); }
SDFT_D double from_below(double v)
{
  const int lo = lane_from_below(__double2loint(v)), hi = lane_from_below(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
SDFT_D double from_above(double v)
{
  const int lo = lane_from_above(__double2loint(v)), hi = lane_from_above(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <typename T> SDFT_D cx<T> from_below_fill(cx<T> o, cx<T> z) { return cmake<T>(from_below_fill(o.re, z.re), from_below_fill(o.im, z.im)); }
template <typename T> SDFT_D cx<T> from_above_fill(cx<T> o, cx<T> z) { return cmake<T>(from_above_fill(o.re, z.re), from_above_fill(o.im, z.im)); }
template <typename T> SDFT_D cx<T> from_below_z(cx<T> z) { return cmake<T>(from_below_z(z.re), from_below_z(z.im)); }
template <typename T> SDFT_D cx<T> from_above_z(cx<T> z) { return cmake<T>(from_above_z(z.re), from_above_z(z.im)); }
template <typename T> SDFT_D cx<T> from_below(cx<T> z) { return cmake<T>(from_below(z.re), from_below(z.im)); }
template <typename T> SDFT_D cx<T> from_above(cx<T> z) { return cmake<T>(from_above(z.re), from_above(z.im)); }

// conjugation by a lane-constant mask (0 or the sign bit): one v_xor_b32, no select
SDFT_D float flip_sign(float v, unsigned mask) { return __int_as_float(__float_as_int(v) ^ (int)mask); }
SDFT_D double flip_sign(double v, unsigned mask) { return __hiloint2double(__double2hiint(v) ^ (int)mask, __double2loint(v)); }

// ------------------------------------------------------------------------------------------
// index reflection for the halo (sdft.h:589-595): X[-i] = conj X[i], X[N-1+i] = conj X[N-1-i],
// iterated for tiny N.  Returns the source bin, sets `flip` when an odd number of conjugations
// applies.  (N == 1 is special: the reference's halo cells stay zero -- handled by the caller.)
// ------------------------------------------------------------------------------------------
SDFT_HD long reflect_bin(long k, long nbins, bool& flip)
{
  flip = false;
  if (nbins <= 1) return 0;                 // N == 1: reflections about bin 0 never settle; caller zeroes the halo
  while (k < 0 || k > nbins - 1)
  {
    k = (k < 0) ? -k : 2 * (nbins - 1) - k;
    flip = !flip;
  }
  return k;
}

// ------------------------------------------------------------------------------------------
// K0  delta + delay line  (sdft.h:186-191, :564)
//   delta[t] = (FD)( x[t] - x[t-2N] ), the subtraction in TD precision.
//   hist is the delay line kept in time order (oldest first); a second buffer receives the
//   last 2N samples of (hist ++ x) for the next call.
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD>
__global__ __launch_bounds__(kBlock) void delta_kernel(const TD* __restrict__ x, size_t x_stride,
                                                       const TD* __restrict__ hist_in, TD* __restrict__ hist_out,
                                                       FD* __restrict__ delta, size_t n, size_t span /*2N*/,
                                                       const cx<FD>* __restrict__ acc_state, const cx<FD>* __restrict__ fid_state,
                                                       cx<FD>* __restrict__ carry0, cx<FD>* __restrict__ seed0,
                                                       unsigned blocks_per_channel)
{
  // channels ride on grid.x (grid.y/z stop at 65535)
  const size_t ch = blockIdx.x / blocks_per_channel;
  const size_t i = (size_t)(blockIdx.x % blocks_per_channel) * kBlock + threadIdx.x;
  const TD* xs = x + ch * x_stride;
  const TD* hi = hist_in + ch * span;
  if (i < n)
  {
    const TD cur = xs[i];
    const TD old = (i < span) ? hi[i] : xs[i - span];
    const TD d = cur - old;                       // TD precision
    delta[ch * n + i] = (FD)d;
  }
  if (i < span)
  {
    // element i of the new history = element (n + i) of the concatenation hist ++ x, minus span
    const size_t j = n + i;
    hist_out[ch * span + i] = (j >= span) ? xs[j - span] : hi[j];
  }
  // single-chunk calls: the stream state is the carry; copied here (instead of two extra copy
  // launches) because halo lanes / mirror publishers read bins whose owner may already have
  // written the new state
  if (carry0 && i < span / 2)
  {
    carry0[ch * (span / 2) + i] = acc_state[ch * (span / 2) + i];
    seed0[ch * (span / 2) + i] = fid_state[ch * (span / 2) + i];
  }
}

// ------------------------------------------------------------------------------------------
// shared pieces of the recurrence
// ------------------------------------------------------------------------------------------
template <typename FD> SDFT_D FD wave_sum_f(FD v)
{
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename FD> struct BinState { cx<FD> acc, fid, tw; };

// Completion word for synchronous short calls.  A kernel's end reaches the host ~6 us later than a store to pinned
// host memory does (scripts/launch_latency.hip): the launch's last workgroup -- found by an agent-scope ticket that
// also publishes the workgroup's stores -- sets `flag` to `seq`, and the host polls that word instead of the stream.
struct DoneSignal
{
  unsigned* flag;             // pinned host memory, or nullptr: no signal wanted
  unsigned* count;            // device word, zero between launches
  unsigned seq, total;        // value to publish, workgroups that must have finished
};
// call with the workgroup's stores issued; one lane of the workgroup's last wave
SDFT_D void signal_done(const DoneSignal& d)
{
  if (!d.flag) return;
  const unsigned finished = __hip_atomic_fetch_add(d.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (finished + 1u == d.total)
  {
    __hip_atomic_store(d.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// normal step (sdft.h:583-585) -- returns the demodulated bin
template <typename FD> SDFT_D cx<FD> step_normal(BinState<FD>& s, FD delta)
{
  s.acc = cadd(s.acc, cscale(s.fid, delta));
  s.fid = cmul(s.fid, s.tw);
  return cmul(s.acc, cconj(s.fid));
}
// roll-over step (sdft.h:572-574)
template <typename FD> SDFT_D cx<FD> step_wrap(BinState<FD>& s, FD delta)
{
  s.acc = cadd(s.acc, cscale(s.fid, delta));
  s.fid = cmake<FD>((FD)1, (FD)0);
  return s.acc;
}
// Fused-multiply-add forms of the same step, selectable for the chunk-parallel FD double path
// (option "fused"; its carry-in already differs from the serial reference in summation order):
// 16 instead of 24 fp64 operations per bin-sample for a Hann window.  Never used in exact-carry
// mode or for single-chunk calls, which stay bit-identical to the reference.
template <typename FD> SDFT_D cx<FD> step_normal_fused(BinState<FD>& s, FD delta)
{
  s.acc.re = __builtin_fma(s.fid.re, delta, s.acc.re);
  s.acc.im = __builtin_fma(s.fid.im, delta, s.acc.im);
  const FD nr = __builtin_fma(s.fid.re, s.tw.re, -(s.fid.im * s.tw.im));
  const FD ni = __builtin_fma(s.fid.re, s.tw.im, s.fid.im * s.tw.re);
  s.fid.re = nr; s.fid.im = ni;
  return cmake<FD>(__builtin_fma(s.acc.re, nr, s.acc.im * ni), __builtin_fma(s.acc.im, nr, -(s.acc.re * ni)));
}
template <typename FD> SDFT_D cx<FD> step_wrap_fused(BinState<FD>& s, FD delta)
{
  s.acc.re = __builtin_fma(s.fid.re, delta, s.acc.re);
  s.acc.im = __builtin_fma(s.fid.im, delta, s.acc.im);
  s.fid = cmake<FD>((FD)1, (FD)0);
  return s.acc;
}

// recurrence without the demodulation (carry passes)
template <typename FD> SDFT_D void advance_normal(BinState<FD>& s, FD delta)
{
  s.acc = cadd(s.acc, cscale(s.fid, delta));
  s.fid = cmul(s.fid, s.tw);
}
template <typename FD> SDFT_D void advance_wrap(BinState<FD>& s, FD delta)
{
  s.acc = cadd(s.acc, cscale(s.fid, delta));
  s.fid = cmake<FD>((FD)1, (FD)0);
}

// ------------------------------------------------------------------------------------------
// K1a (fast carry, FD double): per (chunk, bin) partial sums of delta*fid over one chunk,
// written to carry[ch][chunk+1][k]; K1b turns them into carry-ins by an exclusive scan over
// chunks.  fid is seeded from the table W[j] = exp(-i*pi*j/N), j = k*cursor mod 2N, and then
// advanced exactly like the main kernel does, so both see the same rotation sequence.
// ------------------------------------------------------------------------------------------
// the same for a workgroup of several waves: every wave waits for its own stores, the workgroup meets, one lane reports
SDFT_D void signal_done_workgroup(const DoneSignal& d)
{
  if (!d.flag) return;                                      // workgroup-uniform
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) signal_done(d);
}

template <typename FD> struct CarryArgs
{
  const FD* delta;            // [channels][n]
  const cx<FD>* tw;           // [N]
  const cx<FD>* wtab;         // [2N]
  cx<FD>* carry;              // [channels][chunks][N]
  cx<FD>* seed;               // [channels][chunks][N]   (exact mode only)
  const cx<FD>* acc_state;    // [channels][N]  state at the first chunk of this launch
  const cx<FD>* fid_state;    // [channels][N]
  cx<FD>* acc_next;           // [channels][N]  exact pass: state after the last chunk of this launch
  cx<FD>* fid_next;           //                (nullptr when the launch ends with the call's last chunk)
  size_t n;
  unsigned nbins, chunks, chunk_len, cursor0;
  unsigned chunk0, launch_chunks;   // exact pass: this launch covers chunks [chunk0, chunk0 + launch_chunks)
};

// Closed form instead of the rotation recurrence: with W[j] = exp(-i*pi*j/N) (period 2N, so the
// roll-over needs no special case) and the chunk cut into blocks of kSumBlock samples,
//   S = sum_a W[k*(c0 + a*B)] * ( sum_{b<B} delta[a*B + b] * W[k*b] ),
// i.e. 2 FMAs per sample against B lane-constant factors plus one complex multiply-add and one
// rotation per block: ~3 fp64 FMAs per bin-sample instead of 10 operations.  FMAs are fine
// here: this pass only feeds the carry, whose summation order differs from the reference anyway.
constexpr int kSumBlock = 8;

// Differences formed by the carry pass itself (one launch less in front of the forward kernel): when `x` is set the
// FFT kernels below compute delta[t] = (FD)(x[t] - x[t - 2N]) (sdft.h:564, the subtraction in TD precision) for their
// chunk, fold it into LDS AND write it to delta_out for the forward kernel; the workgroup of the call's last chunk
// (which has no partial sum to form) writes its differences and the channel's new delay line.
template <typename TD, typename FD> struct DeltaIn
{
  const TD* x;                // [channels][n], or nullptr: read CarryArgs::delta as before
  size_t x_stride;
  const TD* hist_in;          // [channels][2N] delay line in time order
  TD* hist_out;
  FD* delta_out;              // [channels][n]
};
template <typename TD, typename FD>
SDFT_D FD chunk_delta(const DeltaIn<TD, FD>& di, const TD* xs, const TD* hs, size_t t, size_t span)
{
  const TD cur = xs[t];
  const TD old = (t < span) ? hs[t] : xs[t - span];
  const TD dd = cur - old;                                  // TD precision
  return (FD)dd;
}
template <typename TD, typename FD>
__global__ __launch_bounds__(kBlock) void chunk_sum_kernel(CarryArgs<FD> a, DeltaIn<TD, FD> di)
{
  constexpr int B = kSumBlock;
  // 1-D grid = bin blocks x (chunks - 1) x channels (grid.y/z stop at 65535); with fused differences one more
  // chunk per channel: the last one has no partial sum to form, only its differences and the delay line to write
  const unsigned bin_blocks = (a.nbins + kBlock - 1) / kBlock;
  const unsigned per_ch = di.x ? a.chunks : a.chunks - 1;
  const unsigned bb = blockIdx.x % bin_blocks;
  const unsigned k = bb * kBlock + threadIdx.x;
  const unsigned j = (blockIdx.x / bin_blocks) % per_ch;           // chunk 0 .. chunks-2: all of full length, a multiple of B
  const size_t ch = (blockIdx.x / bin_blocks) / per_ch;
  const unsigned kk = k < a.nbins ? k : a.nbins - 1;
  const unsigned span = 2u * a.nbins;
  const size_t t0 = (size_t)j * a.chunk_len;
  const unsigned c0 = (unsigned)(((size_t)a.cursor0 + t0) % span);
  const SDFT_CONSTANT TD* xs = nullptr;
  const SDFT_CONSTANT TD* hs = nullptr;
  if (di.x)
  {
    const TD* xv = di.x + ch * di.x_stride;
    const TD* hv = di.hist_in + ch * (size_t)span;
    if (bb == 0)
    {
      // the chunk's differences for the forward kernel (one workgroup per chunk writes them)
      FD* dout = di.delta_out + ch * a.n;
      const size_t t1 = (t0 + a.chunk_len < a.n) ? t0 + a.chunk_len : a.n;
      for (size_t t = t0 + threadIdx.x; t < t1; t += kBlock) dout[t] = chunk_delta(di, xv, hv, t, (size_t)span);
      if (j + 1 == a.chunks)
      {
        TD* ho = di.hist_out + ch * (size_t)span;            // element i of the last 2N samples of (hist ++ x)
        for (size_t i = threadIdx.x; i < span; i += kBlock)
        {
          const size_t q = a.n + i;
          ho[i] = (q >= span) ? xv[q - span] : hv[q];
        }
      }
    }
    if (j + 1 == a.chunks) return;
    xs = as_uniform(xv);
    hs = as_uniform(hv);
  }

  cx<FD> w[B];
#pragma unroll
  for (int b = 0; b < B; ++b) w[b] = a.wtab[(size_t)(((unsigned long long)kk * b) % span)];
  cx<FD> rot = a.wtab[(size_t)(((unsigned long long)kk * c0) % span)];
  const cx<FD> rotB = a.wtab[(size_t)(((unsigned long long)kk * B) % span)];
  cx<FD> s = cmake<FD>((FD)0, (FD)0);
  const SDFT_CONSTANT FD* d = as_uniform(a.delta + ch * a.n + t0);

  const unsigned blocks = a.chunk_len / B;
  for (unsigned blk = 0; blk < blocks; ++blk)
  {
    FD dl[B];
    if (di.x)
    {
      // differences from the input and the delay line (sdft.h:564), the subtraction in TD precision
      const size_t tt = t0 + (size_t)blk * B;
      TD cur[B], old[B];
#pragma unroll
      for (int b = 0; b < B; ++b) cur[b] = xs[tt + b];
      if (tt + B <= span)
      {
#pragma unroll
        for (int b = 0; b < B; ++b) old[b] = hs[tt + b];
      }
      else if (tt >= span)
      {
#pragma unroll
        for (int b = 0; b < B; ++b) old[b] = xs[tt - span + b];
      }
      else
      {
#pragma unroll
        for (int b = 0; b < B; ++b) old[b] = (tt + b < span) ? hs[tt + b] : xs[tt + b - span];
      }
#pragma unroll
      for (int b = 0; b < B; ++b) { const TD dd = cur[b] - old[b]; dl[b] = (FD)dd; }
    }
    else
    {
#pragma unroll
      for (int b = 0; b < B; ++b) dl[b] = d[(size_t)blk * B + b];
    }
    FD ire = dl[0], iim = (FD)0;              // w[0] == 1
#pragma unroll
    for (int b = 1; b < B; ++b)
    {
      ire = __builtin_fma(dl[b], w[b].re, ire);
      iim = __builtin_fma(dl[b], w[b].im, iim);
    }
    s.re = __builtin_fma(rot.re, ire, s.re); s.re = __builtin_fma(-rot.im, iim, s.re);
    s.im = __builtin_fma(rot.re, iim, s.im); s.im = __builtin_fma(rot.im, ire, s.im);
    const FD nr = __builtin_fma(rot.re, rotB.re, -(rot.im * rotB.im));
    const FD ni = __builtin_fma(rot.re, rotB.im, rot.im * rotB.re);
    rot.re = nr; rot.im = ni;
  }
  if (k < a.nbins)
    a.carry[(ch * a.chunks + j) * a.nbins + k] = s;
}

// K1a (FFT form, N a power of two): the same partial sums are the first N bins of a 2N-point DFT
// of the chunk -- W[j] = exp(-2*pi*i*j/(2N)) is exactly its twiddle table, and chunks longer
// than 2N fold onto themselves because W has period 2N:
//   S[k] = W[k*c0] * sum_{v<2N} ( sum_q delta[v + 2N*q] ) * W[k*v].
// One workgroup per (chunk, channel): fold the chunk into LDS, radix-2 decimation-in-frequency
// in place (log2(2N) barriers), read bin k from its bit-reversed slot.  O(N log N) per chunk
// instead of O(L*N): 149 us -> ~15 us at n = 1e6, N = 1024.
// fold one chunk into the 2N LDS cells (cell v = sum of the chunk's samples v, v + 2N, ...); returns false for the
// workgroup of the last chunk, which has only differences and the delay line to write
template <typename TD, typename FD>
SDFT_D bool chunk_fold(const CarryArgs<FD>& a, const DeltaIn<TD, FD>& di, cx<FD>* x, unsigned m, unsigned j, size_t ch)
{
  const size_t t0 = (size_t)j * a.chunk_len;
  if (!di.x)
  {
    const FD* d = a.delta + ch * a.n + t0;
    for (unsigned v = threadIdx.x; v < m; v += kBlock)
    {
      FD acc = (FD)0;
      for (size_t u = v; u < a.chunk_len; u += m) acc += d[u];
      x[v] = cmake<FD>(acc, (FD)0);
    }
    return true;
  }
  const size_t span = 2 * (size_t)a.nbins;
  const TD* xs = di.x + ch * di.x_stride;
  const TD* hs = di.hist_in + ch * span;
  FD* dout = di.delta_out + ch * a.n;
  const size_t t1 = (t0 + a.chunk_len < a.n) ? t0 + a.chunk_len : a.n;
  if (j + 1 == a.chunks)
  {
    for (size_t t = t0 + threadIdx.x; t < t1; t += kBlock) dout[t] = chunk_delta(di, xs, hs, t, span);
    TD* ho = di.hist_out + ch * span;                       // element i of the last 2N samples of (hist ++ x)
    for (size_t i = threadIdx.x; i < span; i += kBlock)
    {
      const size_t q = a.n + i;
      ho[i] = (q >= span) ? xs[q - span] : hs[q];
    }
    return false;
  }
  for (unsigned v = threadIdx.x; v < m; v += kBlock)
  {
    FD acc = (FD)0;
    for (size_t u = v; u < a.chunk_len; u += m)
    {
      const FD d = chunk_delta(di, xs, hs, t0 + u, span);
      dout[t0 + u] = d;
      acc += d;
    }
    x[v] = cmake<FD>(acc, (FD)0);
  }
  return true;
}

template <typename TD, typename FD>
__global__ __launch_bounds__(kBlock) void chunk_fft_kernel(CarryArgs<FD> a, unsigned log2m, DeltaIn<TD, FD> di)
{
  extern __shared__ __align__(16) unsigned char fft_lds_raw[];
  cx<FD>* x = reinterpret_cast<cx<FD>*>(fft_lds_raw);
  const unsigned m = 1u << log2m;                        // 2N
  // chunk 0 .. chunks-2 (full length) form partial sums; with fused differences the grid has one more workgroup
  // per channel, for the last chunk's differences and the delay line
  const unsigned per_ch = di.x ? a.chunks : a.chunks - 1;
  const unsigned j = blockIdx.x % per_ch;
  const size_t ch = blockIdx.x / per_ch;
  const size_t t0 = (size_t)j * a.chunk_len;
  const unsigned c0 = (unsigned)(((size_t)a.cursor0 + t0) % m);
  if (!chunk_fold(a, di, x, m, j, ch)) return;
  __syncthreads();
  for (unsigned st = 0; st < log2m; ++st)
  {
    const unsigned half = m >> (st + 1);
    for (unsigned i = threadIdx.x; i < (m >> 1); i += kBlock)
    {
      const unsigned pos = i & (half - 1);
      const unsigned lo = ((i - pos) << 1) + pos, hi = lo + half;
      const cx<FD> p = x[lo], q = x[hi];
      const cx<FD> w = a.wtab[(size_t)pos << st];       // exp(-2*pi*i*pos/(2*half))
      x[lo] = cadd(p, q);
      x[hi] = cmul(csub(p, q), w);
    }
    __syncthreads();
  }
  for (unsigned k = threadIdx.x; k < a.nbins; k += kBlock)
  {
    const unsigned r = __brev(k) >> (32 - log2m);
    const cx<FD> rot = a.wtab[(size_t)(((unsigned long long)k * c0) % m)];
    a.carry[(ch * a.chunks + j) * a.nbins + k] = cmul(x[r], rot);
  }
}

// K1a (mixed-radix FFT form): the same 2N-point DFT for sizes that are not powers of two but
// factor into 2, 3, 4, 5 (the reference's own test size N = 1000: 2N = 4*4*5*5*5).  Stockham
// autosort between two LDS buffers, natural-order output, generic r-point butterflies with all
// roots taken from the plan's table W[j] = exp(-2*pi*i*j/(2N)).
struct RadixList { unsigned char count; unsigned char r[15]; };

template <typename TD, typename FD>
__global__ __launch_bounds__(kBlock) void chunk_fft_mixed_kernel(CarryArgs<FD> a, unsigned m, RadixList rl, DeltaIn<TD, FD> di)
{
  extern __shared__ __align__(16) unsigned char fft_lds_raw2[];
  cx<FD>* x = reinterpret_cast<cx<FD>*>(fft_lds_raw2);
  cx<FD>* y = x + m;
  const unsigned per_ch = di.x ? a.chunks : a.chunks - 1;    // see chunk_fft_kernel
  const unsigned j = blockIdx.x % per_ch;
  const size_t ch = blockIdx.x / per_ch;
  const size_t t0 = (size_t)j * a.chunk_len;
  const unsigned c0 = (unsigned)(((size_t)a.cursor0 + t0) % m);
  if (!chunk_fold(a, di, x, m, j, ch)) return;
  __syncthreads();
  unsigned ns = 1;                                       // product of the radices already applied
  for (unsigned st = 0; st < rl.count; ++st)
  {
    const unsigned r = rl.r[st];
    const unsigned nr = m / r;
    const unsigned tstep = m / (ns * r);                 // table stride of the stage twiddle
    const unsigned rstep = nr;                           // table stride of the r-th roots of unity
    for (unsigned i = threadIdx.x; i < nr; i += kBlock)
    {
      const unsigned k = i % ns;
      cx<FD> v[5];
#pragma unroll
      for (unsigned t = 0; t < 5; ++t)
        if (t < r)
        {
          const cx<FD> in = x[i + t * nr];
          v[t] = t == 0 ? in : cmul(in, a.wtab[(size_t)(((unsigned long long)t * k * tstep) % m)]);
        }
      const unsigned base = (i / ns) * ns * r + k;
#pragma unroll
      for (unsigned q = 0; q < 5; ++q)
        if (q < r)
        {
          cx<FD> o = v[0];
#pragma unroll
          for (unsigned t = 1; t < 5; ++t)
            if (t < r) o = cadd(o, cmul(v[t], a.wtab[(size_t)(((unsigned long long)q * t * rstep) % m)]));
          y[base + q * ns] = o;
        }
    }
    __syncthreads();
    cx<FD>* tmp = x; x = y; y = tmp;
    ns *= r;
  }
  for (unsigned k = threadIdx.x; k < a.nbins; k += kBlock)
  {
    const cx<FD> rot = a.wtab[(size_t)(((unsigned long long)k * c0) % m)];
    a.carry[(ch * a.chunks + j) * a.nbins + k] = cmul(x[k], rot);
  }
}

// K1b: exclusive scan over chunks, in place: carry[j] = acc_state + sum_{i<j} partial[i].
// Two levels: a workgroup owns kScanBins bins; its kScanSlices thread groups each own a
// contiguous slice of the chunks, slice totals are combined through LDS.  16 bins x 64 slices
// (256-byte row segments, 64 workgroups at N = 1024) instead of 64 x 16: four times the
// parallelism for a pass that is pure latency.  (partial[chunks-1] does not exist and is not read.)
constexpr int kScanSlices = 64;
constexpr int kScanBins = 16;

template <typename FD>
__global__ __launch_bounds__(kScanBins * kScanSlices) void carry_scan_kernel(CarryArgs<FD> a)
{
  __shared__ cx<FD> totals[kScanSlices][kScanBins];
  const int bin = threadIdx.x % kScanBins;
  const int slice = threadIdx.x / kScanBins;
  const unsigned bin_blocks = (a.nbins + kScanBins - 1) / kScanBins;
  const unsigned k = (blockIdx.x % bin_blocks) * kScanBins + bin;
  const size_t ch = blockIdx.x / bin_blocks;
  const unsigned kk = k < a.nbins ? k : a.nbins - 1;
  const unsigned per = (a.chunks + kScanSlices - 1) / kScanSlices;
  const unsigned j0 = slice * per;
  const unsigned j1 = (j0 + per < a.chunks) ? j0 + per : a.chunks;
  cx<FD>* col = a.carry + ch * a.chunks * a.nbins + kk;

  cx<FD> sum = cmake<FD>((FD)0, (FD)0);
  for (unsigned j = j0; j < j1 && j + 1 < a.chunks; ++j) sum = cadd(sum, col[(size_t)j * a.nbins]);
  totals[slice][bin] = sum;
  __syncthreads();
  cx<FD> run = a.acc_state[ch * a.nbins + kk];
  for (int s = 0; s < slice; ++s) run = cadd(run, totals[s][bin]);
  if (k >= a.nbins) return;
  for (unsigned j = j0; j < j1; ++j)
  {
    const bool has = (j + 1 < a.chunks);
    const cx<FD> part = has ? col[(size_t)j * a.nbins] : cmake<FD>((FD)0, (FD)0);
    col[(size_t)j * a.nbins] = run;
    run = cadd(run, part);
  }
}

// ------------------------------------------------------------------------------------------
// K1a' (exact carry): time-serial per (channel, bin); reproduces the reference's rounding
// sequence of acc and fid and records both at every chunk start.  Used for FD float, where the
// 1e-4 parity bar is tighter than float's own accumulation error.
// ------------------------------------------------------------------------------------------
// The pass is a serial dependency chain, n steps long, with only N-fold parallelism, so what
// counts is instructions and latency per step on a wave that is alone on its SIMD:
//  * the real and imaginary part of a bin live in a lane pair (even lane: re, odd lane: im):
//      acc += fid * delta                          1 mul + 1 add
//      fid' = fid*T1 + partner(fid)*T2             2 mul + 1 add, partner via DPP quad_perm
//    with T1 = tw.re and T2 = -tw.im (re lane) / +tw.im (im lane): 5 VALU ops per step instead
//    of 10, and exactly the reference's roundings (a + (-b) == a - b, addition commutes);
//  * one wave per workgroup, 32 bins per wave: the N/32 waves spread over as many SIMDs;
//  * the wave-uniform differences are staged through LDS in blocks of kExactBlock samples
//    (coalesced vector load of the next block is in flight while the current one is consumed;
//    LDS broadcasts return in order, so the compiler can wait with counted lgkmcnt).
constexpr int kExactBlock = 512;

SDFT_D int lane_partner(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, false); }
SDFT_D float partner(float v) { return __int_as_float(lane_partner(__float_as_int(v))); }
SDFT_D double partner(double v)
{
  const int lo = lane_partner(__double2loint(v)), hi = lane_partner(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <typename FD>
__global__ __launch_bounds__(kWave) void carry_exact_kernel(CarryArgs<FD> a)
{
  constexpr int EB = kExactBlock;
  constexpr int PER = EB / kWave;                       // samples staged per lane
  __shared__ FD stage[2][EB];
  // This wave is a serial chain that the whole call waits for, and it shares its SIMD with
  // forward-kernel waves of earlier segments: let it win every issue arbitration.
  __builtin_amdgcn_s_setprio(3);

  const int lane = threadIdx.x;
  const int comp = lane & 1;
  const unsigned bin_blocks = (a.nbins + kWave / 2 - 1) / (kWave / 2);
  const unsigned bin = (blockIdx.x % bin_blocks) * (kWave / 2) + (lane >> 1);
  const size_t ch = blockIdx.x / bin_blocks;
  const bool valid = bin < a.nbins;
  const unsigned kk = valid ? bin : a.nbins - 1;
  const unsigned span = 2u * a.nbins, maxc = span - 1u;

  const cx<FD> tw = a.tw[kk];
  const cx<FD> acc0 = a.acc_state[ch * a.nbins + kk];
  const cx<FD> fid0 = a.fid_state[ch * a.nbins + kk];
  const FD T1 = tw.re;
  const FD T2 = comp ? tw.im : -tw.im;
  const FD one = comp ? (FD)0 : (FD)1;
  FD acc = comp ? acc0.im : acc0.re;
  FD f = comp ? fid0.im : fid0.re;

  FD* carry = reinterpret_cast<FD*>(a.carry);
  FD* seed = reinterpret_cast<FD*>(a.seed);
  // this launch: chunks [chunk0, chunk0 + launch_chunks); every chunk is dumped at its start and then
  // run, except the call's very last chunk, which the forward kernel runs itself
  const unsigned jend = a.chunk0 + a.launch_chunks;
  const bool ends_call = (jend == a.chunks);
  const size_t tbase = (size_t)a.chunk0 * a.chunk_len;
  const FD* d = a.delta + ch * a.n + tbase;
  const size_t total = (size_t)(a.launch_chunks - (ends_call ? 1 : 0)) * a.chunk_len;
  auto fetch = [&](size_t base, FD (&r)[PER])
  {
#pragma unroll
    for (int q = 0; q < PER; ++q)
    {
      const size_t i = base + (size_t)lane * PER + q;
      r[q] = (i < total) ? d[i] : (FD)0;
    }
  };
  auto put = [&](int buf, const FD (&r)[PER])
  {
#pragma unroll
    for (int q = 0; q < PER; ++q) stage[buf][lane * PER + q] = r[q];
  };
  auto dump = [&](unsigned j)
  {
    if (valid)
    {
      const size_t o = (((ch * a.chunks + j) * a.nbins) + bin) * 2 + comp;
      carry[o] = acc;
      seed[o] = f;
    }
  };

  unsigned c = (unsigned)(((size_t)a.cursor0 + tbase) % span);
  unsigned j = a.chunk0;
  size_t next_dump = 0;
  FD regs[PER];
  fetch(0, regs);
  put(0, regs);
  for (size_t base = 0; base < total; base += EB)
  {
    const int buf = (int)((base / EB) & 1);
    const bool more = base + EB < total;
    if (more) fetch(base + EB, regs);                    // global loads in flight during the block
    __syncthreads();                                     // single-wave group: orders the LDS writes
    const unsigned m = (total - base < (size_t)EB) ? (unsigned)(total - base) : (unsigned)EB;
    unsigned u = 0;
    while (u < m)
    {
      if (base + u == next_dump) { dump(j); ++j; next_dump += a.chunk_len; }
      unsigned run = m - u;
      if ((size_t)run > next_dump - (base + u)) run = (unsigned)(next_dump - (base + u));
      if (run > maxc - c) run = maxc - c;
      if (run == 0)
      {
        // roll-over step (sdft.h:572-573)
        acc = acc + f * stage[buf][u];
        f = one;
        ++u; c = 0;
        continue;
      }
      auto step = [&](FD dl)
      {
        if constexpr (sizeof(FD) == 4)
        {
          // Pinned, packed sequence on the register pair v[40:41] = {fid component, acc component}:
          //   q        = {f*T1, f*delta}                 v_pk_mul_f32 (f broadcast to both halves)
          //   v40      = partner(f) * T2                 v_mul_f32_dpp, in place (f is consumed)
          //   v[40:41] = q + {partner(f)*T2, acc}        v_pk_add_f32  ->  {f', acc'}
          // Packed f32 mul/add round each half like the scalar ops, so the results are the
          // reference's bit for bit.  The s_nop supplies the second wait state the DPP read of
          // v40 needs after the v_pk_add of the previous step (the v_pk_mul is the first).
          typedef float v2f __attribute__((ext_vector_type(2)));
          v2f e; e.x = f; e.y = acc;
          v2f td; td.x = T1; td.y = dl;
          v2f q;
          asm volatile(
              "v_pk_mul_f32 %[q], v[40:41], %[td] op_sel_hi:[0,1]\n\t"
              "s_nop 0\n\t"
              "v_mul_f32_dpp v40, v40, %[t2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
              "v_pk_add_f32 v[40:41], %[q], v[40:41]"
              : [q] "=&v"(q), "+{v[40:41]}"(e)
              : [td] "v"(td), [t2] "v"(T2));
          f = e.x; acc = e.y;
        }
        else
        {
          const FD g = partner(f);
          acc = acc + f * dl;                            // sdft.h:583
          const FD m1 = f * T1;
          const FD m2 = g * T2;
          f = m1 + m2;                                   // sdft.h:584
        }
      };
      constexpr int R = 8;                               // LDS reads are pipelined R samples ahead
      unsigned i = 0;
      if constexpr (sizeof(FD) == 4)
      {
        // Hand-written inner loop for long runs, 32 samples per trip: the differences come
        // straight from memory over the scalar unit (two alternating s_load_dwordx16 bursts, the
        // next one in flight while the current one is consumed), and a step is four VALU
        // instructions on pinned registers, v[40:41] = {fid component, acc component}:
        //   v42 = f*T1 ; v43 = f*delta ; v40 = partner(f)*T2 (DPP, in place) ;
        //   v[40:41] = v[42:43] + v[40:41]  ->  {f', acc'}
        // The two multiplies between the packed add and the DPP read of v40 are the two wait
        // states that read needs.  Same roundings as the scalar formulation (bit-exact tests).
        // The burst prefetch reads up to 64 floats past the run: the delta buffer is padded.
        if (run >= 32u)
        {
          typedef float v2f __attribute__((ext_vector_type(2)));
          v2f e; e.x = f; e.y = acc;
          unsigned trips = run / 32u;
          const FD* src = d + base + u;
          asm volatile(
              "s_load_dwordx16 s[64:79], s[96:97], 0x0\n\t"
              "s_waitcnt lgkmcnt(0)\n"
              "1:\n\t"
              "s_load_dwordx16 s[80:95], s[96:97], 0x40\n\t"
#define SDFT_EXACT_STEP(sr)                                                                          \
              "v_mul_f32 v42, v40, %[t1]\n\t"                                                        \
              "v_mul_f32 v43, " sr ", v40\n\t"                                                       \
              "v_mul_f32_dpp v40, v40, %[t2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     \
              "v_pk_add_f32 v[40:41], v[42:43], v[40:41]\n\t"
              SDFT_EXACT_STEP("s64") SDFT_EXACT_STEP("s65") SDFT_EXACT_STEP("s66") SDFT_EXACT_STEP("s67")
              SDFT_EXACT_STEP("s68") SDFT_EXACT_STEP("s69") SDFT_EXACT_STEP("s70") SDFT_EXACT_STEP("s71")
              SDFT_EXACT_STEP("s72") SDFT_EXACT_STEP("s73") SDFT_EXACT_STEP("s74") SDFT_EXACT_STEP("s75")
              SDFT_EXACT_STEP("s76") SDFT_EXACT_STEP("s77") SDFT_EXACT_STEP("s78") SDFT_EXACT_STEP("s79")
              "s_waitcnt lgkmcnt(0)\n\t"
              "s_load_dwordx16 s[64:79], s[96:97], 0x80\n\t"
              SDFT_EXACT_STEP("s80") SDFT_EXACT_STEP("s81") SDFT_EXACT_STEP("s82") SDFT_EXACT_STEP("s83")
              SDFT_EXACT_STEP("s84") SDFT_EXACT_STEP("s85") SDFT_EXACT_STEP("s86") SDFT_EXACT_STEP("s87")
              SDFT_EXACT_STEP("s88") SDFT_EXACT_STEP("s89") SDFT_EXACT_STEP("s90") SDFT_EXACT_STEP("s91")
              SDFT_EXACT_STEP("s92") SDFT_EXACT_STEP("s93") SDFT_EXACT_STEP("s94") SDFT_EXACT_STEP("s95")
#undef SDFT_EXACT_STEP
              "s_waitcnt lgkmcnt(0)\n\t"
              "s_add_u32 s96, s96, 0x80\n\t"
              "s_addc_u32 s97, s97, 0\n\t"
              "s_sub_u32 s98, s98, 1\n\t"
              "s_cmp_lg_u32 s98, 0\n\t"
              "s_cbranch_scc1 1b"
              : "+{v[40:41]}"(e), "+{s[96:97]}"(src), "+{s98}"(trips)
              : [t1] "v"(T1), [t2] "v"(T2)
              : "v42", "v43", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76",
                "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91",
                "s92", "s93", "s94", "s95", "scc", "memory");
          f = e.x; acc = e.y;
          i = (run / 32u) * 32u;
        }
      }
      if (run - i >= (unsigned)R)
      {
        FD cur[R];
#pragma unroll
        for (int q = 0; q < R; ++q) cur[q] = stage[buf][u + i + q];
        for (; i + 2 * R <= run; i += R)
        {
          FD nxt[R];
#pragma unroll
          for (int q = 0; q < R; ++q) nxt[q] = stage[buf][u + i + R + q];
#pragma unroll
          for (int q = 0; q < R; ++q) step(cur[q]);
#pragma unroll
          for (int q = 0; q < R; ++q) cur[q] = nxt[q];
        }
#pragma unroll
        for (int q = 0; q < R; ++q) step(cur[q]);
        i += R;
      }
      for (; i < run; ++i) step(stage[buf][u + i]);
      u += run; c += run;
    }
    if (more) put(buf ^ 1, regs);
  }
  if (ends_call) dump(j);                                // carry-in of the call's last chunk
  else if (valid)
  {
    // hand the running state to the next segment's launch
    FD* an = reinterpret_cast<FD*>(a.acc_next);
    FD* fn = reinterpret_cast<FD*>(a.fid_next);
    const size_t o = ((ch * a.nbins) + bin) * 2 + comp;
    an[o] = acc;
    fn[o] = f;
  }
}

// ------------------------------------------------------------------------------------------
// K1a'' (exact carry, chain form)  the same job as carry_exact_kernel -- the reference's rounding
// sequence of acc at every chunk start -- with the serial part cut down to what the reference's
// summation order really dictates: one dependent addition per sample,
//     acc(t+1) = fl( acc(t) + fl( fid(c_t) * delta_t ) )                       (sdft.h:583 / :572).
// Everything else is off the chain, because fid does not depend on the data: it is re-seeded to 1
// at every roll-over (sdft.h:573) and multiplied by a constant otherwise (:584), so fid_k(c) is a
// pure function of (bin, cursor) with period 2N.  fid_seed_kernel tabulates it once per plan at
// every L-th cursor; any block of L consecutive steps can then be regenerated from its seed, and
// blocks of different time are independent.
//
// One workgroup = 32 bins (re / im in a lane pair, as in carry_exact_kernel) = 1 consumer wave +
// P producer waves.  Time runs in rounds of R = P*L steps: in period i producer p regenerates the
// rotations of block p of round i from the seed table (3 VALU per step: f*T1, partner(f)*T2 by
// DPP, add), forms the products fid*delta (1 VALU) and parks them in LDS; the consumer adds the
// products of round i-1 to acc in time order (1 dependent VALU per step + LDS reads) and writes
// acc to `carry` whenever a chunk starts.  One barrier per period; two product buffers.
// The forward kernels seed their own fid from the same table (ForwardArgs::fseed).
// Bit-identical to the serial pass: same operands, same operations, same order on the chain.
// ------------------------------------------------------------------------------------------
template <typename FD> struct ChainArgs
{
  const FD* delta;            // [channels][n]
  const cx<FD>* tw;           // [N]
  const cx<FD>* fseed;        // [2N/L][N]  fid at cursor b*L
  cx<FD>* carry;              // [channels][chunks][N]
  const cx<FD>* acc_state;    // [channels][N]  acc at the first step of this launch
  cx<FD>* acc_next;           // [channels][N]  acc after the last step (nullptr when the launch ends the call)
  size_t n;
  unsigned nbins, chunks, chunk_len, cursor0;
  unsigned chunk0, launch_chunks;
  unsigned L, P;              // block length (divides 2N, multiple of 8), producer waves
  unsigned NB;                // ring form: blocks the LDS ring holds
  unsigned chunk_shift;       // ring form: chunk j > 0 starts at sample j*chunk_len - chunk_shift (0 elsewhere)
  unsigned debug;             // measurement aid: bit 0 = consumer idles, bit 1 = producers idle (results are garbage);
                              // bit 5 (ring form): test aid, the producers stop publishing after their first block
  unsigned long long* stats;  // measurement aid: per wave of workgroup 0, cycles in {work, tail waits, barrier} (or nullptr)
  unsigned* status;           // ring form: word in pinned host memory, incremented by every wave whose poll loop ran out
  unsigned chunks_channels;   // relay form: channels of the plan (relays = bin blocks x channels; P = waves per relay)
  // relay form, flow mode (one relay launch per call, the forward launch waits for carries chunk by chunk):
  unsigned* ready;            // [channels][chunks][bin blocks]: set to ready_seq once this relay's carries of the chunk are in memory
  unsigned ready_seq;
  unsigned* started;          // signal memory: every workgroup adds 1 when it has started (gates the forward launch)
};

template <typename FD>
__global__ __launch_bounds__(kWave) void fid_seed_kernel(const cx<FD>* __restrict__ tw, cx<FD>* __restrict__ fseed,
                                                         unsigned nbins, unsigned L)
{
  const unsigned k = blockIdx.x * kWave + threadIdx.x;
  if (k >= nbins) return;
  const cx<FD> t = tw[k];
  cx<FD> f = cmake<FD>((FD)1, (FD)0);                      // fid at cursor 0 (sdft.h:446, :573)
  const unsigned span = 2u * nbins;
  for (unsigned c = 0; c < span; ++c)
  {
    if (c % L == 0) fseed[(size_t)(c / L) * nbins + k] = f;
    f = cmul(f, t);                                        // sdft.h:584
  }
}

constexpr int kChainSlack = 32;                            // steps of a lane's row the consumer's read-ahead may touch past a round

// LDS image of one product buffer: [lane][S], S = R + slack + one 16-byte vector, so that a lane's
// products of consecutive steps are contiguous (16-byte reads and writes move 4 (FD float) or 2
// (FD double) steps each) and S/(16 bytes) is odd: both the 8-lane groups of ds_write_b128 and the
// 16-lane groups of ds_read_b128 then fall on distinct banks.
template <typename FD> SDFT_HD constexpr int chain_row(int R) { return R + kChainSlack + 16 / (int)sizeof(FD); }

// one producer step on a lane pair: p = fid*delta (fid before its rotation), then fid *= tw.
// FD float is spelled out in ISA: left to itself the compiler packs the two multiplies of the
// rotation into v_pk_mul_f32 / v_pk_add_f32 plus moves (2.5x the issue slots of four plain VALU
// ops on a lone wave).  The DPP read of f needs two wait states after the v_add that wrote it:
// the two plain multiplies at the head of the next step are those.
SDFT_D float chain_step(float& f, float dl, float T1, float T2)
{
  float p, m1, m2;
  asm volatile(
      "v_mul_f32_e32 %[p], %[dl], %[f]\n\t"
      "v_mul_f32_e32 %[m1], %[f], %[t1]\n\t"
      "v_mul_f32_dpp %[m2], %[f], %[t2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_e32 %[f], %[m1], %[m2]"
      : [p] "=&v"(p), [m1] "=&v"(m1), [m2] "=&v"(m2), [f] "+v"(f)
      : [dl] "s"(dl), [t1] "v"(T1), [t2] "v"(T2));
  return p;
}
// lane-pair partner without an `old` operand to set up (bound_ctrl; every lane of a quad has a source)
SDFT_D double partner_nc(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
SDFT_D double chain_step(double& f, double dl, double T1, double T2)
{
  const double p = f * dl;
  const double g = partner_nc(f);
  const double m1 = f * T1;
  const double m2 = g * T2;
  f = m1 + m2;                                             // sdft.h:584
  return p;
}

template <typename FD, int L>
__global__ __launch_bounds__(kWave * 8) void carry_chain_kernel(ChainArgs<FD> a)
{
  constexpr int NV = 16 / (int)sizeof(FD);                 // steps per 16-byte LDS access
  typedef FD vec_t __attribute__((ext_vector_type(NV)));
  extern __shared__ __align__(16) unsigned char chain_lds_raw[];
  FD* prod = reinterpret_cast<FD*>(chain_lds_raw);         // [2][64][S]

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int comp = lane & 1;
  const unsigned bin_blocks = (a.nbins + kWave / 2 - 1) / (kWave / 2);
  const unsigned bin = (blockIdx.x % bin_blocks) * (kWave / 2) + (lane >> 1);
  const size_t ch = blockIdx.x / bin_blocks;
  const bool valid = bin < a.nbins;
  const unsigned kk = valid ? bin : a.nbins - 1;
  const unsigned span = 2u * a.nbins;
  const int P = (int)a.P, R = L * P, S = chain_row<FD>(R);

  // this launch: steps [tb, te) of the call; chunk starts inside it are dumped, the call's very last
  // chunk is run by the forward kernel itself (its start is the final dump)
  const unsigned jend = a.chunk0 + a.launch_chunks;
  const bool ends_call = (jend == a.chunks);
  const long long tb = (long long)a.chunk0 * a.chunk_len;
  const long long total = (long long)(a.launch_chunks - (ends_call ? 1 : 0)) * a.chunk_len;
  // absolute step index u = cursor0 + t; blocks are aligned to multiples of L in u
  const long long u0 = (long long)a.cursor0 + tb, u1 = u0 + total;
  const long long q0 = u0 / L;
  const long long nblocks = (u1 + L - 1) / L - q0;
  const long long rounds = (nblocks + P - 1) / P;

  if (wave == 0) __builtin_amdgcn_s_setprio(3);            // the chain: wins every issue arbitration
  else __builtin_amdgcn_s_setprio(2);

  // consumer state
  FD acc = (FD)0;
  long long done = 0, next_dump = 0;
  unsigned j = a.chunk0;
  FD* carry = reinterpret_cast<FD*>(a.carry);
  if (wave == 0)
  {
    const cx<FD> acc0 = a.acc_state[ch * a.nbins + kk];
    acc = comp ? acc0.im : acc0.re;
  }
  // producer constants
  const cx<FD> tw = a.tw[kk];
  const FD T1 = tw.re;
  const FD T2 = comp ? tw.im : -tw.im;
  const SDFT_CONSTANT FD* dch = as_uniform(a.delta + ch * a.n);

  auto dump = [&]()
  {
    if (valid) carry[(((ch * a.chunks + j) * a.nbins) + bin) * 2 + comp] = acc;
    ++j; next_dump += a.chunk_len;
  };

  // producers: seed and differences of the wave's first block
  cx<FD> sd_next = cmake<FD>((FD)1, (FD)0);
  FD dl_next[L];
#pragma unroll
  for (int s = 0; s < L; ++s) dl_next[s] = (FD)0;
  // cursor of the block whose seed is in sd_next, kept in 32 bits and advanced by R per period (a
  // 64-bit modulo per block costs a lone wave more than the block's arithmetic)
  unsigned cb_next = 0;
  if (wave > 0 && (long long)(wave - 1) < nblocks)
  {
    const long long ub = (q0 + (wave - 1)) * L;
    cb_next = (unsigned)(ub % span);
    sd_next = a.fseed[(size_t)(cb_next / L) * a.nbins + kk];
    if (ub >= u0 && ub + L <= u1)
    {
#pragma unroll
      for (int s = 0; s < L; ++s) dl_next[s] = dch[ub - (long long)a.cursor0 + s];
    }
  }

  unsigned long long st_work = 0, st_tail = 0, st_bar = 0;
  for (long long period = 0; period <= rounds; ++period)
  {
    const unsigned long long stamp0 = a.stats ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long stamp1 = stamp0;
    if (wave == 0)
    {
      if (period > 0 && !(a.debug & 1u))
      {
        const long long r = period - 1;
        const FD* pb = prod + ((size_t)(r & 1) * kWave + lane) * S;      // this lane's row of the round
        const long long ub = (q0 + r * P) * L;             // first step of the round
        int s0 = (int)((u0 > ub ? u0 : ub) - ub);
        const int s1 = (int)((u1 < ub + R ? u1 : ub + R) - ub);
        while (s0 < s1)
        {
          if (done == next_dump) dump();
          int run = s1 - s0;
          if ((long long)run > next_dump - done) run = (int)(next_dump - done);
          const FD* ps = pb + s0;
          int i = 0;
          for (; i < run && ((s0 + i) % NV) != 0; ++i) acc = acc + ps[i];     // up to the next 16-byte boundary
          if (run - i >= 32)
          {
            // Four register sets in rotation: the LDS reads of a group of eight products are issued
            // three groups (24 dependent additions) before the chain consumes them, so the chain never
            // waits for LDS.  The scheduling fences keep the compiler from re-sorting reads behind
            // additions.  (The read-ahead at the end of a run touches steps past it: inside the lane's
            // row, never used.)
            vec_t v0[8 / NV], v1[8 / NV], v2[8 / NV], v3[8 / NV];
            auto fetch = [&](vec_t (&v)[8 / NV], int at)
            {
#pragma unroll
              for (int q = 0; q < 8 / NV; ++q) v[q] = *reinterpret_cast<const vec_t*>(ps + at + q * NV);
              __builtin_amdgcn_sched_barrier(0);
            };
            auto chain8 = [&](const vec_t (&v)[8 / NV])
            {
#pragma unroll
              for (int q = 0; q < 8 / NV; ++q)
#pragma unroll
                for (int e = 0; e < NV; ++e) acc = acc + v[q][e];               // the chain (sdft.h:583)
              __builtin_amdgcn_sched_barrier(0);
            };
            fetch(v0, i); fetch(v1, i + 8); fetch(v2, i + 16);
            for (; i + 32 <= run; i += 32)
            {
              fetch(v3, i + 24); chain8(v0);
              fetch(v0, i + 32); chain8(v1);
              fetch(v1, i + 40); chain8(v2);
              fetch(v2, i + 48); chain8(v3);
            }
          }
          if (run - i >= 8)
          {
            // what is left of the run (< 32 steps): every read first, then the additions
            vec_t v[3][8 / NV];
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
              for (int q = 0; q < 8 / NV; ++q) v[g][q] = *reinterpret_cast<const vec_t*>(ps + i + g * 8 + q * NV);
            const int groups = (run - i) / 8;
#pragma unroll
            for (int g = 0; g < 3; ++g)
              if (g < groups)
              {
#pragma unroll
                for (int q = 0; q < 8 / NV; ++q)
#pragma unroll
                  for (int e = 0; e < NV; ++e) acc = acc + v[g][q][e];
              }
            i += groups * 8;
          }
          if (run - i > 0)
          {
            const int rem = run - i;                         // < 8
            vec_t v[8 / NV];
#pragma unroll
            for (int q = 0; q < 8 / NV; ++q) v[q] = *reinterpret_cast<const vec_t*>(ps + i + q * NV);
#pragma unroll
            for (int q = 0; q < 8 / NV; ++q)
#pragma unroll
              for (int e = 0; e < NV; ++e)
                if (q * NV + e < rem) acc = acc + v[q][e];
          }
          s0 += run; done += run;
        }
      }
    }
    else if (period < rounds && !(a.debug & 2u))
    {
      const long long q = q0 + period * P + (wave - 1);      // this wave's block
      if (q - q0 < nblocks)
      {
        const long long ub = q * L;
        FD* pw = prod + ((size_t)(period & 1) * kWave + lane) * S + (size_t)(wave - 1) * L;
        const long long t_first = ub - (long long)a.cursor0;   // sample index of the block's first step
        const long long qn = q + P, ubn = qn * L;              // the block after this one (next period)
        const bool more = (qn - q0 < nblocks);
        const bool full_n = more && ubn >= u0 && ubn + L <= u1;

        FD f = comp ? sd_next.im : sd_next.re;
        // all products of the block first, then the LDS stores: a store issued in the middle would make
        // the next multiplies wait until it has read its source registers (measured: 35 cycles per
        // ds_write_b32, 140 per ds_write_b128 between dependent VALU work)
        vec_t pv[L / NV];
        if (ub >= u0 && ub + L <= u1)
        {
#pragma unroll
          for (int s = 0; s < L; ++s) pv[s / NV][s % NV] = chain_step(f, dl_next[s], T1, T2);
        }
        else
        {
#pragma unroll
          for (int s = 0; s < L; ++s)                        // ragged first / last block of the launch
          {
            const long long u = ub + s;
            const FD dl = (u >= u0 && u < u1) ? dch[t_first + s] : (FD)0;
            pv[s / NV][s % NV] = chain_step(f, dl, T1, T2);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < L; s += NV) *reinterpret_cast<vec_t*>(pw + s) = pv[s / NV];
        if (a.stats) stamp1 = __builtin_amdgcn_s_memtime();
        // seed and differences of the next block: requested now, they arrive while this wave waits at the
        // barrier for the consumer -- a producer never waits for memory at the top of a round
        unsigned cbn = cb_next + (unsigned)R;
        while (cbn >= span) cbn -= span;
        cb_next = cbn;
        if (more) sd_next = a.fseed[(size_t)(cbn / L) * a.nbins + kk];
        if (full_n)
        {
#pragma unroll
          for (int s = 0; s < L; ++s) dl_next[s] = dch[ubn - (long long)a.cursor0 + s];
        }
      }
    }
    if (a.stats)
    {
      if (wave == 0) stamp1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long stamp2 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long stamp3 = __builtin_amdgcn_s_memtime();
      st_work += stamp1 - stamp0; st_tail += stamp2 - stamp1; st_bar += stamp3 - stamp2;
    }
    else
    __syncthreads();
  }
  if (a.stats && blockIdx.x == 0 && lane == 0)
  {
    a.stats[wave * 4 + 0] = st_work; a.stats[wave * 4 + 1] = st_tail; a.stats[wave * 4 + 2] = st_bar; a.stats[wave * 4 + 3] = (unsigned long long)rounds;
  }
  if (wave == 0)
  {
    if (done == next_dump && j < jend) dump();              // the chunk that starts where this launch ends
    if (!ends_call && valid)
      reinterpret_cast<FD*>(a.acc_next)[((ch * a.nbins) + bin) * 2 + comp] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// K1a-ring (exact carry, ring form)  carry_chain_kernel without rounds and without run bookkeeping.
// There the producers and the consumer of a workgroup meet at a barrier once per round, and what the
// consumer spends per round on bookkeeping (a lone wave pays 4.5 cycles for EVERY instruction, scalar ones
// included), on LDS latency at run starts and at the barrier is half of its time.  Here
//  * the products go through a RING of NB blocks of L steps: producer p fills blocks p, p+P, p+2P, ... as
//    soon as the slot is free and publishes each with a sequence number (ready[slot] = block + 1); the
//    consumer publishes how many blocks it has left behind.  Both sides poll LDS words; the LDS serves a
//    wave's instructions in order, so a flag written after the data (producer) or read before it
//    (consumer) orders them.  Every poll loop is bounded and a time-out is sticky for the workgroup: a
//    protocol error (or a wave starved for seconds) ends the kernel instead of hanging the GPU, and the
//    wave that ran out reports it through ChainArgs::status -- the host then restores the stream state
//    the call started from and re-runs it with the serial pass (Plan::forward_checked);
//  * the host shifts the chunk grid so that every chunk but the first starts on a block boundary
//    (ChainArgs::chunk_shift = cursor0 mod L; the forward kernels use the same grid): the consumer
//    then walks WHOLE blocks -- one block's products are fetched while the previous block's are added,
//    a chunk start is a block counter reaching zero -- and all its bookkeeping is a handful of 32-bit
//    scalar instructions per block.
// ------------------------------------------------------------------------------------------
constexpr int kRingMaxBlocks = 48;
constexpr unsigned kRingPollCap = 1u << 20;

// LDS words of the ring protocol, accessed as workgroup-scope atomics on the __shared__ objects themselves
// (a volatile access through a generic pointer compiles to flat_load/flat_store sc0 sc1 and drags a full
// s_waitcnt behind it -- measured: 325 cycles per block on the consumer)
SDFT_D unsigned ring_peek(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SDFT_D void ring_poke(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
SDFT_D int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a poll loop ran out: sticky for the workgroup (everybody leaves) and reported to the host, which re-runs the call's
// carries with the serial pass (Plan::forward_checked); wave-uniform call
SDFT_D void ring_abort(unsigned* aborted, unsigned* status)
{
  ring_poke(aborted, 1u);
  if (status && (threadIdx.x & (kWave - 1)) == 0) __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename FD, int L>
__global__ __launch_bounds__(kWave * 8) void carry_ring_kernel(ChainArgs<FD> a)
{
  constexpr int NV = 16 / (int)sizeof(FD);                 // steps per 16-byte LDS access
  constexpr int VB = L / NV;                               // 16-byte vectors per block and lane
  typedef FD vec_t __attribute__((ext_vector_type(NV)));
  extern __shared__ __align__(16) unsigned char ring_lds_raw[];
  __shared__ unsigned ready[kRingMaxBlocks];
  __shared__ unsigned consumed_blocks;
  __shared__ unsigned aborted;                             // a poll loop ran out: everybody leaves

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int comp = lane & 1;
  const unsigned bin_blocks = (a.nbins + kWave / 2 - 1) / (kWave / 2);
  const unsigned bin = (blockIdx.x % bin_blocks) * (kWave / 2) + (lane >> 1);
  const size_t ch = blockIdx.x / bin_blocks;
  const bool valid = bin < a.nbins;
  const unsigned kk = valid ? bin : a.nbins - 1;
  const unsigned span = 2u * a.nbins;
  const int P = (int)a.P, NB = (int)a.NB;
  const int S = NB * L + NV;                               // a lane's row: the ring + one vector (odd multiple of 16 B)
  FD* prod = reinterpret_cast<FD*>(ring_lds_raw) + (size_t)lane * S;

  // shifted chunk grid: chunk j starts at sample j*len - shift (chunk 0 at 0), i.e. on a block boundary
  const unsigned jend = a.chunk0 + a.launch_chunks;
  const bool ends_call = (jend == a.chunks);
  const long long len = a.chunk_len, sh = a.chunk_shift;
  const long long tb = a.chunk0 ? a.chunk0 * len - sh : 0;
  const long long te = (long long)(ends_call ? jend - 1 : jend) * len - sh;       // first step this launch does NOT take
  const long long total = te > tb ? te - tb : 0;
  const long long u0 = (long long)a.cursor0 + tb, u1 = u0 + total;               // absolute steps; blocks start at multiples of L
  const long long q0 = u0 / L;
  const int nblocks = (int)((u1 + L - 1) / L - q0);        // u1 is a block boundary whenever total > 0
  const int off0 = (int)(u0 - q0 * L);                     // > 0 only for a launch that starts the call mid-block

  if (threadIdx.x < kRingMaxBlocks) ready[threadIdx.x] = 0;
  if (threadIdx.x == 0) { consumed_blocks = 0; aborted = 0; }
  __syncthreads();

  if (wave == 0)
  {
    // ---------------- consumer: the chain ----------------
    __builtin_amdgcn_s_setprio(3);
    if (a.debug & 8u) return;                                // measurement aid: producers alone
    const cx<FD> acc0 = a.acc_state[ch * a.nbins + kk];
    FD acc = comp ? acc0.im : acc0.re;
    unsigned j = a.chunk0;
    FD* cptr = reinterpret_cast<FD*>(a.carry) + (((ch * a.chunks + a.chunk0) * a.nbins) + kk) * 2 + comp;
    const size_t cstride = (size_t)a.nbins * 2;
    const int blocks_per_chunk = (int)(len / L);
    int ready_upto = 0;                                     // blocks known to be published (wave-uniform)
    auto dump = [&]() { if (valid) *cptr = acc; cptr += cstride; ++j; };
    // Wait until `want` blocks are published: one lane per slot looks at the flags; returns the new count
    // of consecutive published blocks, or -1 when the poll budget ran out (sticky for the workgroup).
    auto await = [&](int have, int want) -> int
    {
      if (a.debug & 4u) return want;                         // measurement aid: the consumer runs free (garbage results)
      unsigned polls = 0;
      for (;;)
      {
        const int g = have + lane;
        int slot = g % NB;
        const unsigned flag = (lane < NB) ? ring_peek(&ready[slot]) : 0u;
        const unsigned long long mask = __ballot(lane < NB && g < nblocks && flag == (unsigned)(g + 1));
        have = uniform(have + (int)__builtin_ctzll(~mask));  // consecutive published blocks
        if (have >= want) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return have; }
        if (ring_peek(&aborted)) return -1;
        if (++polls > kRingPollCap) { ring_abort(&aborted, a.status); return -1; }
        __builtin_amdgcn_s_sleep(1);
      }
    };
    int fslot = 0;                                          // ring slot of the next block to be fetched
    auto fetch = [&](vec_t (&v)[VB])
    {
      const FD* ps = prod + fslot * L;
      fslot = uniform((fslot + 1 == NB) ? 0 : fslot + 1);
#pragma unroll
      for (int q = 0; q < VB; ++q) v[q] = *reinterpret_cast<const vec_t*>(ps + q * NV);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto chain = [&](const vec_t (&v)[VB])
    {
#pragma unroll
      for (int q = 0; q < VB; ++q)
#pragma unroll
        for (int e = 0; e < NV; ++e) acc = acc + v[q][e];                           // the chain (sdft.h:583)
      __builtin_amdgcn_sched_barrier(0);
    };

    dump();                                                 // the chunk this launch starts with
    int to_dump = blocks_per_chunk;                         // whole blocks until the next chunk start
    int b = 0;
    if (nblocks > 0 && off0 > 0)
    {
      // the call starts mid-block (chunk 0 only): steps off0 .. L-1 of block 0, one by one
      ready_upto = await(0, 1);
      if (ready_upto > 0)
        for (int s = off0; s < L; ++s) acc = acc + prod[s];
      if (lane == 0) ring_poke(&consumed_blocks, 1u);
      b = 1; fslot = (NB == 1) ? 0 : 1; --to_dump;          // the shifted first chunk ends blocks_per_chunk - 1 blocks later
      if (to_dump == 0 && b < nblocks) { dump(); to_dump = blocks_per_chunk; }
    }
    if (ready_upto >= 0 && b < nblocks)
    {
      // block b's products are in v0 while block b+1's are fetched into v1, and vice versa; a block is
      // fetched only after its flag has been seen
      vec_t v0[VB], v1[VB];
      const int last = nblocks - 1;
      if (ready_upto < b + 1) ready_upto = await(ready_upto, b + 1);
      if (ready_upto >= 0) fetch(v0);
      while (ready_upto >= 0)
      {
        if (b < last) { if (ready_upto < b + 2) ready_upto = await(ready_upto, b + 2); if (ready_upto < 0) break; fetch(v1); }
        chain(v0);
        b = uniform(b + 1); to_dump = uniform(to_dump - 1);
        if (b > last) break;
        if (to_dump == 0) { dump(); to_dump = blocks_per_chunk; }
        if (b < last) { if (ready_upto < b + 2) ready_upto = await(ready_upto, b + 2); if (ready_upto < 0) break; fetch(v0); }
        chain(v1);
        b = uniform(b + 1); to_dump = uniform(to_dump - 1);
        if (lane == 0) ring_poke(&consumed_blocks, (unsigned)b);               // once per pair of blocks
        if (b > last) break;
        if (to_dump == 0) { dump(); to_dump = blocks_per_chunk; }
      }
      if (lane == 0) ring_poke(&consumed_blocks, (unsigned)b);
    }
    if (j < jend) dump();                                   // the chunk that starts where this launch ends
    if (!ends_call && valid)
      reinterpret_cast<FD*>(a.acc_next)[((ch * a.nbins) + bin) * 2 + comp] = acc;
    return;
  }

  // ---------------- producers ----------------
  __builtin_amdgcn_s_setprio(2);
  if (wave > P || (a.debug & 4u)) return;
  const cx<FD> tw = a.tw[kk];
  const FD T1 = tw.re;
  const FD T2 = comp ? tw.im : -tw.im;
  int g = wave - 1;                                          // this wave's block (relative to q0)
  if (g >= nblocks) return;
  if constexpr (sizeof(FD) == 4)
  {
    // FD float: the block's differences sit in L scalar registers, requested one block ahead (FD double
    // would need 64 of them for 32 steps and spills: it takes the vector form below)
    const SDFT_CONSTANT FD* dch = as_uniform(a.delta + ch * a.n);
    unsigned cb = (unsigned)(((q0 + g) * L) % span);         // its cursor, kept in 32 bits from here on
    const unsigned step_cb = (unsigned)(((long long)P * L) % span);
    cx<FD> sd = a.fseed[(size_t)(cb / L) * a.nbins + kk];
    FD dl[L];
#pragma unroll
    for (int s = 0; s < L; ++s) dl[s] = (FD)0;
    {
      const long long ub = (q0 + g) * L;
      if (ub >= u0 && ub + L <= u1)
      {
#pragma unroll
        for (int s = 0; s < L; ++s) dl[s] = dch[ub - (long long)a.cursor0 + s];
      }
    }
    unsigned seen_consumed = 0;
    int pslot = g % NB;                                        // ring slot of block g, advanced by P per block
    const int pstep = P % NB;
    for (; g < nblocks; g += P)
    {
      const long long ub = (q0 + g) * L;
      FD f = comp ? sd.im : sd.re;
      vec_t pv[VB];
      if (ub >= u0 && ub + L <= u1)
      {
#pragma unroll
        for (int s = 0; s < L; ++s) pv[s / NV][s % NV] = chain_step(f, dl[s], T1, T2);
      }
      else
      {
#pragma unroll
        for (int s = 0; s < L; ++s)                            // the block the call starts in
        {
          const long long u = ub + s;
          const FD d1 = (u >= u0 && u < u1) ? dch[ub - (long long)a.cursor0 + s] : (FD)0;
          pv[s / NV][s % NV] = chain_step(f, d1, T1, T2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the next block's seed and differences: on their way while this one waits for its slot
      const int gn = g + P;
      cb += step_cb; if (cb >= span) cb -= span;
      if (gn < nblocks)
      {
        sd = a.fseed[(size_t)(cb / L) * a.nbins + kk];
        const long long ubn = (q0 + gn) * L;
        if (ubn >= u0 && ubn + L <= u1)
        {
#pragma unroll
          for (int s = 0; s < L; ++s) dl[s] = dch[ubn - (long long)a.cursor0 + s];
        }
      }
      // the slot is free once the consumer has left block g - NB behind
      if (g >= NB && !(a.debug & 8u))
      {
        unsigned polls = 0;
        while ((int)seen_consumed < g - NB + 1)
        {
          seen_consumed = ring_peek(&consumed_blocks);
          if ((int)seen_consumed >= g - NB + 1) break;
          if (ring_peek(&aborted)) return;
          if (++polls > kRingPollCap) { ring_abort(&aborted, a.status); return; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if ((a.debug & 32u) && g >= P) return;                 // test aid: a producer that died (the consumer's poll runs out)
      FD* pw = prod + pslot * L;
#pragma unroll
      for (int s = 0; s < L; s += NV) *reinterpret_cast<vec_t*>(pw + s) = pv[s / NV];
      // publish: release store, i.e. the flag is written after the products have landed
      if (lane == 0) __hip_atomic_store(&ready[pslot], (unsigned)(g + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      pslot += pstep; if (pslot >= NB) pslot -= NB;
    }
  }
  else
  {
    // A block's differences travel in ONE vector register (lane s holds the block's step s; steps outside
    // the launch read as 0) and are handed to the arithmetic by v_readlane, its seed in two; both are
    // requested TWO blocks ahead.  Under a saturated HBM (the forward kernel of an earlier segment is
    // streaming the matrix) a load takes microseconds: one block ahead left the producers waiting and
    // stretched the pass by a third, and scalar loads cannot be queued that deep (64 SGPRs per block).
    const FD* dvec = a.delta + ch * a.n;
    auto load_delta = [&](int blk) -> FD
    {
      const long long u = (q0 + blk) * L + lane;
      const bool in = lane < L && blk < nblocks && u >= u0 && u < u1;
      return in ? dvec[(size_t)(u - (long long)a.cursor0)] : (FD)0;
    };
    unsigned cb0 = (unsigned)(((q0 + g) * L) % span);          // cursors of blocks g, g+P, g+2P, kept in 32 bits
    const unsigned step_cb = (unsigned)(((long long)P * L) % span);
    unsigned cb1 = cb0 + step_cb; if (cb1 >= span) cb1 -= span;
    unsigned cb2 = cb1 + step_cb; if (cb2 >= span) cb2 -= span;
    auto load_seed = [&](unsigned cbx, int blk) -> cx<FD>
    {
      return blk < nblocks ? a.fseed[(size_t)(cbx / L) * a.nbins + kk] : cmake<FD>((FD)1, (FD)0);
    };
    FD d0 = load_delta(g), d1 = load_delta(g + P), d2;
    cx<FD> sd0 = load_seed(cb0, g), sd1 = load_seed(cb1, g + P), sd2;
    unsigned seen_consumed = 0;
    int pslot = g % NB;                                        // ring slot of block g, advanced by P per block
    const int pstep = P % NB;
    for (; g < nblocks; g += P)
    {
      d2 = load_delta(g + 2 * P);
      sd2 = load_seed(cb2, g + 2 * P);
      __builtin_amdgcn_sched_barrier(0);
      FD f = comp ? sd0.im : sd0.re;
      vec_t pv[VB];
#pragma unroll
      for (int s = 0; s < L; ++s)
      {
        FD dl;
        if constexpr (sizeof(FD) == 4) dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d0), s));
        else dl = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d0), s), __builtin_amdgcn_readlane(__double2loint(d0), s));
        pv[s / NV][s % NV] = chain_step(f, dl, T1, T2);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the slot is free once the consumer has left block g - NB behind
      if (g >= NB && !(a.debug & 8u))
      {
        unsigned polls = 0;
        while ((int)seen_consumed < g - NB + 1)
        {
          seen_consumed = ring_peek(&consumed_blocks);
          if ((int)seen_consumed >= g - NB + 1) break;
          if (ring_peek(&aborted)) return;
          if (++polls > kRingPollCap) { ring_abort(&aborted, a.status); return; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if ((a.debug & 32u) && g >= P) return;                 // test aid: a producer that died (the consumer's poll runs out)
      FD* pw = prod + pslot * L;
#pragma unroll
      for (int s = 0; s < L; s += NV) *reinterpret_cast<vec_t*>(pw + s) = pv[s / NV];
      // publish: release store, i.e. the flag is written after the products have landed
      if (lane == 0) __hip_atomic_store(&ready[pslot], (unsigned)(g + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      pslot += pstep; if (pslot >= NB) pslot -= NB;
      d0 = d1; d1 = d2; sd0 = sd1; sd1 = sd2;
      cb2 += step_cb; if (cb2 >= span) cb2 -= span;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1a-relay (exact carry, relay form; round 3)  The chain  acc(t+1) = fl( acc(t) + fl( fid(c_t) * delta_t ) )  costs a
// lone wave one issue slot (4.5 cycles) per step; in the ring form the consumer wave spends as many slots again on
// fetching the products from LDS (a ds_read_b128 is 11 cycles of issue) and on the ring protocol: 9.5 cycles per step.
// Here NO wave fetches products: the C waves of a workgroup are identical and take the blocks of L steps in turn --
// wave w owns blocks w, w + C, w + 2C, ... .  While the other waves hold the chain, a wave regenerates the rotations of
// its next block from the seed table and leaves the block's L products IN ITS OWN REGISTERS (4 VALU per step, off the
// chain); when the token -- the running acc, a sequence number beside it in one LDS word pair per lane -- reaches it, it
// adds its L registers to acc in time order (1 dependent VALU per step, nothing else) and passes the token on.  The
// chain's cost per step is 4.5 cycles + (token hand-off) / L; no LDS ring, no flags, no producer / consumer roles.
//  * differences: one vector load per 16 steps, the same 16 values in each row of 16 lanes, requested a whole turn
//    ahead (microseconds, so a saturated HBM does not stall the wave); the product instruction picks step s with the
//    DPP row broadcast (v_mul_f32_dpp ... row_newbcast:s): no scalar registers, no v_readlane;
//  * one seed per block (the rotation runs through the block; blocks start on multiples of L of the cursor and L
//    divides 2N, so a roll-over -- fid = 1 exactly, sdft.h:573 -- is always a block start and a seed);
//  * the chunk grid is the ring form's (shifted onto block boundaries): a chunk start is a block start, the wave that
//    receives the token there stores acc to `carry`; a call that starts mid-block gives its first block -0.0 for the
//    steps before it (x + -0.0 == x for every x, bit for bit);
//  * polls are bounded; a time-out is reported through ChainArgs::status like the ring form's.
// Same operands, same operations, same order on the chain: bit-identical to the serial pass.
// ------------------------------------------------------------------------------------------
constexpr unsigned kRelayPollCap = 1u << 22;

// Off the chain: the products of up to 16 consecutive steps, p_s = delta_s * fid (delta_s picked from lane s of the row by
// the DPP broadcast), each followed by the rotation fid = fid*T1 + partner(fid)*T2 (sdft.h:584 on a lane pair, as
// chain_step).  FD float: ONE asm statement per 16 steps -- between two asm statements the compiler's hazard recogniser
// has to assume the worst and puts an s_nop; inside, the two plain multiplies are the wait states the DPP read of fid needs.
#define SDFT_RELAY_STEP(i)                                                                                        \
  "v_mul_f32_dpp %[p" #i "], %[d], %[f] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"                         \
  "v_mul_f32_e32 %[m1], %[f], %[t1]\n\t"                                                                          \
  "v_mul_f32_dpp %[m2], %[f], %[t2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                            \
  "v_add_f32_e32 %[f], %[m1], %[m2]\n\t"
template <int COUNT> SDFT_D void relay_products(float* v, float& f, float d, float T1, float T2)
{
  float m1, m2;
  if constexpr (COUNT == 16)
    asm volatile(SDFT_RELAY_STEP(0) SDFT_RELAY_STEP(1) SDFT_RELAY_STEP(2) SDFT_RELAY_STEP(3) SDFT_RELAY_STEP(4) SDFT_RELAY_STEP(5)
                 SDFT_RELAY_STEP(6) SDFT_RELAY_STEP(7) SDFT_RELAY_STEP(8) SDFT_RELAY_STEP(9) SDFT_RELAY_STEP(10) SDFT_RELAY_STEP(11)
                 SDFT_RELAY_STEP(12) SDFT_RELAY_STEP(13) SDFT_RELAY_STEP(14) SDFT_RELAY_STEP(15)
                 : [p0] "=&v"(v[0]), [p1] "=&v"(v[1]), [p2] "=&v"(v[2]), [p3] "=&v"(v[3]), [p4] "=&v"(v[4]), [p5] "=&v"(v[5]),
                   [p6] "=&v"(v[6]), [p7] "=&v"(v[7]), [p8] "=&v"(v[8]), [p9] "=&v"(v[9]), [p10] "=&v"(v[10]), [p11] "=&v"(v[11]),
                   [p12] "=&v"(v[12]), [p13] "=&v"(v[13]), [p14] "=&v"(v[14]), [p15] "=&v"(v[15]),
                   [m1] "=&v"(m1), [m2] "=&v"(m2), [f] "+v"(f)
                 : [d] "v"(d), [t1] "v"(T1), [t2] "v"(T2));
  else
    asm volatile(SDFT_RELAY_STEP(0) SDFT_RELAY_STEP(1) SDFT_RELAY_STEP(2) SDFT_RELAY_STEP(3) SDFT_RELAY_STEP(4) SDFT_RELAY_STEP(5)
                 SDFT_RELAY_STEP(6) SDFT_RELAY_STEP(7)
                 : [p0] "=&v"(v[0]), [p1] "=&v"(v[1]), [p2] "=&v"(v[2]), [p3] "=&v"(v[3]), [p4] "=&v"(v[4]), [p5] "=&v"(v[5]),
                   [p6] "=&v"(v[6]), [p7] "=&v"(v[7]), [m1] "=&v"(m1), [m2] "=&v"(m2), [f] "+v"(f)
                 : [d] "v"(d), [t1] "v"(T1), [t2] "v"(T2));
}
#undef SDFT_RELAY_STEP
// FD double: v_mul_f64 has no DPP form -- the difference is broadcast by two moves, the partner by two more; the step is
// spelled out all the same, on pinned registers (v[8:9] = fid, v[10:15] scratch), because left to the compiler the
// products sink towards their use: it keeps every step's fid and difference alive and multiplies right before the
// chain (256 VGPRs + scratch at 64 steps, 2400 cycles on the chain per block instead of 350).
#define SDFT_RELAY_STEP_D(i)                                                                      \
  "v_mov_b32_dpp v12, %[dlo] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_mov_b32_dpp v13, %[dhi] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_mov_b32_dpp v10, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
  "v_mov_b32_dpp v11, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
  "v_mul_f64 %[p" #i "], v[8:9], v[12:13]\n\t"                                                     \
  "v_mul_f64 v[14:15], v[8:9], %[t1]\n\t"                                                          \
  "v_mul_f64 v[10:11], v[10:11], %[t2]\n\t"                                                        \
  "v_add_f64 v[8:9], v[14:15], v[10:11]\n\t"
template <int COUNT> SDFT_D void relay_products(double* v, double& f, double d, double T1, double T2)
{
  const int dlo = __double2loint(d), dhi = __double2hiint(d);
  if constexpr (COUNT == 16)
    asm volatile(SDFT_RELAY_STEP_D(0) SDFT_RELAY_STEP_D(1) SDFT_RELAY_STEP_D(2) SDFT_RELAY_STEP_D(3) SDFT_RELAY_STEP_D(4) SDFT_RELAY_STEP_D(5)
                 SDFT_RELAY_STEP_D(6) SDFT_RELAY_STEP_D(7) SDFT_RELAY_STEP_D(8) SDFT_RELAY_STEP_D(9) SDFT_RELAY_STEP_D(10) SDFT_RELAY_STEP_D(11)
                 SDFT_RELAY_STEP_D(12) SDFT_RELAY_STEP_D(13) SDFT_RELAY_STEP_D(14) SDFT_RELAY_STEP_D(15)
                 : [p0] "=&v"(v[0]), [p1] "=&v"(v[1]), [p2] "=&v"(v[2]), [p3] "=&v"(v[3]), [p4] "=&v"(v[4]), [p5] "=&v"(v[5]),
                   [p6] "=&v"(v[6]), [p7] "=&v"(v[7]), [p8] "=&v"(v[8]), [p9] "=&v"(v[9]), [p10] "=&v"(v[10]), [p11] "=&v"(v[11]),
                   [p12] "=&v"(v[12]), [p13] "=&v"(v[13]), [p14] "=&v"(v[14]), [p15] "=&v"(v[15]), "+{v[8:9]}"(f)
                 : [dlo] "v"(dlo), [dhi] "v"(dhi), [t1] "v"(T1), [t2] "v"(T2)
                 : "v10", "v11", "v12", "v13", "v14", "v15");
  else
    asm volatile(SDFT_RELAY_STEP_D(0) SDFT_RELAY_STEP_D(1) SDFT_RELAY_STEP_D(2) SDFT_RELAY_STEP_D(3) SDFT_RELAY_STEP_D(4) SDFT_RELAY_STEP_D(5)
                 SDFT_RELAY_STEP_D(6) SDFT_RELAY_STEP_D(7)
                 : [p0] "=&v"(v[0]), [p1] "=&v"(v[1]), [p2] "=&v"(v[2]), [p3] "=&v"(v[3]), [p4] "=&v"(v[4]), [p5] "=&v"(v[5]),
                   [p6] "=&v"(v[6]), [p7] "=&v"(v[7]), "+{v[8:9]}"(f)
                 : [dlo] "v"(dlo), [dhi] "v"(dhi), [t1] "v"(T1), [t2] "v"(T2)
                 : "v10", "v11", "v12", "v13", "v14", "v15");
}
#undef SDFT_RELAY_STEP_D
// products of a whole block into v[L]
template <typename FD, int L> SDFT_D void relay_block(FD (&v)[L], FD& f, const FD (&dv)[(L + 15) / 16], FD T1, FD T2)
{
  if constexpr (L >= 16)
  {
#pragma unroll
    for (int q = 0; q < L / 16; ++q) relay_products<16>(&v[16 * q], f, dv[q], T1, T2);
  }
  else relay_products<8>(&v[0], f, dv[0], T1, T2);
}

// the token: acc and the number of the block it is for, in ONE LDS access per lane (8 bytes for FD float, 16 for FD
// double: a lane's bytes of a ds_write_b64 / ds_write_b128 land in one LDS cycle, so a reader never sees half a token)
template <typename FD> struct RelayToken;
template <> struct RelayToken<float>
{
  typedef unsigned raw_t __attribute__((ext_vector_type(2)));
  static SDFT_D raw_t pack(float acc, unsigned seq) { raw_t r; r.x = (unsigned)__float_as_int(acc); r.y = seq; return r; }
  static SDFT_D float acc(raw_t r) { return __int_as_float((int)r.x); }
  static SDFT_D unsigned seq(raw_t r) { return r.y; }
};
template <> struct RelayToken<double>
{
  typedef unsigned raw_t __attribute__((ext_vector_type(4)));
  static SDFT_D raw_t pack(double acc, unsigned seq)
  {
    raw_t r; r.x = (unsigned)__double2loint(acc); r.y = (unsigned)__double2hiint(acc); r.z = seq; r.w = seq; return r;
  }
  static SDFT_D double acc(raw_t r) { return __hiloint2double((int)r.y, (int)r.x); }
  static SDFT_D unsigned seq(raw_t r) { return r.z; }
};

// Waiting for the token of block `want`: the loop is spelled out in ISA -- left to the compiler, the not-yet path of a
// bounded poll loop is a dozen scalar instructions and branches, and a lone wave pays 4.5 cycles for each of them
// (measured: 370 cycles from one wave's publish to the next wave's first addition, against 140 with this loop,
// scripts/relay_probe.hip).  One ds_read per poll of the lane's own {acc, seq}; the token has arrived when every lane
// sees `want`.  Returns false after kRelayPollRound polls (the caller then looks at the abort flag and tries again).
constexpr unsigned kRelayPollRound = 1u << 12;
template <typename FD> SDFT_D bool relay_wait(volatile __attribute__((address_space(3))) typename RelayToken<FD>::raw_t* mailbox, unsigned want, FD& acc);
template <> SDFT_D bool relay_wait<float>(volatile __attribute__((address_space(3))) RelayToken<float>::raw_t* mailbox, unsigned want, float& acc)
{
  unsigned left = kRelayPollRound;
  unsigned long long tk;
  const unsigned addr = (unsigned)(unsigned long long)mailbox;
  asm volatile(
      "1:\n\t"
      "ds_read_b64 v[4:5], %[addr]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_ne_u32_e32 vcc, %[want], v5\n\t"
      "s_cbranch_vccz 2f\n\t"
      "s_sub_u32 %[left], %[left], 1\n\t"
      "s_cmp_lg_u32 %[left], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "2:"
      : "={v[4:5]}"(tk), [left] "+s"(left)
      : [addr] "v"(addr), [want] "s"(want)
      : "vcc", "scc", "memory");
  acc = __int_as_float((int)(unsigned)(tk & 0xffffffffull));
  return left != 0;
}
template <> SDFT_D bool relay_wait<double>(volatile __attribute__((address_space(3))) RelayToken<double>::raw_t* mailbox, unsigned want, double& acc)
{
  unsigned left = kRelayPollRound;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 tk;
  const unsigned addr = (unsigned)(unsigned long long)mailbox;
  asm volatile(
      "1:\n\t"
      "ds_read_b128 v[4:7], %[addr]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_ne_u32_e32 vcc, %[want], v6\n\t"
      "s_cbranch_vccz 2f\n\t"
      "s_sub_u32 %[left], %[left], 1\n\t"
      "s_cmp_lg_u32 %[left], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "2:"
      : "={v[4:7]}"(tk), [left] "+s"(left)
      : [addr] "v"(addr), [want] "s"(want)
      : "vcc", "scc", "memory");
  acc = __hiloint2double((int)tk.y, (int)tk.x);
  return left != 0;
}

// STATS: measurement build (ChainArgs::stats; instantiated for the longest block only)
// A workgroup may hold TWO relays (ChainArgs::P waves each, 32 bins each): the pass then occupies half as many CUs, and
// the forward launches of earlier segments -- whose 16-wave workgroups cannot share a CU with it -- keep three quarters
// of the chip instead of half (config 3: 128 relays).  FD float only: 12 waves of <= 168 registers fit a CU, FD double's
// 185 registers allow 8.
template <typename FD> struct relay_limits { static constexpr int waves = sizeof(FD) == 4 ? 12 : 8; static constexpr int groups = sizeof(FD) == 4 ? 2 : 1; };
template <typename FD, int L, bool STATS = false>
__global__ __launch_bounds__(kWave * relay_limits<FD>::waves) void carry_relay_kernel(ChainArgs<FD> a)
{
  constexpr int DV = (L + 15) / 16;                        // difference vectors per block (16 steps each)
  using token = RelayToken<FD>;
  using raw_t = typename token::raw_t;
  __shared__ __align__(16) raw_t mails[relay_limits<FD>::groups][kWave];
  __shared__ unsigned aborted;

  const int lane = threadIdx.x & (kWave - 1);
  const int wave_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = (int)a.P;                                  // waves per relay
  const int groups = (int)(blockDim.x >> 6) / C;           // relays in this workgroup
  const int group = wave_wg / C, wave = wave_wg - group * C;
  raw_t* const mail = mails[group];
  const int comp = lane & 1;
  const unsigned bin_blocks = (a.nbins + kWave / 2 - 1) / (kWave / 2);
  const unsigned relay = blockIdx.x * (unsigned)groups + (unsigned)group;            // 32 bins of one channel
  const bool idle = relay >= bin_blocks * a.chunks_channels;                          // odd count: the last workgroup's second relay
  const unsigned bin = ((idle ? 0u : relay) % bin_blocks) * (kWave / 2) + (lane >> 1);
  const size_t ch = (idle ? 0u : relay) / bin_blocks;
  const bool valid = bin < a.nbins && !idle;
  const unsigned kk = valid ? bin : a.nbins - 1;
  const unsigned span = 2u * a.nbins;

  // shifted chunk grid, as in carry_ring_kernel: chunk j starts at sample j*len - shift (chunk 0 at 0)
  const unsigned jend = a.chunk0 + a.launch_chunks;
  const bool ends_call = (jend == a.chunks);
  const long long len = a.chunk_len, sh = a.chunk_shift;
  const long long tb = a.chunk0 ? a.chunk0 * len - sh : 0;
  const long long te = (long long)(ends_call ? jend - 1 : jend) * len - sh;       // first step this launch does NOT take
  const long long total = te > tb ? te - tb : 0;
  const long long u0 = (long long)a.cursor0 + tb, u1 = u0 + total;               // absolute steps; blocks start at multiples of L
  const long long q0 = u0 / L;
  const int nblocks = (int)((u1 + L - 1) / L - q0);        // u1 is a block boundary whenever total > 0
  const int off0 = (int)(u0 - q0 * L);                     // > 0 only for a launch that starts the call mid-block
  const int bpc = (int)(len / L);                          // blocks per chunk

  FD* const carry0 = reinterpret_cast<FD*>(a.carry) + (((ch * a.chunks + a.chunk0) * a.nbins) + kk) * 2 + comp;
  const size_t cstride = (size_t)a.nbins * 2;
  const cx<FD> acc00 = a.acc_state[ch * a.nbins + kk];
  const FD acc0 = comp ? acc00.im : acc00.re;
  // the token starts in the mailbox: block 0 "receives" the state like every other block receives its predecessor's acc
  if (wave == 0) mail[lane] = token::pack(acc0, 0u);
  if (threadIdx.x == 0)
  {
    aborted = 0;
    if (a.started) __hip_atomic_fetch_add(a.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // resident: the forward launch may go
  }
  __syncthreads();
  if (idle) return;
  // flow mode: this relay's flag of chunk (chunk0 + j); a carry is stored write-through, waited for, then flagged
  unsigned* const flags = a.ready ? a.ready + (ch * a.chunks + a.chunk0) * (size_t)bin_blocks + relay % bin_blocks : nullptr;
  auto put_carry = [&](size_t j, FD value)
  {
    if (valid) __hip_atomic_store(carry0 + j * cstride, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto flag_chunk = [&](size_t j)                          // wave-uniform; the carries of chunk j by this wave are stored
  {
    if (!flags) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(flags + j * bin_blocks, a.ready_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // the chunk that starts with the launch's first block: its carry-in is the state (off every chain)
  if (wave == 0) { put_carry(0, acc0); flag_chunk(0); }
  if (nblocks <= 0) return;                                // the launch is the call's last chunk alone

  const cx<FD> tw = a.tw[kk];
  const FD T1 = tw.re;
  const FD T2 = comp ? tw.im : -tw.im;
  const FD* dch = a.delta + ch * a.n;
  const FD* fseed = reinterpret_cast<const FD*>(a.fseed) + (size_t)kk * 2 + comp;     // this lane's component of a seed row
  const int sub = lane & 15;
  // LDS accesses through address-space pointers (a volatile access through a generic pointer compiles to flat_load sc0 sc1
  // plus a full wait)
  typedef volatile __attribute__((address_space(3))) raw_t* lds_token_p;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  lds_token_p my_mail = (lds_token_p)&mail[lane];
#pragma clang diagnostic pop

  // a block's inputs: DV difference vectors (lane l of every row of 16 holds step 16q + l%16) and this lane's component
  // of the seed.  Unconditional loads (the compiler can then count them: the products of a turn wait for the loads of
  // the turn before, not for the ones just issued).  Only the call's first block can be ragged, and that one does not
  // come here.
  auto load_block = [&](int g, unsigned cb, FD (&dv)[DV], FD& sd)
  {
    sd = fseed[(size_t)(cb / L) * a.nbins * 2];
    const FD* d0 = dch + (size_t)((q0 + g) * L - (long long)a.cursor0) + (sub < L ? sub : 0);
#pragma unroll
    for (int q = 0; q < DV; ++q) dv[q] = d0[16 * q];
  };

  int g = wave;
  if (g >= nblocks) return;
  int gm = g % bpc, gj = g / bpc;                          // block g starts chunk chunk0 + gj iff gm == 0
  unsigned cb = (unsigned)(((q0 + g) * L) % span);         // cursor of block g, kept in 32 bits from here on
  const unsigned step_cb = (unsigned)(((long long)C * L) % span);
  const bool die = (a.debug & 32u) != 0;                    // test aid: a wave that keeps the token

  // what a block owes to memory once its token is stored; advances (g, cb, gm, gj) to the wave's next block, false = done
  long long owed = -1;                                      // flow mode: chunk whose flag this wave still has to set
  auto after_block = [&](FD acc) -> bool
  {
    // the chunk that starts with the NEXT block takes the acc this block ends with
    gm += 1;
    if (gm == bpc && g + 1 < nblocks) { put_carry((size_t)(gj + 1), acc); owed = gj + 1; }
    if (g == nblocks - 1)
    {
      if (ends_call) { put_carry((size_t)(jend - 1 - a.chunk0), acc); owed = (long long)(jend - 1 - a.chunk0); }   // carry-in of the call's last chunk
      else if (valid) reinterpret_cast<FD*>(a.acc_next)[((ch * a.nbins) + bin) * 2 + comp] = acc;
    }
    g += C;
    cb += step_cb; if (cb >= span) cb -= span;
    gm += C - 1; while (gm >= bpc) { gm -= bpc; ++gj; }
    if (g >= nblocks && owed >= 0) { flag_chunk((size_t)owed); owed = -1; }       // last turn: nothing later to hide the wait behind
    return g < nblocks;
  };

  if (g == 0 && off0 > 0)
  {
    // The call starts mid-block (first launch of a call only): steps off0 .. L-1 of block 0, one by one, before
    // the relay proper (once per call; kept out of the loop below, whose every instruction is on or near the chain)
    FD acc = acc0;
    FD f = fseed[(size_t)(cb / L) * a.nbins * 2];
    const SDFT_CONSTANT FD* du = as_uniform(dch);
    for (int s = 0; s < L; ++s)
    {
      const bool in = s >= off0;
      const FD dl = in ? du[s - off0] : (FD)0;
      const FD pr = chain_step(f, dl, T1, T2);
      if (in) acc = acc + pr;
    }
    *my_mail = token::pack(acc, 1u);
    if (!after_block(acc)) return;
    if (owed >= 0) { flag_chunk((size_t)owed); owed = -1; }
  }

  FD dv[DV], dvn[DV], sd = (FD)0, sdn = (FD)0;
#pragma unroll
  for (int q = 0; q < DV; ++q) dvn[q] = (FD)0;
  load_block(g, cb, dv, sd);
  __builtin_amdgcn_s_setprio(1);

  // measurement aids (STATS builds): debug bit 6 = cycles of every wave of workgroup 0 in {products, waiting for the
  // token, chain, rest of the turn}; bit 7 = stamps of its first 1024 turns (token seen, additions done, token stored)
  const bool timed = STATS && a.stats != nullptr && relay == 0 && !(a.debug & 128u);
  const bool stamped = STATS && a.stats != nullptr && relay == 0 && (a.debug & 128u);
  unsigned long long st_prod = 0, st_poll = 0, st_chain = 0, st_rest = 0, st_turns = 0, tA = 0, tB = 0, tC = 0;
  unsigned long long stamp = timed ? __builtin_amdgcn_s_memtime() : 0;
  auto lap = [&](unsigned long long& bucket)
  {
    if constexpr (STATS)
      if (timed) { const unsigned long long now = __builtin_amdgcn_s_memtime(); bucket += now - stamp; stamp = now; }
  };

  while (true)
  {
    const int gn = g + C;
    unsigned cbn = cb + step_cb; if (cbn >= span) cbn -= span;
    if (gn < nblocks) load_block(gn, cbn, dvn, sdn);       // a whole turn ahead
    __builtin_amdgcn_sched_barrier(0);
    lap(st_rest);
    // what this wave will store for the token: its acc and the next block's number (a wave told to die stores a number
    // nobody waits for)
    const unsigned seq_out = (die && g >= C) ? 0xffffffffu : (unsigned)(g + 1);
    unsigned seq_reg = seq_out;
    asm volatile("" : "+v"(seq_reg));                       // in a vector register now, not between the token and the chain

    // ---- off the chain: the block's products into registers ----
    FD v[L];
    FD f = sd;
    relay_block<FD, L>(v, f, dv, T1, T2);
    __builtin_amdgcn_sched_barrier(0);
    // flow mode: the carry this wave stored at the end of its previous turn has long arrived (so have the loads above)
    if (owed >= 0) { flag_chunk((size_t)owed); owed = -1; }
    lap(st_prod);

    // ---- the token (a waiting wave outranks the waves that are still multiplying) ----
    __builtin_amdgcn_s_setprio(2);
    // a wave whose turn is more than one block away sleeps most of the distance (a block on the chain takes
    // >= 4.5 * L cycles): only the wave that is next polls the mailbox without pause
    {
      const raw_t peek = *my_mail;
      const int away = g - (int)uniform((int)token::seq(peek));
      if (away >= 2)
      {
        const int naps = (away - 1) * ((L * 4) / 64 > 0 ? (L * 4) / 64 : 1);     // s_sleep counts 64 cycles
        for (int i = 0; i < naps; i += 8) __builtin_amdgcn_s_sleep(8);
      }
    }
    FD acc;
    if (__builtin_expect(!relay_wait<FD>(my_mail, (unsigned)g, acc), 0))
    {
      unsigned rounds = 0;
      for (;;)
      {
        if (ring_peek(&aborted)) return;
        if (++rounds > kRelayPollCap / kRelayPollRound) { ring_abort(&aborted, a.status); return; }
        if (relay_wait<FD>(my_mail, (unsigned)g, acc)) break;
      }
    }
    // ---- on the chain: L dependent additions and the token's store, nothing else ----
    lap(st_poll);
    if constexpr (STATS) { if (stamped) tA = __builtin_amdgcn_s_memtime(); }       // (read at the end of the turn: nothing waits)
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int s = 0; s < L; ++s) acc = acc + v[s];           // sdft.h:583 / :572, in time order
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STATS) { if (stamped) tB = __builtin_amdgcn_s_memtime(); }
    *my_mail = token::pack(acc, seq_reg);
    if constexpr (STATS) { if (stamped) tC = __builtin_amdgcn_s_memtime(); }
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
    lap(st_chain);
    if constexpr (STATS)
    {
      ++st_turns;
      if (stamped && g < 1024 && lane == 0) { a.stats[64 + 3 * g] = tA; a.stats[64 + 3 * g + 1] = tB; a.stats[64 + 3 * g + 2] = tC; }
    }
    if (seq_out == 0xffffffffu) return;

    // ---- off the chain again ----
    if (!after_block(acc))
    {
      if constexpr (STATS)
      {
        if (timed && lane == 0)
        {
          a.stats[wave * 4 + 0] = st_prod; a.stats[wave * 4 + 1] = st_poll; a.stats[wave * 4 + 2] = st_chain + (st_rest << 32);
          a.stats[wave * 4 + 3] = st_turns;
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < DV; ++q) dv[q] = dvn[q];
    sd = sdn;
  }
}

// ------------------------------------------------------------------------------------------
// spectral window (sdft.h:350-402); e[] holds X[k-2] .. X[k+2] at index 0..4
// ------------------------------------------------------------------------------------------
template <typename FD, int WIN> SDFT_D cx<FD> window_tap(cx<FD> m2, cx<FD> m1, cx<FD> c0, cx<FD> p1, cx<FD> p2, FD w)
{
  if constexpr (WIN == WIN_HANN)
  {
    const cx<FD> a = cadd(c0, c0);
    const cx<FD> b = cadd(m1, p1);
    return cscale(csub(a, b), w);                       // w = weight * 0.25, formed on the host
  }
  else if constexpr (WIN == WIN_HAMMING)
  {
    const cx<FD> a = cscale(c0, (FD)(0.54));
    const cx<FD> b = cscale(cadd(m1, p1), (FD)(0.23));
    return cscale(csub(a, b), w);
  }
  else if constexpr (WIN == WIN_BLACKMAN)
  {
    const cx<FD> a = cscale(c0, (FD)(0.42));
    const cx<FD> b = cscale(cadd(m1, p1), (FD)(0.25));
    const cx<FD> d = cscale(cadd(m2, p2), (FD)(0.04));
    return cscale(cadd(csub(a, b), d), w);
  }
  else
  {
    return cscale(c0, w);
  }
}

// fused variant (see step_normal_fused); w is weight*0.25 for Hann, weight otherwise
template <typename FD, int WIN> SDFT_D cx<FD> window_tap_fused(cx<FD> m2, cx<FD> m1, cx<FD> c0, cx<FD> p1, cx<FD> p2, FD w)
{
  if constexpr (WIN == WIN_HANN)
  {
    const cx<FD> b = cadd(m1, p1);                        // ((c0+c0) - b) * w  ==  c0*(2w) - b*w
    const FD w2 = w + w;
    return cmake<FD>(__builtin_fma(c0.re, w2, -(b.re * w)), __builtin_fma(c0.im, w2, -(b.im * w)));
  }
  else if constexpr (WIN == WIN_HAMMING)
  {
    const cx<FD> b = cadd(m1, p1);
    const FD wa = (FD)(0.54) * w, wb = (FD)(0.23) * w;
    return cmake<FD>(__builtin_fma(c0.re, wa, -(b.re * wb)), __builtin_fma(c0.im, wa, -(b.im * wb)));
  }
  else if constexpr (WIN == WIN_BLACKMAN)
  {
    const cx<FD> b = cadd(m1, p1);
    const cx<FD> d = cadd(m2, p2);
    const FD wa = (FD)(0.42) * w, wb = (FD)(0.25) * w, wd = (FD)(0.04) * w;
    return cmake<FD>(__builtin_fma(d.re, wd, __builtin_fma(c0.re, wa, -(b.re * wb))),
                     __builtin_fma(d.im, wd, __builtin_fma(c0.im, wa, -(b.im * wb))));
  }
  else
  {
    return cscale(c0, w);
  }
}

// ------------------------------------------------------------------------------------------
// K1  forward: recurrence + mirror + window + coalesced store of the (n, N) matrix
// ------------------------------------------------------------------------------------------
// fid of bin kk at cursor c, rebuilt from the plan's seed table exactly as the reference would have
// rotated it since the last roll-over (sdft.h:584, unfused)
template <typename FD> SDFT_D cx<FD> fid_from_table(const cx<FD>* fseed, unsigned L, unsigned nbins, long kk, unsigned c, cx<FD> tw)
{
  cx<FD> f = fseed[(size_t)(c / L) * nbins + kk];
  for (unsigned i = c % L; i > 0; --i) f = cmul(f, tw);
  return f;
}

template <typename FD> struct ForwardArgs
{
  const FD* delta;            // [channels][n]
  const cx<FD>* tw;           // [N]
  const cx<FD>* wtab;         // [2N]   (used when seed == nullptr)
  const cx<FD>* carry;        // [channels][chunks][N]
  const cx<FD>* seed;         // [channels][chunks][N] or nullptr
  const cx<FD>* fseed;        // [2N/fseed_L][N] fid at every fseed_L-th cursor (exact mode, chain form) or nullptr
  unsigned fseed_L;
  cx<FD>* out;                // rows: out + ch*out_stride + t*N
  size_t out_stride;
  cx<FD>* const* out_rows;    // optional row-pointer table [channels*n] (sdft_sdft_nd); nullptr = dense
  cx<FD>* acc_state;          // [channels][N]  written by the last chunk
  cx<FD>* fid_state;
  size_t n;
  unsigned long long total_waves;
  unsigned nbins, chunks, chunk_len, tiles, interior_lanes, cursor0;
  unsigned chunk0, launch_chunks;   // this launch covers time chunks [chunk0, chunk0 + launch_chunks)
  unsigned chunk_shift;             // chunk j > 0 starts at sample j*chunk_len - chunk_shift (exact carries, ring form; else 0)
  int vec_store;              // BPL==2: 16-byte stores allowed (even N, 16-byte aligned base)
  FD wscale;                  // weight (or weight*0.25 for Hann)
  DoneSignal done;            // row-group kernels of short synchronous calls: total = workgroups of the launch
  // exact carries, relay form in flow mode: the carries of a chunk are ready when the ready_n words of its row all hold
  // ready_seq (written by the relay kernel, which runs beside this launch); workgroups are then numbered time-major
  const unsigned* ready;      // [channels][chunks][ready_n] or nullptr
  unsigned ready_seq, ready_n, ready_channels;
  unsigned* ready_status;     // pinned host word: a workgroup whose wait ran out adds 1 (the host re-runs the call)
  unsigned ready_status_seen; // its value when the call was launched: once it differs (a relay gave up) nobody waits on
};

// Flow mode: which (chunk, channel) a workgroup takes, and the wait for the chunk's carries.  The relay kernel stores
// carries write-through (sc1), waits for them, then stores the flag (sc1); here: relaxed agent-scope polls of the flags,
// one agent-scope acquire, then plain loads (MI355X_MICROARCH.md, inter-workgroup visibility, form R1).
template <typename FD> SDFT_D void flow_position(const ForwardArgs<FD>& a, unsigned& chunk, size_t& ch)
{
  if (a.ready) { chunk = a.chunk0 + blockIdx.x / a.ready_channels; ch = blockIdx.x % a.ready_channels; }
  else { chunk = a.chunk0 + blockIdx.x % a.launch_chunks; ch = blockIdx.x / a.launch_chunks; }
}
constexpr unsigned kFlowPollCap = 1u << 19;                // x (sleep + barrier): about half a second
template <typename FD> SDFT_D bool flow_wait(const ForwardArgs<FD>& a, unsigned chunk, size_t ch)
{
  if (!a.ready) return true;                               // workgroup-uniform
  const unsigned* row = a.ready + (ch * a.chunks + chunk) * (size_t)a.ready_n;
  for (unsigned polls = 0;; ++polls)
  {
    bool ok = true;
    for (unsigned i = threadIdx.x; i < a.ready_n; i += blockDim.x)
      ok = ok && __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ready_seq;
    if (__syncthreads_and(ok ? 1 : 0)) break;
    if (polls > kFlowPollCap)
    {
      if (threadIdx.x == 0 && a.ready_status) __hip_atomic_fetch_add(a.ready_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    // has a relay given up meanwhile?  (one lane asks -- the word lives in host memory -- and not often)
    if ((polls & 1023u) == 1023u && a.ready_status)
    {
      const bool gone = threadIdx.x == 0 && __hip_atomic_load(a.ready_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.ready_status_seen;
      if (__syncthreads_or(gone ? 1 : 0)) return false;
    }
    __builtin_amdgcn_s_sleep(32);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return true;
}

// ------------------------------------------------------------------------------------------
// Self-carried time chunks (chunk-parallel FD double path, 2N a power of two): ONE launch per call.
// The carry-in of a chunk is acc(t0) = acc(0) + sum_{t < t0} delta_t * fid(c_t) (sdft.h:583 unrolled), and with
// fid(c) = W[k*c], W[j] = exp(-2*pi*i*j/(2N)), that sum over ALL earlier samples is one 2N-point DFT of the
// differences folded by cursor:  cell[v] = sum of delta_t over the t < t0 that arrive at cursor v,
//     acc_k(t0) = acc_k(0) + sum_v cell[v] * W[k*v].
// So the workgroup of chunk j folds the call's first t0 samples into 2N LDS cells (one load per sample:
// the "old" sample of t is the "current" one of t - 2N), runs the FFT in place and has its carry-in -- no
// partial sums in memory, no scan, no dependency on any other workgroup, no launch in front of the forward
// kernel.  The differences of its own samples are formed in the time loop from scalar loads of the input and
// the delay line, as in forward_hop_kernel.  Cost per workgroup: t0 / threads loads + one FFT (a few us);
// the pre-pass it replaces was two launches, 21 us at n = 48000.  State is double-buffered like in the hop
// kernels: every workgroup reads acc(0), the last chunk's writes the new state to the other buffer.
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD> struct SelfArgs
{
  const TD* x;                // [channels][n] the call's samples; nullptr = carries and differences come from a pre-pass
  size_t x_stride;
  const TD* hist_in;          // [channels][2N] delay line in time order
  TD* hist_out;               // the other buffer: written by the workgroup of the call's last chunk
  const cx<FD>* acc_in;       // [channels][N] accumulator before the call (ForwardArgs::acc_state receives the new one)
  unsigned log2m;             // 2N = 1 << log2m, or 0: 2N = product of rl's radices (2, 3, 4, 5), Stockham between two buffers
  unsigned lds_deltas;        // fused kernel: samples of a chunk whose differences are staged in dynamic LDS (0: formed in the loop)
#ifdef SDFT_SELF_STAMPS
  unsigned long long* stamps; // development build: cycle stamps of the last chunk's workgroup (scripts/self_stamps.py)
#endif
  RadixList rl;
};

// cells[v] = sum of the differences (sdft.h:564, the subtraction in TD precision) of the samples t < t0 whose
// cursor is v; whole workgroup, no barrier inside
template <int CP, int QB, typename TD, typename FD>
SDFT_D void self_fold(const SelfArgs<TD, FD>& sa, cx<FD>* cells, unsigned m, unsigned cursor0, size_t ch, size_t t0)
{
  // A thread owns up to CP cells (cursor values v, v + threads, ...); cell v collects the samples tv, tv + 2N, ... < t0.
  // QB rows of all its cells are requested before the first is used -- CP*QB independent loads in flight, every one of
  // them unconditional (an index past the fold is clamped, its value ignored): the compiler can count them and wait once.
  // (CP*QB registers: 4 x 8 in the forward kernel, 2 x 4 in the fused one, which lives on 64 registers per lane)
  const TD* xs = sa.x + ch * sa.x_stride;
  const TD* hs = sa.hist_in + ch * (size_t)m;
  const unsigned nthr = blockDim.x;
  const size_t rows = (t0 + m - 1) / m;                    // t0 >= 1
  for (unsigned v0 = threadIdx.x; v0 < m; v0 += CP * nthr)
  {
    size_t tv[CP]; FD sum[CP]; TD prev[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c)
    {
      const unsigned v = v0 + (unsigned)c * nthr;
      tv[c] = (size_t)(((v < m ? v : v0) + m - cursor0) % m);           // first sample that arrives at cursor v (cursor0 < m)
      sum[c] = (FD)0;
      prev[c] = hs[tv[c]];                                 // x[tv - 2N]
    }
    for (size_t q = 0; q < rows; q += QB)
    {
      TD cur[CP][QB];
#pragma unroll
      for (int c = 0; c < CP; ++c)
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
        {
          const size_t t = tv[c] + (q + qq) * (size_t)m;
          cur[c][qq] = xs[t < t0 ? t : t0 - 1];
        }
#pragma unroll
      for (int c = 0; c < CP; ++c)
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
        {
          const size_t t = tv[c] + (q + qq) * (size_t)m;
          if (t < t0) { const TD dd = cur[c][qq] - prev[c]; sum[c] += (FD)dd; prev[c] = cur[c][qq]; }   // TD precision (sdft.h:564)
        }
    }
#pragma unroll
    for (int c = 0; c < CP; ++c)
    {
      const unsigned v = v0 + (unsigned)c * nthr;
      if (v < m) cells[v] = cmake<FD>(sum[c], (FD)0);
    }
  }
}

// in-place 2N-point DFT in LDS, decimation in frequency, two radix-2 stages per barrier; bin k ends in cell
// bitreverse(k); whole workgroup; ends with a barrier.  w[j] = exp(-2*pi*i*j/m), j < m.
template <typename FD>
SDFT_D void lds_fft_dif(cx<FD>* x, unsigned log2m, const cx<FD>* __restrict__ w)
{
  const unsigned m = 1u << log2m;
  unsigned st = 0;
  for (; st + 2 <= log2m; st += 2)
  {
    const unsigned half = m >> (st + 1), quarter = half >> 1;         // stage st pairs (p, p + half), stage st + 1 (p, p + quarter)
    for (unsigned i = threadIdx.x; i < (m >> 2); i += blockDim.x)
    {
      const unsigned pos = i & (quarter - 1);
      const unsigned base = ((i - pos) << 2) + pos;
      const cx<FD> a0 = x[base], a1 = x[base + quarter], a2 = x[base + half], a3 = x[base + half + quarter];
      const cx<FD> w1 = w[(size_t)pos << st];                          // stage st, pair (a0, a2)
      const cx<FD> w2 = w[(size_t)pos << (st + 1)];                    // stage st + 1, both pairs
      const cx<FD> b0 = cadd(a0, a2), b2 = cmul(csub(a0, a2), w1);
      const cx<FD> b1 = cadd(a1, a3), tq = cmul(csub(a1, a3), w1);
      const cx<FD> b3 = cmake<FD>(tq.im, -tq.re);                      // pair (a1, a3): twiddle index + m/4, i.e. times -i
      x[base] = cadd(b0, b1);
      x[base + quarter] = cmul(csub(b0, b1), w2);
      x[base + half] = cadd(b2, b3);
      x[base + half + quarter] = cmul(csub(b2, b3), w2);
    }
    __syncthreads();
  }
  if (st < log2m)                                                      // odd log2m: the last stage pairs neighbours, twiddle 1
  {
    for (unsigned i = threadIdx.x; i < (m >> 1); i += blockDim.x)
    {
      const cx<FD> p = x[2 * i], q = x[2 * i + 1];
      x[2 * i] = cadd(p, q);
      x[2 * i + 1] = csub(p, q);
    }
    __syncthreads();
  }
}

// the same DFT for 2N = product of 2, 3, 4, 5 (the reference's own test size N = 1000: 2N = 4*4*5*5*5): Stockham autosort
// between x and x + m, natural-order output, the workgroup's version of chunk_fft_mixed_kernel; returns the buffer that
// holds the result; ends with a barrier
template <typename FD>
SDFT_D cx<FD>* lds_fft_mixed(cx<FD>* x, unsigned m, const RadixList& rl, const cx<FD>* __restrict__ w)
{
  cx<FD>* y = x + m;
  unsigned ns = 1;                                         // product of the radices already applied
  for (unsigned st = 0; st < rl.count; ++st)
  {
    const unsigned r = rl.r[st];
    const unsigned nr = m / r;
    const unsigned tstep = m / (ns * r);                   // table stride of the stage twiddle
    const unsigned rstep = nr;                             // table stride of the r-th roots of unity
    for (unsigned i = threadIdx.x; i < nr; i += blockDim.x)
    {
      const unsigned k = i % ns;
      cx<FD> v[5];
#pragma unroll
      for (unsigned t = 0; t < 5; ++t)
        if (t < r)
        {
          const cx<FD> in = x[i + t * nr];
          v[t] = t == 0 ? in : cmul(in, w[(size_t)(((unsigned long long)t * k * tstep) % m)]);
        }
      const unsigned base = (i / ns) * ns * r + k;
#pragma unroll
      for (unsigned q = 0; q < 5; ++q)
        if (q < r)
        {
          cx<FD> o = v[0];
#pragma unroll
          for (unsigned t = 1; t < 5; ++t)
            if (t < r) o = cadd(o, cmul(v[t], w[(size_t)(((unsigned long long)q * t * rstep) % m)]));
          y[base + q * ns] = o;
        }
    }
    __syncthreads();
    cx<FD>* tmp = x; x = y; y = tmp;
    ns *= r;
  }
  return x;
}

// the whole prologue of a self-carried chunk: delay line for the next call (last chunk's workgroup), fold, FFT.
// Returns the buffer that holds the DFT (bin k at self_slot(k)), or nullptr: chunks that start at sample 0 need none.
// Workgroup-uniform.
template <typename TD, typename FD> SDFT_D unsigned self_slot(const SelfArgs<TD, FD>& sa, unsigned k)
{
  return sa.log2m ? (__brev(k) >> (32u - sa.log2m)) : k;
}
template <int CP, int QB, typename TD, typename FD>
SDFT_D cx<FD>* self_carry(const SelfArgs<TD, FD>& sa, const ForwardArgs<FD>& a, cx<FD>* cells, unsigned chunk, size_t ch, size_t t0)
{
  const unsigned m = 2u * a.nbins;
  if (chunk + 1 == a.chunks)
  {
    const TD* xv = sa.x + ch * sa.x_stride;
    const TD* hv = sa.hist_in + ch * (size_t)m;
    TD* ho = sa.hist_out + ch * (size_t)m;                             // element i of the last 2N samples of (hist ++ x)
    for (size_t i = threadIdx.x; i < m; i += blockDim.x)
    {
      const size_t q = a.n + i;
      ho[i] = (q >= m) ? xv[q - m] : hv[q];
    }
  }
  if (t0 == 0) return nullptr;
#ifdef SDFT_SELF_STAMPS
  const bool st_on = sa.stamps && chunk + 1 == a.chunks && threadIdx.x == 0;
  if (st_on) sa.stamps[1] = __builtin_readcyclecounter();
#endif
  self_fold<CP, QB>(sa, cells, m, a.cursor0, ch, t0);
  __syncthreads();
#ifdef SDFT_SELF_STAMPS
  if (st_on) sa.stamps[2] = __builtin_readcyclecounter();
#endif
  if (sa.log2m)
  {
    lds_fft_dif(cells, sa.log2m, a.wtab);
#ifdef SDFT_SELF_STAMPS
    if (st_on) sa.stamps[3] = __builtin_readcyclecounter();
#endif
    return cells;
  }
  return lds_fft_mixed(cells, m, sa.rl, a.wtab);
}

// differences of G consecutive samples from scalar loads of the input and the delay line (wave-uniform)
template <int G, typename TD, typename FD>
SDFT_D void self_deltas(FD (&dl)[G], const SDFT_CONSTANT TD* xs, const SDFT_CONSTANT TD* hs, size_t tt, size_t span)
{
  TD cur[G], old[G];
#pragma unroll
  for (int u = 0; u < G; ++u) cur[u] = xs[tt + u];
  if (tt + G <= span)
  {
#pragma unroll
    for (int u = 0; u < G; ++u) old[u] = hs[tt + u];
  }
  else if (tt >= span)
  {
#pragma unroll
    for (int u = 0; u < G; ++u) old[u] = xs[tt - span + u];
  }
  else
  {
#pragma unroll
    for (int u = 0; u < G; ++u) old[u] = (tt + u < span) ? hs[tt + u] : xs[tt + u - span];
  }
#pragma unroll
  for (int u = 0; u < G; ++u) { const TD dd = cur[u] - old[u]; dl[u] = (FD)dd; }     // TD precision (sdft.h:564)
}
template <typename TD, typename FD>
SDFT_D FD self_delta1(const SDFT_CONSTANT TD* xs, const SDFT_CONSTANT TD* hs, size_t tt, size_t span)
{
  const TD cur = xs[tt];
  const TD old = (tt < span) ? hs[tt] : xs[tt - span];
  const TD dd = cur - old;
  return (FD)dd;
}

// native clang vectors (the nontemporal builtin rejects HIP's struct-wrapped double2/float4)
typedef double sdft_v2f64 __attribute__((ext_vector_type(2)));
typedef float sdft_v4f32 __attribute__((ext_vector_type(4)));
typedef float sdft_v2f32 __attribute__((ext_vector_type(2)));
template <typename FD, int BPL> struct StoreVec;
template <> struct StoreVec<double, 1> { using type = sdft_v2f64; };
template <> struct StoreVec<float, 2>  { using type = sdft_v4f32; };
template <> struct StoreVec<float, 1>  { using type = sdft_v2f32; };

// (a non-temporal variant of this store was measured on MI355X: 3.205 vs 3.217 ms at n=1e6, N=1024 --
// no effect on a pure write stream -- and removed)
template <typename V> SDFT_D void store_vec(V* p, V v) { *p = v; }

template <typename FD, int BPL, int WIN, bool ROWS>
__global__ __launch_bounds__(kBlock) void forward_kernel(ForwardArgs<FD> a)
{
  constexpr int H = win_halo<WIN>::value;                 // halo bins per side
  constexpr int HL = (H + BPL - 1) / BPL;                 // halo lanes per side

  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long wave = (unsigned long long)blockIdx.x * kWavesPerBlock + wib;
  if (wave >= a.total_waves) return;

  const unsigned tile = (unsigned)(wave % a.tiles);
  const unsigned long long rest = wave / a.tiles;
  const unsigned chunk = a.chunk0 + (unsigned)(rest % a.launch_chunks);
  const size_t ch = (size_t)(rest / a.launch_chunks);

  const long nbins = (long)a.nbins;
  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  const size_t t0 = chunk ? (size_t)chunk * a.chunk_len - a.chunk_shift : 0;
  const size_t tn = (size_t)(chunk + 1) * a.chunk_len - a.chunk_shift;
  const size_t t1 = tn < a.n ? tn : a.n;
  unsigned c = (unsigned)(((size_t)a.cursor0 + t0) % span);

  // lane -> bins
  const long kfirst = (long)tile * a.interior_lanes * BPL + (long)(lane - HL) * BPL;
  const bool owner = (lane >= HL) && (lane < HL + (int)a.interior_lanes);

  BinState<FD> s[BPL];
  bool flip[BPL], live[BPL], keep[BPL];
  const size_t cbase = (ch * a.chunks + chunk) * a.nbins;
#pragma unroll
  for (int b = 0; b < BPL; ++b)
  {
    const long k = kfirst + b;
    const long kk = reflect_bin(k, nbins, flip[b]);
    live[b] = !(nbins == 1 && k != 0);                    // N == 1: halo cells are zero for ever
    keep[b] = owner && k >= 0 && k < nbins;
    s[b].tw = a.tw[kk];
    s[b].acc = a.carry[cbase + kk];
    s[b].fid = a.fseed ? fid_from_table(a.fseed, a.fseed_L, a.nbins, kk, c, s[b].tw)
             : a.seed  ? a.seed[cbase + kk] : a.wtab[(size_t)(((unsigned long long)kk * c) % span)];
  }

  const SDFT_CONSTANT FD* d = as_uniform(a.delta + ch * a.n);
  const FD w = a.wscale;
  const bool last_chunk = (chunk + 1 == a.chunks);

  // destination of this lane's first bin in row t0
  cx<FD>* dst = a.out + ch * a.out_stride + t0 * (size_t)a.nbins + kfirst;
  // ROWS: destination rows come from a pointer table (sdft_sdft_nd); kept out of the dense
  // instantiation so that its stores stay global_store_dwordx4 (a loaded pointer would force flat)
  cx<FD>* const* rows = ROWS ? a.out_rows + ch * a.n : nullptr;

  auto emit = [&](cx<FD> (&x)[BPL], size_t t)
  {
    // mirror lanes conjugate; N == 1 halo is zero
#pragma unroll
    for (int b = 0; b < BPL; ++b)
    {
      if (flip[b]) x[b].im = -x[b].im;
      if (!live[b]) x[b] = cmake<FD>((FD)0, (FD)0);
    }
    // gather X[k-2..k+2] for every bin of the lane
    cx<FD> e[BPL + 4] = {};
#pragma unroll
    for (int b = 0; b < BPL; ++b) e[b + 2] = x[b];
    if constexpr (H >= 1)
    {
      e[1] = from_below(x[BPL - 1]);             // X[k-1] of the lane's first bin
      e[BPL + 2] = from_above(x[0]);             // X[k+1] of the lane's last bin
    }
    if constexpr (H >= 2)
    {
      if constexpr (BPL >= 2)
      {
        e[0] = from_below(x[BPL - 2]);
        e[BPL + 3] = from_above(x[1]);
      }
      else
      {
        e[0] = from_below(e[1]);                 // two lanes down
        e[BPL + 3] = from_above(e[BPL + 2]);     // two lanes up
      }
    }
    cx<FD> y[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b)
      y[b] = window_tap<FD, WIN>(e[b], e[b + 1], e[b + 2], e[b + 3], e[b + 4], w);

    cx<FD>* p = dst;
    if constexpr (ROWS) p = rows[t] + kfirst;
    if constexpr (BPL == 2)
    {
      if (a.vec_store && !ROWS)
      {
        if (keep[0])
        {
          using V = typename StoreVec<FD, 2>::type;
          V v; v.x = y[0].re; v.y = y[0].im; v.z = y[1].re; v.w = y[1].im;
          store_vec(reinterpret_cast<V*>(p), v);
        }
      }
      else
      {
        if (keep[0]) p[0] = y[0];
        if (keep[1]) p[1] = y[1];
      }
    }
    else
    {
      if (keep[0])
      {
        using V = typename StoreVec<FD, 1>::type;
        V v; v.x = y[0].re; v.y = y[0].im;
        store_vec(reinterpret_cast<V*>(p), v);
      }
    }
    dst += a.nbins;
  };

  size_t t = t0;
  while (t < t1)
  {
    size_t run = maxc - c;
    if (run > t1 - t) run = t1 - t;
    size_t i = 0;
    for (; i + kGroup <= run; i += kGroup)          // one s_load burst per kGroup samples
    {
      FD dl[kGroup];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) dl[u] = d[t + i + u];
#pragma unroll
      for (int u = 0; u < kGroup; ++u)
      {
        cx<FD> x[BPL];
#pragma unroll
        for (int b = 0; b < BPL; ++b) x[b] = step_normal(s[b], dl[u]);
        emit(x, t + i + u);
      }
    }
    for (; i < run; ++i)
    {
      const FD dl = d[t + i];
      cx<FD> x[BPL];
#pragma unroll
      for (int b = 0; b < BPL; ++b) x[b] = step_normal(s[b], dl);
      emit(x, t + i);
    }
    t += run; c += (unsigned)run;
    if (t < t1)
    {
      const FD dl = d[t];
      cx<FD> x[BPL];
#pragma unroll
      for (int b = 0; b < BPL; ++b) x[b] = step_wrap(s[b], dl);
      emit(x, t);
      ++t; c = 0;
    }
  }

  if (last_chunk)
  {
#pragma unroll
    for (int b = 0; b < BPL; ++b)
      if (keep[b])
      {
        a.acc_state[ch * a.nbins + kfirst + b] = s[b].acc;
        a.fid_state[ch * a.nbins + kfirst + b] = s[b].fid;
      }
  }
}

// ------------------------------------------------------------------------------------------
// K1 (hop form)  forward for calls that are one time chunk (hop-wise streaming, SURVEY.md 8 f1:
// /root/reference/test/test.c:69-83 calls sdft_sdft_n with 100 samples per hop).  ONE launch does
// what delta_kernel + forward kernel do for long calls:
//   * the differences x[t] - x[t-2N] (sdft.h:564, TD precision) are formed in the kernel from the
//     samples and the delay line, both read over the scalar unit (wave-uniform, read-only here);
//   * a wave owns a tile of bins plus redundant halo lanes (like forward_kernel), and every wave is
//     its own workgroup, so the tiles of a row spread over as many CUs: one CU alone stores only
//     ~40 GB/s, which is what bounded the single-workgroup form (36 us per 100-sample hop);
//   * the stream state is double-buffered (read acc/fid/delay line from the current set, write the
//     other one), so no wave can observe a neighbour's new state and no copy launch is needed.
// Arithmetic is the unfused reference sequence: bit-identical to the reference for every type.
// (N == 1, where the reference's halo cells stay zero, keeps the three-launch path.)
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD> struct HopArgs
{
  const TD* x;                // [channels][n]
  size_t x_stride;
  const TD* hist_in;          // [channels][2N] delay line in time order
  TD* hist_out;
  const cx<FD>* tw;           // [N]
  const cx<FD>* acc_in;       // [channels][N]
  const cx<FD>* fid_in;
  cx<FD>* acc_out;
  cx<FD>* fid_out;
  cx<FD>* out;                // rows: out + ch*out_stride + t*N
  size_t out_stride;
  cx<FD>* const* out_rows;    // optional row-pointer table [channels*n]
  size_t n;
  unsigned long long total_waves;
  unsigned nbins, tiles, interior_lanes, cursor0;
  int vec_store;
  FD wscale;
  DoneSignal done;            // WPB == 1 launches only: total = workgroups
  unsigned long long* stamps; // development builds (-DSDFT_HOP_STAMPS): realtime stamps of workgroup 0, else nullptr
};

template <typename TD, typename FD, int BPL, int WIN, bool ROWS, int WPB>
__global__ __launch_bounds__(kWave * WPB) void forward_hop_kernel(HopArgs<TD, FD> a)
{
  constexpr int H = win_halo<WIN>::value;
  constexpr int HL = (H + BPL - 1) / BPL;

  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long wave = (unsigned long long)blockIdx.x * WPB + wib;
  if (wave >= a.total_waves) return;
  const unsigned tile = (unsigned)(wave % a.tiles);
  const size_t ch = (size_t)(wave / a.tiles);

  const long nbins = (long)a.nbins;
  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  unsigned c = a.cursor0;

  const long kfirst = (long)tile * a.interior_lanes * BPL + (long)(lane - HL) * BPL;
  const bool owner = (lane >= HL) && (lane < HL + (int)a.interior_lanes);

  BinState<FD> s[BPL];
  bool flip[BPL], keep[BPL];
  unsigned flipmask[BPL];
  const size_t sbase = ch * a.nbins;
#pragma unroll
  for (int b = 0; b < BPL; ++b)
  {
    const long k = kfirst + b;
    const long kk = reflect_bin(k, nbins, flip[b]);
    flipmask[b] = flip[b] ? 0x80000000u : 0u;
    keep[b] = owner && k >= 0 && k < nbins;
    s[b].tw = a.tw[kk];
    s[b].acc = a.acc_in[sbase + kk];
    s[b].fid = a.fid_in[sbase + kk];
  }

  // delay line for the next call: element i of the last 2N samples of (hist ++ x)
  {
    const TD* xv = a.x + ch * a.x_stride;
    const TD* hv = a.hist_in + ch * span;
    TD* ho = a.hist_out + ch * span;
    for (size_t i = (size_t)tile * kWave + lane; i < span; i += (size_t)a.tiles * kWave)
    {
      const size_t j = a.n + i;
      ho[i] = (j >= span) ? xv[j - span] : hv[j];
    }
  }

  const SDFT_CONSTANT TD* xs = as_uniform(a.x + ch * a.x_stride);
  const SDFT_CONSTANT TD* hs = as_uniform(a.hist_in + ch * span);
  const FD w = a.wscale;
  // destination = wave-uniform row base (scalar registers) + lane-constant 32-bit element offset
  cx<FD>* row = a.out + ch * a.out_stride;
  const unsigned off_bytes = (keep[0] || (BPL == 2 && keep[BPL - 1])) ? (unsigned)(kfirst < 0 ? 0 : kfirst) * (unsigned)sizeof(cx<FD>) : 0u;
  cx<FD>* const* rows = ROWS ? a.out_rows + ch * a.n : nullptr;

  auto emit = [&](cx<FD> (&x)[BPL], size_t t)
  {
#pragma unroll
    for (int b = 0; b < BPL; ++b) x[b].im = flip_sign(x[b].im, flipmask[b]);     // mirror lanes conjugate
    cx<FD> e[BPL + 4] = {};
#pragma unroll
    for (int b = 0; b < BPL; ++b) e[b + 2] = x[b];
    if constexpr (H >= 1)
    {
      e[1] = from_below_z(x[BPL - 1]);
      e[BPL + 2] = from_above_z(x[0]);
    }
    if constexpr (H >= 2)
    {
      if constexpr (BPL >= 2)
      {
        e[0] = from_below_z(x[BPL - 2]);
        e[BPL + 3] = from_above_z(x[1]);
      }
      else
      {
        e[0] = from_below_z(e[1]);
        e[BPL + 3] = from_above_z(e[BPL + 2]);
      }
    }
    cx<FD> y[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b)
      y[b] = window_tap<FD, WIN>(e[b], e[b + 1], e[b + 2], e[b + 3], e[b + 4], w);

    cx<FD>* p = reinterpret_cast<cx<FD>*>(reinterpret_cast<char*>(row) + off_bytes);
    if constexpr (ROWS) p = rows[t] + kfirst;
    if constexpr (BPL == 2)
    {
      if (a.vec_store && !ROWS)
      {
        if (keep[0])
        {
          using V = typename StoreVec<FD, 2>::type;
          V v; v.x = y[0].re; v.y = y[0].im; v.z = y[1].re; v.w = y[1].im;
          store_vec(reinterpret_cast<V*>(p), v);
        }
      }
      else
      {
        if (keep[0]) p[0] = y[0];
        if (keep[1]) p[1] = y[1];
      }
    }
    else
    {
      if (keep[0])
      {
        using V = typename StoreVec<FD, 1>::type;
        V v; v.x = y[0].re; v.y = y[0].im;
        store_vec(reinterpret_cast<V*>(p), v);
      }
    }
    row += a.nbins;
  };

  size_t t = 0;
  while (t < a.n)
  {
    size_t run = maxc - c;
    if (run > a.n - t) run = a.n - t;
    size_t i = 0;
    for (; i + kGroup <= run; i += kGroup)
    {
      // differences of kGroup samples (sdft.h:564): the old sample comes from the delay line while
      // t < 2N, from the call's own input afterwards
      const size_t tt = t + i;
      TD cur[kGroup], old[kGroup];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) cur[u] = xs[tt + u];
      if (tt + kGroup <= span)
      {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) old[u] = hs[tt + u];
      }
      else if (tt >= span)
      {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) old[u] = xs[tt - span + u];
      }
      else
      {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) old[u] = (tt + u < span) ? hs[tt + u] : xs[tt + u - span];
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u)
      {
        const TD dd = cur[u] - old[u];                    // TD precision
        const FD dl = (FD)dd;
        cx<FD> x[BPL];
#pragma unroll
        for (int b = 0; b < BPL; ++b) x[b] = step_normal(s[b], dl);
        emit(x, tt + u);
      }
    }
    for (; i <= run && t + i < a.n; ++i)                  // tail of the run, then the roll-over step
    {
      const size_t tt = t + i;
      const TD cur = xs[tt];
      const TD old = (tt < span) ? hs[tt] : xs[tt - span];
      const TD dd = cur - old;
      const FD dl = (FD)dd;
      cx<FD> x[BPL];
      if (i < run)
      {
#pragma unroll
        for (int b = 0; b < BPL; ++b) x[b] = step_normal(s[b], dl);
      }
      else
      {
#pragma unroll
        for (int b = 0; b < BPL; ++b) x[b] = step_wrap(s[b], dl);
      }
      emit(x, tt);
    }
    if (t + run < a.n) { t += run + 1; c = 0; }            // the roll-over step was taken
    else { t += run; c += (unsigned)run; }
  }

#pragma unroll
  for (int b = 0; b < BPL; ++b)
    if (keep[b])
    {
      a.acc_out[sbase + kfirst + b] = s[b].acc;
      a.fid_out[sbase + kfirst + b] = s[b].fid;
    }
  if constexpr (WPB == 1) { if (lane == 0) signal_done(a.done); }
}

// ------------------------------------------------------------------------------------------
// K1h, two waves per tile (small launches).  A lone wave pays 5-8 cycles per fp64 instruction whatever its
// dependencies are, so a tile's 38 instructions per sample are split between two waves on two SIMDs of the CU:
// wave 0 runs the recurrence (16 instructions), conjugates the mirror lanes and parks the demodulated bins of an
// 8-sample group in LDS; wave 1 takes the group one barrier later, reads each lane's window neighbours straight
// from that image (three 16-byte reads instead of eight DPP moves per neighbour pair), applies the window and
// stores the rows.  Double-buffered image, one s_barrier per group; same operations on the same operands as
// forward_hop_kernel, bit for bit.  Differences: staged in LDS by one round of vector loads, as in
// process_hop_kernel (calls of one time chunk are shorter than kHopMax samples; longer ones keep the one-wave form).
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD, int BPL, int WIN, bool ROWS>
__global__ __launch_bounds__(2 * kWave) void forward_hop2_kernel(HopArgs<TD, FD> a)
{
  constexpr int H = win_halo<WIN>::value;
  constexpr int HL = (H + BPL - 1) / BPL;
  constexpr int G = kGroup;
  __shared__ cx<FD> image[2][G][BPL][kWave];               // [buffer][sample of the group][bin of the lane][lane]
  __shared__ TD diff_lds[kHopMax + G];

#ifdef SDFT_HOP_STAMPS
  unsigned long long stamp[4]; stamp[0] = __builtin_amdgcn_s_memrealtime();
#define SDFT_HOP2_STAMP(i) stamp[i] = __builtin_amdgcn_s_memrealtime()
#else
#define SDFT_HOP2_STAMP(i)
#endif
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0 recurrence, 1 window + stores
  const unsigned tile = blockIdx.x % a.tiles;
  const size_t ch = blockIdx.x / a.tiles;

  const long nbins = (long)a.nbins;
  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  const long kfirst = (long)tile * a.interior_lanes * BPL + (long)(lane - HL) * BPL;
  const bool owner = (lane >= HL) && (lane < HL + (int)a.interior_lanes);
  const size_t sbase = ch * a.nbins;

  // delay line for the next call (both waves share the copy): element i of the last 2N samples of (hist ++ x)
  {
    const TD* xv = a.x + ch * a.x_stride;
    const TD* hv = a.hist_in + ch * span;
    TD* ho = a.hist_out + ch * span;
    for (size_t i = (size_t)tile * (2 * kWave) + threadIdx.x; i < span; i += (size_t)a.tiles * (2 * kWave))
    {
      const size_t j = a.n + i;
      ho[i] = (j >= span) ? xv[j - span] : hv[j];
    }
    // differences of the whole call (sdft.h:564), the subtraction in TD precision
    for (size_t tt = threadIdx.x; tt < a.n; tt += 2 * kWave)
    {
      const TD cur = xv[tt];
      const TD old = (tt < span) ? hv[tt] : xv[tt - span];
      diff_lds[tt] = cur - old;
    }
  }

  const size_t groups = (a.n + G - 1) / G;
  if (role == 0)
  {
    // ---------------- recurrence ----------------
    BinState<FD> s[BPL];
    unsigned flipmask[BPL];
    bool keep[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b)
    {
      bool flip;
      const long k = kfirst + b;
      const long kk = reflect_bin(k, nbins, flip);
      flipmask[b] = flip ? 0x80000000u : 0u;
      keep[b] = owner && k >= 0 && k < nbins;
      s[b].tw = a.tw[kk];
      s[b].acc = a.acc_in[sbase + kk];
      s[b].fid = a.fid_in[sbase + kk];
    }
    __syncthreads();                                         // the differences are staged
    SDFT_HOP2_STAMP(1);
    unsigned c = a.cursor0;
    int buf = 0;
    for (size_t g = 0; g < groups; ++g)
    {
      const size_t t = g * G;
      const int m = (a.n - t < (size_t)G) ? (int)(a.n - t) : G;
      TD dd[G];
#pragma unroll
      for (int u = 0; u < G; ++u) dd[u] = diff_lds[t + u];   // broadcast reads (cells past n were never written: unused)
      if (m == G && c + G <= maxc)
      {
        // a whole group without the roll-over: no per-sample decisions
#pragma unroll
        for (int u = 0; u < G; ++u)
        {
          const FD dl = (FD)dd[u];
#pragma unroll
          for (int b = 0; b < BPL; ++b)
          {
            cx<FD> x = step_normal(s[b], dl);
            x.im = flip_sign(x.im, flipmask[b]);             // mirror lanes conjugate
            image[buf][u][b][lane] = x;
          }
        }
        c += G;
      }
      else
      {
#pragma unroll
        for (int u = 0; u < G; ++u)
        {
          if (u < m)
          {
            const FD dl = (FD)dd[u];
            const bool wrap = (c == maxc);                   // wave-uniform
#pragma unroll
            for (int b = 0; b < BPL; ++b)
            {
              cx<FD> x;
              if (wrap) x = step_wrap(s[b], dl); else x = step_normal(s[b], dl);
              x.im = flip_sign(x.im, flipmask[b]);
              image[buf][u][b][lane] = x;
            }
            c = wrap ? 0 : c + 1;
          }
        }
      }
      __syncthreads();                                       // group g is in the image
      buf ^= 1;
    }
    SDFT_HOP2_STAMP(2);
#pragma unroll
    for (int b = 0; b < BPL; ++b)
      if (keep[b])
      {
        a.acc_out[sbase + kfirst + b] = s[b].acc;
        a.fid_out[sbase + kfirst + b] = s[b].fid;
      }
#ifdef SDFT_HOP_STAMPS
    if (a.stamps && blockIdx.x == 0 && lane == 0) for (int i = 0; i < 3; ++i) a.stamps[i] = stamp[i];
#endif
  }
  else
  {
    // ---------------- window + stores ----------------
    bool keep[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b) { const long k = kfirst + b; keep[b] = owner && k >= 0 && k < nbins; }
    const FD w = a.wscale;
    cx<FD>* row = a.out + ch * a.out_stride;
    const unsigned off_bytes = (keep[0] || (BPL == 2 && keep[BPL - 1])) ? (unsigned)(kfirst < 0 ? 0 : kfirst) * (unsigned)sizeof(cx<FD>) : 0u;
    cx<FD>* const* rows = ROWS ? a.out_rows + ch * a.n : nullptr;
    // neighbour lanes, clamped: lanes that would read outside the wave own no bins (their rows are not stored)
    const int lb1 = lane >= 1 ? lane - 1 : 0, lb2 = lane >= 2 ? lane - 2 : 0;
    const int la1 = lane <= kWave - 2 ? lane + 1 : kWave - 1, la2 = lane <= kWave - 3 ? lane + 2 : kWave - 1;
    __syncthreads();                                         // (pairs with the barrier after the staging)
    SDFT_HOP2_STAMP(1);
    int buf = 0;
    for (size_t g = 0; g < groups; ++g)
    {
      const size_t t = g * G;
      const int m = (a.n - t < (size_t)G) ? (int)(a.n - t) : G;
      __syncthreads();                                       // group g is in the image
      // every read of the group is requested before the first sample is windowed (a lone wave has nothing else to
      // put into an LDS round trip; rows past the call's end hold stale bins and are not stored)
      cx<FD> eg[G][BPL + 4];
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
#pragma unroll
        for (int i = 0; i < BPL + 4; ++i) eg[u][i] = cmake<FD>((FD)0, (FD)0);
#pragma unroll
        for (int b = 0; b < BPL; ++b) eg[u][b + 2] = image[buf][u][b][lane];
        if constexpr (H >= 1)
        {
          eg[u][1] = image[buf][u][BPL - 1][lb1];
          eg[u][BPL + 2] = image[buf][u][0][la1];
        }
        if constexpr (H >= 2)
        {
          if constexpr (BPL >= 2)
          {
            eg[u][0] = image[buf][u][BPL - 2][lb1];
            eg[u][BPL + 3] = image[buf][u][1][la1];
          }
          else
          {
            eg[u][0] = image[buf][u][0][lb2];
            eg[u][BPL + 3] = image[buf][u][0][la2];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
        if (m == G || u < m)
        {
          const cx<FD> (&e)[BPL + 4] = eg[u];
          cx<FD> y[BPL];
#pragma unroll
          for (int b = 0; b < BPL; ++b)
            y[b] = window_tap<FD, WIN>(e[b], e[b + 1], e[b + 2], e[b + 3], e[b + 4], w);

          cx<FD>* p = reinterpret_cast<cx<FD>*>(reinterpret_cast<char*>(row) + off_bytes);
          if constexpr (ROWS) p = rows[t + u] + kfirst;
          if constexpr (BPL == 2)
          {
            if (a.vec_store && !ROWS)
            {
              if (keep[0])
              {
                using V = typename StoreVec<FD, 2>::type;
                V v; v.x = y[0].re; v.y = y[0].im; v.z = y[1].re; v.w = y[1].im;
                store_vec(reinterpret_cast<V*>(p), v);
              }
            }
            else
            {
              if (keep[0]) p[0] = y[0];
              if (keep[1]) p[1] = y[1];
            }
          }
          else
          {
            if (keep[0])
            {
              using V = typename StoreVec<FD, 1>::type;
              V v; v.x = y[0].re; v.y = y[0].im;
              store_vec(reinterpret_cast<V*>(p), v);
            }
          }
          row += a.nbins;
        }
      }
      buf ^= 1;
    }
#ifdef SDFT_HOP_STAMPS
    SDFT_HOP2_STAMP(2);
    if (a.stamps && blockIdx.x == 0 && lane == 0) for (int i = 0; i < 3; ++i) a.stamps[4 + i] = stamp[i];
#endif
  }
  // completion word: both waves' stores are out before one lane reports
  signal_done_workgroup(a.done);
}

// ------------------------------------------------------------------------------------------
// Spectral operation between analysis and synthesis (fused path): what a host of the
// reference does to the (n, N) matrix between sdft_sdft_n and sdft_isdft_n (README.md:42-47),
// applied per bin while the row is in flight.
//   identity            X'_k = X_k
//   gain  g[N] (real)   X'_k = X_k * g_k          (complex times real: both parts scaled)
//   shift s (bins)      X'_k = X_{k-s}, zero where k-s falls outside [0, N)
// synth_term returns what sdft_isdft adds for SOURCE bin k (sdft.h:643 / :650).  The reference adds
// output bins in ascending order; a shift keeps source bins in the same order, and the bins it
// empties add +-0, which never changes a running sum that started at +0.
// ------------------------------------------------------------------------------------------
//   gate  (thr, floor)  X'_k = X_k if |X_k| >= thr, else X_k * floor            (not linear: windowed rows only)
//   power (p, scale)    X'_k = X_k * scale * |X_k|^(p-1), i.e. |X'_k| = scale * |X_k|^p with the phase kept
// Gains may change with time: `rows` gain vectors, row r for the call's samples [r*hop, (r+1)*hop), the last one for
// everything after it (what a host does when it recomputes its mask every hop; README.md:42-47 leaves that loop to it).
//   user  (expression)   X'_k = whatever the host's statements leave in (re, im): compiled at run time (user_op below)
enum : int { OP_IDENTITY = 0, OP_GAIN = 1, OP_SHIFT = 2, OP_CGAIN = 3, OP_GATE = 4, OP_POWER = 5, OP_USER = 6 };
template <typename FD> struct SpectralOp
{
  int kind;
  const FD* gain;             // OP_GAIN: [rows][N] real factors; OP_CGAIN: [rows][N] complex factors (re, im interleaved)
  long shift;                 // OP_SHIFT
  unsigned rows;              // gain vectors (<= 1: one for the whole call)
  size_t hop;                 // samples per gain vector
  size_t t0;                  // index, within the host's call, of the first row a launch sees (two-pass segments)
  FD p0, p1;                  // OP_GATE: threshold, floor; OP_POWER: exponent, scale
  FD pv[8];                   // OP_USER: up to eight parameters travel with the kernel arguments (more: `gain` points at them)
};
template <typename FD> SDFT_HD bool op_is_linear(int kind) { return kind <= OP_CGAIN; }
// the operation a kernel serves: the library's own build dispatches on SpectralOp::kind at run time; a run-time
// compilation (the host's statements) is for one operation, and every other branch leaves the code
#ifndef SDFT_FIXED_OP
#define SDFT_FIXED_OP -1
#endif
template <typename FD> SDFT_D int op_kind_of(const SpectralOp<FD>& op) { return SDFT_FIXED_OP >= 0 ? SDFT_FIXED_OP : op.kind; }
template <int V> struct OpTag { static constexpr int value = V; };   // an operation known where the code is generated (-1: not)
// the gain vector of row t of the launch
template <typename FD> SDFT_D const FD* gain_row(const SpectralOp<FD>& op, size_t t, unsigned nbins)
{
  if (op.rows <= 1 || !op.gain) return op.gain;
  size_t r = (op.t0 + t) / op.hop;
  if (r >= op.rows) r = op.rows - 1;
  return op.gain + r * (size_t)nbins * (op.kind == OP_CGAIN ? 2u : 1u);
}
// ... walked forward in time (the row-group kernels): one division at the start, additions afterwards
template <typename FD> struct GainCursor
{
  const FD* g; size_t next, hop, stride; unsigned left;     // next: launch-relative time at which the next vector starts
  SDFT_D void start(const SpectralOp<FD>& op, size_t t, unsigned nbins)
  {
    g = op.gain; next = ~(size_t)0; hop = op.hop; left = 0; stride = (size_t)nbins * (op.kind == OP_CGAIN ? 2u : 1u);
    if (op.rows <= 1 || !op.gain || (op.kind != OP_GAIN && op.kind != OP_CGAIN)) return;
    size_t r = (op.t0 + t) / op.hop;
    if (r >= op.rows) r = op.rows - 1;
    g = op.gain + r * stride;
    left = op.rows - 1 - (unsigned)r;
    if (left) next = (r + 1) * op.hop - op.t0;
  }
  SDFT_D void seek(size_t t)                               // t never decreases
  {
    while (left && t >= next) { g += stride; --left; next = left ? next + hop : ~(size_t)0; }
  }
};
// x^h for a positive, finite, normal double x: exp(h * ln x) with both functions written out -- ln x = e*ln2 + 2*atanh(z),
// z = (r - 1)/(r + 1) for the mantissa r in [sqrt(1/2), sqrt(2)), a polynomial of degree 10 in z^2; exp by k = rint(t/ln2),
// a Taylor polynomial of degree 13 on |s| <= ln2/2 and one v_ldexp_f64.  About 50 fp64 instructions and a dozen registers
// (the library's log and exp, which also serve arguments this caller never has, take three times both: the power law at
// N = 2048, where the kernel has no registers to spare, 18.8 -> x ms).  Relative error of the result: 2e-16 * (1 + |h ln x|).
// a double constant in a scalar register pair at the point of use (two s_mov_b32): left to itself the compiler keeps the 25
// polynomial coefficients below in 50 vector registers for the whole kernel -- and spills them
SDFT_D double scalar_const(double c) { asm volatile("" : "+s"(c)); return c; }
SDFT_D double pow_positive(double x, double h)
{
  const long long bits = __double_as_longlong(x);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  double r = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);      // [1, 2)
  if (r > 1.4142135623730951) { r *= 0.5; ++e; }
  // (no IEEE division: v_rcp_f64 and two Newton steps -- the divide expansion costs 15 instructions and two mode switches)
  const double den = r + 1.0;
  double inv = __builtin_amdgcn_rcp(den);
  inv = __builtin_fma(__builtin_fma(-den, inv, 1.0), inv, inv);
  inv = __builtin_fma(__builtin_fma(-den, inv, 1.0), inv, inv);
  const double z = (r - 1.0) * inv, w = z * z;
  double q = scalar_const(1.0 / 21.0);
  q = __builtin_fma(q, w, scalar_const(1.0 / 19.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 17.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 15.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 13.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 11.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 9.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 7.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 5.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 3.0)); q = __builtin_fma(q, w, 1.0);
  const double ln = __builtin_fma((double)e, scalar_const(0.6931471805599453), 2.0 * z * q);
  const double t = h * ln;
  if (t > 709.0) return __builtin_huge_val();
  if (t < -745.0) return 0.0;
  const double k = __builtin_rint(t * scalar_const(1.4426950408889634));
  const double sred = __builtin_fma(-k, scalar_const(1.9082149292705877e-10), __builtin_fma(-k, scalar_const(0.6931471803691238), t));   // ln2 = hi + lo
  double p = scalar_const(1.0 / 6227020800.0);
  p = __builtin_fma(p, sred, scalar_const(1.0 / 479001600.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 39916800.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 3628800.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 362880.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 40320.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 5040.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 720.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 120.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 24.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 6.0));
  p = __builtin_fma(p, sred, 0.5); p = __builtin_fma(p, sred, 1.0); p = __builtin_fma(p, sred, 1.0);
  return __builtin_ldexp(p, (int)k);
}

// the operations that are not linear in the spectrum, on one windowed bin
template <typename FD> SDFT_D cx<FD> op_pointwise(cx<FD> v, const SpectralOp<FD>& op, int kind)
{
  if (kind == OP_GATE)
  {
    const FD mag2 = v.re * v.re + v.im * v.im;
    return (mag2 < op.p0 * op.p0) ? cscale(v, op.p1) : v;
  }
  if (kind == OP_POWER)
  {
    const FD mag2 = v.re * v.re + v.im * v.im;
    // (FD float: |v| below 1e-19 -- a denormal square, which v_log_f32 would flush -- counts as zero)
    // (FD double: a square below the smallest normal double likewise -- |v| < 1.5e-154)
    if (!(mag2 > (sizeof(FD) == 8 ? (FD)2.2250738585072014e-308 : (FD)1.17549435e-38f))) return cmake<FD>((FD)0, (FD)0);
    if (!(mag2 < (FD)__builtin_huge_val())) return v;                                    // infinities and NaNs pass through
    // |v|^(p-1) = exp((p-1)/2 * ln |v|^2): mag2 is positive and finite here, so none of pow()'s case analysis is needed
    // (a third of its instructions and registers; 1e-15 / 1e-6 of the factor at FD double / float, the float one
    // through v_log_f32 / v_exp_f32)
    const FD h = (op.p0 - (FD)1) * (FD)0.5;
    FD f;
    if constexpr (sizeof(FD) == 8) f = op.p1 * pow_positive(mag2, h);
    else f = op.p1 * __builtin_amdgcn_exp2f(h * __builtin_amdgcn_logf(mag2));
    return cscale(v, f);
  }
  return v;
}

// the host's own operation (sdft_hip_process_n with sdft_hip_op_expr): its statements are the text of the header "sdft_user_expr.inc" of the
// run-time compilation, which defines SDFT_USER_EXPR; the library's own build has no such operation.
// In scope: re, im (sdft_fd_t, read and assign: the windowed value of bin k), k, nbins (unsigned), t (size_t: sample index
// within the call), ch (size_t: channel), p (const sdft_fd_t*: the call's parameters, device memory), and HIP's math.
#ifdef SDFT_USER_EXPR
// p[i]: the call's parameters -- out of the kernel arguments (up to eight: no copy, no launch in front of the kernel; a
// pageable 8-byte hipMemcpyAsync in front of every hop cost a synchronous host 110 us) or out of device memory through
// the constant address space (scalar loads that nothing in the kernel can alias, so they leave the loop)
template <typename FD> struct UserParams
{
  const FD* small; const SDFT_CONSTANT FD* big;
  SDFT_D FD operator[](size_t i) const { return big ? big[i] : small[i]; }
};
template <typename FD> SDFT_D cx<FD> user_op(cx<FD> v, unsigned k, unsigned nbins, size_t t, size_t ch, const SpectralOp<FD>& op)
{
  const UserParams<FD> p{op.pv, op.gain ? as_uniform(op.gain) : nullptr};
  typedef FD sdft_fd_t;
  FD re = v.re, im = v.im;
  {
#include "sdft_user_expr.inc"
  }
  return cmake<FD>(re, im);
}
// rows[ch][t][k] = user_op(rows[ch][t][k]): the two-pass route (rows that no workgroup holds, one-chunk calls)
template <typename FD>
__global__ __launch_bounds__(256) void user_rows_kernel(cx<FD>* mat, size_t stride, size_t rows, unsigned nbins, unsigned channels, SpectralOp<FD> op)
{
  const size_t per = rows * nbins, total = per * channels;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
  {
    const size_t ch = i / per, r = i - ch * per;
    const size_t t = r / nbins, k = r - t * nbins;
    cx<FD>* q = mat + ch * stride + r;
    *q = user_op(*q, (unsigned)k, nbins, op.t0 + t, ch, op);
  }
}
#endif

// grow: the gain vector of the row v belongs to (gain_row / GainCursor); unused by the other operations
template <typename FD, bool LAT1, bool OPS>
SDFT_D FD synth_term(cx<FD> v, unsigned k, const SpectralOp<FD>& op, const cx<FD>* syn, unsigned nbins, const FD* grow = nullptr)
{
  long ko = (long)k;                                       // output bin whose sign / twiddle applies
  if constexpr (OPS)                                       // (plain sdft_isdft_n instantiates without the checks)
  {
    if (op.kind == OP_GAIN) v = cscale(v, grow[k < nbins ? k : 0]);
    else if (op.kind == OP_CGAIN) v = cmul(v, reinterpret_cast<const cx<FD>*>(grow)[k < nbins ? k : 0]);
    else if (op.kind == OP_SHIFT)
    {
      ko += op.shift;
      if (ko < 0 || ko >= (long)nbins) return (FD)0;
    }
    else if (op.kind >= OP_GATE) v = op_pointwise(v, op, op.kind);
  }
  if constexpr (LAT1) return v.re * ((ko & 1) ? (FD)(-1) : (FD)(+1));               // sdft.h:643
  else { const cx<FD> sy = syn[ko < (long)nbins ? ko : 0]; return v.re * sy.re - v.im * sy.im; }   // re of :650
}

// Fused analysis -> operation -> synthesis (forward_rows_kernel with SYN != 0): the rows never leave the
// workgroup unless `store` asks for a copy of the processed spectrum.
template <typename TD, typename FD> struct FuseArgs
{
  TD* y;                      // [channels][n]
  size_t y_stride;
  const cx<FD>* syn;          // [N]
  FD sweight;
  SpectralOp<FD> op;
  int store;                  // also write the processed rows to ForwardArgs::out
  unsigned* walked;           // SYN = 2, float samples: counts the samples whose sum had to be walked in order (or nullptr)
};

// ------------------------------------------------------------------------------------------
// K1 (row-group form)  forward for rows that fit one workgroup: 8 <= N <= 1024*BPL*S bins.
//
// One workgroup = all bins of one (channel, time chunk): wave w owns bins [64*BPL*w, 64*BPL*(w+1)),
// every lane owns BPL adjacent bins; there are no halo lanes and no redundant recurrences (lanes
// past bin N-1 in a partial last wave run the mirrored bins, as in forward_kernel, so that the
// in-wave shifts see the right neighbours).  The waves advance in lockstep, kRowGroup samples at a
// time:
//   phase A  recurrence for kRowGroup samples; the demodulated bins stay in registers; the bins a
//            neighbouring wave needs -- or, at the two ends of the spectrum, their conjugate
//            mirror images (sdft.h:589-595) -- are published to LDS edge slots by the few lanes
//            that own them (exec-masked ds_write, scalar bookkeeping only);
//   barrier  one per group (the edge slots are double-buffered);
//   phase B  window: neighbours by DPP whole-wave shifts whose fill value (what lane 0 / lane 63
//            receive) is the neighbouring wave's edge bin read from LDS by a broadcast ds_read --
//            no selects; then every wave stores its 1 KiB of the row.  The group writes whole
//            rows back to back, the store stream HBM likes best (store-only kernel: 5.85 TB/s vs
//            5.5 TB/s for independent tiles).
// FUSED selects fused-multiply-add arithmetic (option "fused", chunk-parallel FD double path only).
// ------------------------------------------------------------------------------------------
constexpr int kRowWavesMax = 16;
#ifndef SDFT_ROW_GROUP
#define SDFT_ROW_GROUP 8
#endif
#ifndef SDFT_SYN_GROUP_S2F
#define SDFT_SYN_GROUP_S2F 2
#endif
constexpr int kRowGroup = SDFT_ROW_GROUP;                 // samples per lockstep group (one barrier each)

// Rows longer than 1024*BPL bins: every lane owns S "slots"; slot q of physical wave w is the
// virtual wave v = q*nwaves + w, which covers bins [64*BPL*v, 64*BPL*(v+1)).  Edge slots in LDS
// are indexed by virtual wave, so slot boundaries are crossed exactly like wave boundaries.  The
// lockstep group shrinks to kRowGroup/S samples so that registers and LDS stay constant.
constexpr int kRowSlotsMax = 2;      // 4 slots spill at the 128-VGPR cap of a 16-wave group
// samples per lockstep group of the fused synthesis path (the plan sizes the terms image with it)
#ifndef SDFT_SYN_GROUP_S2D
#define SDFT_SYN_GROUP_S2D 4
#endif
#ifndef SDFT_SYN_GROUP_TREE
#define SDFT_SYN_GROUP_TREE 8
#endif
constexpr int syn_group(int S, int BPL, int SYN)
{
  return (S == 2 && BPL == 2 && SYN == 1) ? SDFT_SYN_GROUP_S2F : (SYN == 1 && S == 1) ? SDFT_SYN_GROUP_TREE
       : (S == 2 && BPL == 1) ? SDFT_SYN_GROUP_S2D : kRowGroup / S;
}

// SYN (fused analysis -> operation -> synthesis): 0 = rows are stored (the
// plain forward kernel), 1 = the row is turned into the terms sdft_isdft adds (sdft.h:641-651), parked in
// LDS and summed over bins by a wave-parallel tree, 2 = summed strictly in ascending bin order like the
// reference (lane u of wave 0 walks sample u's terms: bit-identical to sdft_sdft_n + sdft_isdft_n, at the
// price of N dependent additions per lockstep group).  The matrix is written only if FuseArgs::store.
// SELF: self-carried chunks (see SelfArgs): no pre-pass, the workgroup derives its carry-in and its differences itself.
template <typename FD, int BPL, int WIN, bool FUSED, int S, int SYN = 0, bool LAT1 = true, typename TD = float, bool SELF = false>
__global__ __launch_bounds__(kWave * kRowWavesMax) void forward_rows_kernel(ForwardArgs<FD> a, FuseArgs<TD, FD> fz, SelfArgs<TD, FD> sa)
{
  static_assert(!SELF || SYN == 0, "the self-carried form shares the dynamic LDS with the terms image");
  constexpr int H = win_halo<WIN>::value;
  // keeps registers roughly constant; the fused synthesis path takes eight samples per group whatever
  // BPL is (its per-group cost is the walk over the bins, shared by as many lanes as there are samples)
  // and four with two slots per lane (the double-buffered terms image of 2 x 4 padded rows of 2048 cx<double>
  // / 4096 cx<float> bins is 128 KiB of LDS)
  constexpr int G = SYN != 0 ? syn_group(S, BPL, SYN) : ((kRowGroup / (S * BPL)) >= 2 ? kRowGroup / (S * BPL) : 2);
  constexpr int HS = 2;                                   // edge slots per side (H <= 2)
  constexpr int VW = kRowWavesMax * S;                    // virtual waves
  // edgeL[buf][u][v][i] = bin (first bin of virtual wave v) - 1 - i, edgeR[..][i] = (last bin) + 1 + i
  __shared__ cx<FD> edgeL[2][G][VW][HS];
  __shared__ cx<FD> edgeR[2][G][VW][HS];
  // SYN: terms[u][bin], one padded row per sample of the lockstep group (dynamic LDS; the pad of one
  // 16-byte vector puts the G rows on different banks for the ordered walk)
  extern __shared__ __align__(16) unsigned char rows_dyn_lds[];
  FD* terms = reinterpret_cast<FD*>(rows_dyn_lds);

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int nv = nwaves * S;
  unsigned chunk; size_t ch;
  flow_position(a, chunk, ch);
#ifdef SDFT_SELF_STAMPS
  if constexpr (SELF) { if (sa.stamps && chunk + 1 == a.chunks && threadIdx.x == 0) sa.stamps[0] = __builtin_readcyclecounter(); }
#endif
  if (!flow_wait(a, chunk, ch)) return;                    // flow mode: the chunk's carries (a time-out ends the workgroup)

  const long nbins = (long)a.nbins;
  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  const size_t t0 = chunk ? (size_t)chunk * a.chunk_len - a.chunk_shift : 0;
  const size_t tn = (size_t)(chunk + 1) * a.chunk_len - a.chunk_shift;
  const size_t t1 = tn < a.n ? tn : a.n;
  unsigned c = (unsigned)(((size_t)a.cursor0 + t0) % span);

  // SELF: carry-in by fold + FFT of everything before this chunk (dynamic LDS: 2N cells)
  cx<FD>* cells = reinterpret_cast<cx<FD>*>(rows_dyn_lds);
  cx<FD>* dft = nullptr;                                   // the chunk's carry-in minus acc(0), bin k at self_slot(k)
  if constexpr (SELF) dft = self_carry<2, 16>(sa, a, cells, chunk, ch, t0);

  const long vlast_bin = (long)nv * kWave * BPL - 1;      // last (possibly virtual) bin of the group
  BinState<FD> s[S][BPL];
  bool keep[S][BPL], flip[S][BPL];
  // publishing role of each owned bin: LDS destination and whether the published value is conjugated
  cx<FD>* pub[S][BPL];
  bool pubflip[S][BPL], has_role[S][BPL];
  unsigned flipmask[S][BPL], pubmask[S][BPL];            // sign-bit masks: conjugate on use / on publish
  const size_t cbase = (ch * a.chunks + chunk) * a.nbins;
#pragma unroll
  for (int q = 0; q < S; ++q)
  {
    const int v = q * nwaves + wave;
    const long wfirst = (long)v * kWave * BPL;            // first bin of this virtual wave
    const long wlast = wfirst + (long)kWave * BPL - 1;
#pragma unroll
    for (int b = 0; b < BPL; ++b)
    {
      const long k = wfirst + (long)lane * BPL + b;
      const long kk = reflect_bin(k, nbins, flip[q][b]);
      keep[q][b] = k < nbins;
      s[q][b].tw = a.tw[kk];
      if constexpr (SELF)
      {
        s[q][b].acc = sa.acc_in[ch * a.nbins + kk];
        if (dft) s[q][b].acc = cadd(s[q][b].acc, dft[self_slot(sa, (unsigned)kk)]);
        s[q][b].fid = a.wtab[(size_t)(((unsigned long long)kk * c) % span)];
      }
      else
      {
      s[q][b].acc = a.carry[cbase + kk];
      s[q][b].fid = a.fseed ? fid_from_table(a.fseed, a.fseed_L, a.nbins, kk, c, s[q][b].tw)
                  : a.seed  ? a.seed[cbase + kk] : a.wtab[(size_t)(((unsigned long long)kk * c) % span)];
      }

      pub[q][b] = &edgeL[0][0][0][0];
      pubflip[q][b] = false; has_role[q][b] = false;
      if (H >= 1)
      {
        // neighbour roles hold for real bins and for in-group mirror lanes alike (a row may end
        // one bin into a virtual wave: its neighbour still needs two bins from it)
#pragma unroll
        for (int i = 0; i < HS; ++i)
        {
          // next virtual wave's left edge: bins wlast, wlast-1
          if (v + 1 < nv && k == wlast - i) { pub[q][b] = &edgeL[0][0][v + 1][i]; has_role[q][b] = true; }
          // previous virtual wave's right edge: bins wfirst, wfirst+1
          if (v > 0 && k == wfirst + i) { pub[q][b] = &edgeR[0][0][v - 1][i]; has_role[q][b] = true; }
        }
#pragma unroll
        for (int i = 0; i < HS; ++i)
        {
          // spectrum ends: mirror images of the virtual bins -1-i and vlast_bin+1+i
          bool f0; const long r0 = reflect_bin(-1 - i, nbins, f0);
          if (k == r0) { pub[q][b] = &edgeL[0][0][0][i]; pubflip[q][b] = f0; has_role[q][b] = true; }
          // (the right-hand images are consumed only if the group's last lanes own real bins, i.e.
          // fewer than H virtual bins follow bin N-1; otherwise in-wave mirror lanes serve them and a
          // bin must not lose its other role to a publish nobody reads)
          if (vlast_bin - (nbins - 1) < H)
          {
            bool f1; const long r1 = reflect_bin(vlast_bin + 1 + i, nbins, f1);
            if (k == r1) { pub[q][b] = &edgeR[0][0][nv - 1][i]; pubflip[q][b] = f1; has_role[q][b] = true; }
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < S; ++q)
#pragma unroll
    for (int b = 0; b < BPL; ++b)
    {
      flipmask[q][b] = flip[q][b] ? 0x80000000u : 0u;
      pubmask[q][b] = (flip[q][b] != pubflip[q][b]) ? 0x80000000u : 0u;
    }
  constexpr size_t kSlabU = (size_t)VW * HS;              // elements between consecutive u
  constexpr size_t kSlabBuf = (size_t)G * kSlabU;         // elements between the two buffers

  const SDFT_CONSTANT FD* d = as_uniform(a.delta + ch * a.n);
  const SDFT_CONSTANT TD* xin = SELF ? as_uniform(sa.x + ch * sa.x_stride) : nullptr;
  const SDFT_CONSTANT TD* hin = SELF ? as_uniform(sa.hist_in + ch * (size_t)span) : nullptr;
  const FD w = a.wscale;
  cx<FD>* row = a.out + ch * a.out_stride + t0 * (size_t)a.nbins;     // wave-uniform row base
  // lane-constant 32-bit element offsets into a row: the stores then use the scalar-base form
  // (global_store_dwordx4 v_off, v_data, s[row]) with no per-sample address arithmetic
  unsigned off_elems[S];
#pragma unroll
  for (int q = 0; q < S; ++q)
  {
    off_elems[q] = (unsigned)(((q * nwaves + wave) * kWave + lane) * BPL);
    __builtin_assume(off_elems[q] < (1u << 20));          // < 2048*BPL bins: the byte offset fits 32 bits
  }

  // SYN: padded row length of the terms image (bins of all waves + one 16-byte vector)
  const unsigned term_bins = (unsigned)(nv * kWave * BPL);
  const unsigned term_stride = term_bins + 16u / (unsigned)sizeof(FD);
  GainCursor<FD> gcur;                                     // SYN: the gain vector of the sample being finished
  const int opk = op_kind_of(fz.op);
  const bool op_has_rows = opk == OP_GAIN || opk == OP_CGAIN;
  gcur.g = nullptr; gcur.left = 0;
  if constexpr (SYN != 0) { if (op_has_rows) gcur.start(fz.op, t0, a.nbins); }
  size_t gtime = t0;                                       // time of the next sample finish() sees

  auto publish = [&](const cx<FD> (&x)[S][BPL], int buf, int u)
  {
    if constexpr (H >= 1)
    {
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int b = 0; b < BPL; ++b)
        {
          if (has_role[q][b])                             // a handful of lanes per wave (exec mask)
          {
            cx<FD> v = x[q][b];
            v.im = flip_sign(v.im, pubmask[q][b]);
            pub[q][b][(size_t)buf * kSlabBuf + (size_t)u * kSlabU] = v;
          }
        }
    }
  };

  // (optag: the operation as a compile-time constant -- the group loop below is generated once per operation and entered
  // through one switch per group, so the per-sample code carries no dispatch and none of the other operations)
  auto finish = [&](auto optag, const cx<FD> (&xin)[S][BPL], int buf, int u)
  {
    constexpr int kOp = decltype(optag)::value;
    const int opk = kOp >= 0 ? kOp : op_kind_of(fz.op);
    const bool op_has_rows = opk == OP_GAIN || opk == OP_CGAIN;
    if constexpr (SYN != 0) { if (op_has_rows) gcur.seek(gtime); ++gtime; }
#pragma unroll
    for (int q = 0; q < S; ++q)
    {
      const int v = q * nwaves + wave;
      cx<FD> x[BPL];
#pragma unroll
      for (int b = 0; b < BPL; ++b) { x[b] = xin[q][b]; x[b].im = flip_sign(x[b].im, flipmask[q][b]); }
      cx<FD> e[BPL + 4] = {};
#pragma unroll
      for (int b = 0; b < BPL; ++b) e[b + 2] = x[b];
      if constexpr (H >= 1)
      {
        const cx<FD> l0 = edgeL[buf][u][v][0], r0 = edgeR[buf][u][v][0];        // broadcast reads
        if constexpr (BPL == 1)
        {
          e[1] = from_below_fill(l0, x[0]);
          e[3] = from_above_fill(r0, x[0]);
          if constexpr (H >= 2)
          {
            const cx<FD> l1 = edgeL[buf][u][v][1], r1 = edgeR[buf][u][v][1];
            e[0] = from_below_fill(l1, e[1]);             // lane 1 receives lane 0's e[1] = l0
            e[4] = from_above_fill(r1, e[3]);
          }
        }
        else
        {
          e[1] = from_below_fill(l0, x[BPL - 1]);
          e[BPL + 2] = from_above_fill(r0, x[0]);
          if constexpr (H >= 2)
          {
            const cx<FD> l1 = edgeL[buf][u][v][1], r1 = edgeR[buf][u][v][1];
            e[0] = from_below_fill(l1, x[BPL - 2]);
            e[BPL + 3] = from_above_fill(r1, x[1]);
          }
        }
      }
      cx<FD> y[BPL];
#pragma unroll
      for (int b = 0; b < BPL; ++b)
      {
        if constexpr (FUSED) y[b] = window_tap_fused<FD, WIN>(e[b], e[b + 1], e[b + 2], e[b + 3], e[b + 4], w);
        else y[b] = window_tap<FD, WIN>(e[b], e[b + 1], e[b + 2], e[b + 3], e[b + 4], w);
      }
      if constexpr (SYN != 0)
      {
        // spectral operation, then the scalar sdft_isdft adds for this bin, parked at terms[u][bin]
        // (bins past N-1 in a partial last wave park +0: the walk adds whole padded rows)
#pragma unroll
        for (int b = 0; b < BPL; ++b)
        {
          const unsigned k = off_elems[q] + (unsigned)b;
          if (opk == OP_GAIN) y[b] = cscale(y[b], gcur.g[keep[q][b] ? k : 0]);
          else if (opk == OP_CGAIN) y[b] = cmul(y[b], reinterpret_cast<const cx<FD>*>(gcur.g)[keep[q][b] ? k : 0]);
#ifdef SDFT_USER_EXPR
          else if (opk == OP_USER) y[b] = user_op(y[b], k < a.nbins ? k : 0u, a.nbins, gtime - 1, ch, fz.op);
#endif
          else if (opk >= OP_GATE) y[b] = op_pointwise(y[b], fz.op, opk);
          SpectralOp<FD> shift_only = fz.op; shift_only.kind = op_kind_of(fz.op) == OP_SHIFT ? OP_SHIFT : OP_IDENTITY;
          shift_only.gain = nullptr;
          const FD term = synth_term<FD, LAT1, true>(y[b], k, shift_only, fz.syn, a.nbins);
          terms[((size_t)buf * G + (size_t)u) * term_stride + k] = keep[q][b] ? term : (FD)0;
        }
      }
      if (SYN == 0 || fz.store)
      {
      // destination = wave-uniform row base (scalar registers) + lane-constant 32-bit offset: the
      // row advance is scalar arithmetic, no per-lane 64-bit pointer bump
      cx<FD>* p = row + off_elems[q];
      if constexpr (BPL == 2)
      {
        if (a.vec_store)
        {
          if (keep[q][0])
          {
            using V = typename StoreVec<FD, 2>::type;
            V vv; vv.x = y[0].re; vv.y = y[0].im; vv.z = y[1].re; vv.w = y[1].im;
            store_vec(reinterpret_cast<V*>(p), vv);
          }
        }
        else
        {
          if (keep[q][0]) p[0] = y[0];
          if (keep[q][1]) p[1] = y[1];
        }
      }
      else
      {
        if (keep[q][0])
        {
          using V = typename StoreVec<FD, 1>::type;
          V vv; vv.x = y[0].re; vv.y = y[0].im;
          store_vec(reinterpret_cast<V*>(p), vv);
        }
      }
      }
    }
    row += a.nbins;
  };

  auto advance = [&](BinState<FD>& st, FD dl, bool wrap) -> cx<FD>
  {
    if constexpr (FUSED) return wrap ? step_wrap_fused(st, dl) : step_normal_fused(st, dl);
    else return wrap ? step_wrap(st, dl) : step_normal(st, dl);
  };

  // SYN: sum over bins -> one output sample per row of a group whose terms are in buffer `tb`
  auto sum_group = [&](int tb, int gm, size_t gt)
  {
    if constexpr (SYN != 0)
    {
      TD* yo = fz.y + ch * fz.y_stride + gt;
      const FD* tbase = terms + (size_t)tb * G * term_stride;
      if constexpr (SYN == 2 && sizeof(TD) == 4 && sizeof(FD) == 8)
      {
        // The reference's bits without the reference's order, where the output sample is a float: y = (float)(sum * w) is a
        // monotone function of the double sum, ANY order of the n additions is within g = n*2^-53/(1 - n*2^-53) times
        // sum|term| of the exact sum (the reference's order too), so the reference's sum lies within e = 2*g*sum|term| of
        // the tree sum -- and when both ends of that interval round to the same float, that float is the reference's
        // sample.  Otherwise (the interval straddles a rounding boundary of the float: a fraction of a percent of the
        // samples) the wave walks the terms in ascending order as the reference does (sdft.h:641-651).  NaNs fail the
        // comparison and take the walk.
        for (int u = wave; u < gm; u += nwaves)
        {
          const FD* tr = tbase + (size_t)u * term_stride;
          FD part = (FD)0, mag = (FD)0;
          for (unsigned k = lane; k < term_bins; k += kWave) { const FD v = tr[k]; part += v; mag += __builtin_fabs(v); }
          const FD sum = wave_sum_f(part), all = wave_sum_f(mag);
          const FD e = all * ((FD)2.5e-16 * (FD)term_bins);               // 2*g*sum|term| with 12 % to spare (g ~ n * 1.11e-16)
          const TD ylo = (TD)((sum - e) * fz.sweight), yhi = (TD)((sum + e) * fz.sweight);
          TD out = ylo;
          if (!(ylo == yhi))                                                // wave-uniform: every lane holds the same sums
          {
            typedef FD tvec __attribute__((ext_vector_type(2)));
            FD ordered = (FD)0;
            for (unsigned k0 = 0; k0 < term_bins; k0 += 16)                 // term_bins is a multiple of 64
            {
              tvec tv[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) tv[i] = *reinterpret_cast<const tvec*>(tr + k0 + i * 2);     // broadcast reads
#pragma unroll
              for (int i = 0; i < 8; ++i) { ordered += tv[i][0]; ordered += tv[i][1]; }
            }
            out = (TD)(ordered * fz.sweight);                                  // sdft.h:654-656
            if (lane == 0 && fz.walked) atomicAdd(fz.walked, 1u);
          }
          if (lane == 0) yo[u] = out;
        }
      }
      else if constexpr (SYN == 2)
      {
        // the reference's order (sdft.h:641-651): lane u of wave 0 adds sample u's terms bin by bin
        if (wave == 0 && lane < gm)
        {
          typedef FD tvec __attribute__((ext_vector_type(16 / sizeof(FD))));
          constexpr int NV = 16 / (int)sizeof(FD);
          const FD* tr = tbase + (size_t)lane * term_stride;
          FD sum = (FD)0;
          // (the chain of additions is the critical path of the kernel: 12 cycles per addition, 6.5 of them the dependent
          // v_add_f64 itself and the rest the issue of the eight-lane ds_read_b128; requesting the next vectors ahead of
          // the additions changes nothing -- scripts/add_latency_probe.hip)
          for (unsigned k0 = 0; k0 < term_bins; k0 += 8 * NV)       // term_bins is a multiple of 64
          {
            tvec tv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) tv[i] = *reinterpret_cast<const tvec*>(tr + k0 + i * NV);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
              for (int e = 0; e < NV; ++e) sum += tv[i][e];
          }
          yo[lane] = (TD)(sum * fz.sweight);                                   // sdft.h:654-656
        }
      }
      else
      {
        // wave-parallel: wave u sums sample u (lane-strided partial sums, shuffle reduction)
        for (int u = wave; u < gm; u += nwaves)
        {
          const FD* tr = tbase + (size_t)u * term_stride;
          FD part = (FD)0;
          for (unsigned k = lane; k < term_bins; k += kWave) part += tr[k];
          const FD sum = wave_sum_f(part);
          if (lane == 0) yo[u] = (TD)(sum * fz.sweight);
        }
      }
    }
  };
  bool have_prev = false;
  int prev_m = 0;
  size_t prev_t = 0;

  int buf = 0;
  size_t t = t0;
#ifdef SDFT_SELF_STAMPS
  if constexpr (SELF) { if (sa.stamps && chunk + 1 == a.chunks && threadIdx.x == 0) sa.stamps[4] = __builtin_readcyclecounter(); }
#endif
  while (t < t1)                       // all waves of the group take identical trip counts
  {
#ifdef SDFT_SELF_STAMPS
    if constexpr (SELF) { if (sa.stamps && chunk + 1 == a.chunks && threadIdx.x == 0 && t == t0 + (size_t)G) sa.stamps[5] = __builtin_readcyclecounter(); }
#endif
    const int m = (t1 - t < (size_t)G) ? (int)(t1 - t) : G;
    cx<FD> xs[G][S][BPL];
    // phase A
    if (m == G && c + G <= maxc)
    {
      FD dl[G];
      if constexpr (SELF) self_deltas<G>(dl, xin, hin, t, (size_t)span);
      else
      {
#pragma unroll
        for (int u = 0; u < G; ++u) dl[u] = d[t + u];
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
          for (int b = 0; b < BPL; ++b) xs[u][q][b] = advance(s[q][b], dl[u], false);
        publish(xs[u], buf, u);
      }
      c += G;
    }
    else
    {
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
        if (u < m)
        {
          const FD dl = SELF ? self_delta1<TD, FD>(xin, hin, t + u, (size_t)span) : d[t + u];
          const bool wrap = (c == maxc);
#pragma unroll
          for (int q = 0; q < S; ++q)
#pragma unroll
            for (int b = 0; b < BPL; ++b) xs[u][q][b] = advance(s[q][b], dl, wrap);
          c = wrap ? 0 : c + 1;
          publish(xs[u], buf, u);
        }
      }
    }
    __syncthreads();
    // phase B
    auto phase_b = [&](auto optag)
    {
      if (m == G)
      {
#pragma unroll
        for (int u = 0; u < G; ++u) finish(optag, xs[u], buf, u);
      }
      else
      {
#pragma unroll
        for (int u = 0; u < G; ++u)
          if (u < m) finish(optag, xs[u], buf, u);
      }
    };
    if constexpr (SYN != 0 && SDFT_FIXED_OP < 0)
    {
      switch (opk)
      {
        case OP_GAIN:  phase_b(OpTag<OP_GAIN>{}); break;
        case OP_CGAIN: phase_b(OpTag<OP_CGAIN>{}); break;
        case OP_GATE:  phase_b(OpTag<OP_GATE>{}); break;
        case OP_POWER: phase_b(OpTag<OP_POWER>{}); break;
        default:       phase_b(OpTag<OP_IDENTITY>{}); break;          // identity and shift (the shift acts in synth_term)
      }
    }
    else phase_b(OpTag<(SYN != 0 && SDFT_FIXED_OP >= 0) ? SDFT_FIXED_OP : OP_IDENTITY>{});
    if constexpr (SYN != 0)
    {
      // phase C runs one group behind: the terms image is double-buffered, group g's terms are
      // complete once every wave has passed the barrier of group g+1, so the walk over group g needs
      // no barrier of its own and overlaps the other waves' recurrence of group g+1
      if (have_prev) sum_group(buf ^ 1, prev_m, prev_t);
      have_prev = true; prev_m = m; prev_t = t;
    }
    t += m;
    buf ^= 1;
  }
  if constexpr (SYN != 0)
  {
    if (have_prev) { __syncthreads(); sum_group(buf ^ 1, prev_m, prev_t); }      // the last group
  }

  if (chunk + 1 == a.chunks)
  {
#pragma unroll
    for (int q = 0; q < S; ++q)
#pragma unroll
      for (int b = 0; b < BPL; ++b)
        if (keep[q][b])
        {
          const size_t k = (size_t)(q * nwaves + wave) * kWave * BPL + (size_t)lane * BPL + b;
          a.acc_state[ch * a.nbins + k] = s[q][b].acc;
          a.fid_state[ch * a.nbins + k] = s[q][b].fid;
        }
  }
#ifdef SDFT_SELF_STAMPS
  if constexpr (SELF) { if (sa.stamps && chunk + 1 == a.chunks && threadIdx.x == 0) sa.stamps[6] = __builtin_readcyclecounter(); }
#endif
  signal_done_workgroup(a.done);
}

// ------------------------------------------------------------------------------------------
// K2  inverse (sdft.h:635-657): one wave per row, 16-byte coalesced loads, per-lane strided
// partial sums, wave reduction by cross-lane shuffles, lane 0 scales and stores one TD sample.
// Summation order differs from the reference's serial bin loop: kept as the measurement
// alternative to inverse_exact_kernel (option exact_inverse = 0).
// ------------------------------------------------------------------------------------------
template <typename FD> SDFT_D FD wave_sum(FD v)
{
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------------------
// K3 (folded form)  analysis -> operation -> synthesis without ever forming the windowed spectrum.
//
// Everything after the demodulation X = acc * conj(fid) is linear: the window is a 3- or 5-tap
// convolution over bins (sdft.h:350-402) whose out-of-range taps are conjugate mirror images (:589-595),
// the supported operations are a real gain per bin or a shift of bins, and sdft_isdft adds, for every bin,
// re(Y) * (+-1) (latency 1, :643) or re(Y * twiddle) (:650).  So one output sample is
//     y = sweight * sum over bins r of ( alpha[r] * re X[r] + beta[r] * im X[r] )
// with coefficients that depend on the plan and the operation only (fold_coeff_kernel; beta == 0 for
// latency 1).  Per bin and sample that leaves the recurrence, two products for re X and one
// multiply-add: no neighbour exchange, no window arithmetic, no edge slots -- 9 instead of ~45 vector
// instructions per bin-sample at FD double.  The sum over bins: every lane adds its own J bins, a wave
// transposes its G x 64 partial sums through a private LDS tile (lane (u, s) adds eight of sample u's
// values, three DPP steps finish the row), the waves' sums meet in a ring of small tables, one barrier
// per four groups of G samples.  The order of the additions differs from the reference's: this is the tree-sum flavour
// of the fused call (not bit-identical; the ordered walk stays with forward_rows_kernel<SYN = 2>).
// ------------------------------------------------------------------------------------------
// sum over aligned groups of eight lanes, every lane of the group receiving it: two quad permutes and a
// mirror of the half row -- vector-ALU moves, no trip through the LDS crossbar like ds_bpermute
template <int CTRL> SDFT_D float dpp_move(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> SDFT_D double dpp_move(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <typename FD> SDFT_D FD sum_of_eight(FD v)
{
  v += dpp_move<0xB1>(v);                                  // quad_perm:[1,0,3,2]
  v += dpp_move<0x4E>(v);                                  // quad_perm:[2,3,0,1]
  v += dpp_move<0x141>(v);                                 // row_half_mirror: lane i <-> 7 - i of its eight
  return v;
}

// The sum over bins runs in double whatever FD is: the folded terms alpha * re X are an order of magnitude
// larger than the windowed terms the reference adds (the window's cancellation between neighbouring bins has
// moved into the coefficients), which in float arithmetic costs a digit (1.2e-4 against the reference's
// float result at N = 3000); in double the folded sum is the more accurate of the two.
template <typename TD, typename FD> struct ProcArgs
{
  TD* y;                      // [channels][n]
  size_t y_stride;
  const double* alpha;        // [rows][N]
  const double* beta;         // [rows][N]
  FD sweight;
  unsigned rows;              // coefficient vectors (time-varying gains; <= 1: one for the call)
  size_t hop;                 // samples per vector: vector r for the call's samples [r*hop, (r+1)*hop), the last for the rest
};

// alpha / beta for source bin r: every virtual position m whose mirror image is r (m = r, and m = -r or
// 2(N-1) - r at the ends), every tap i, output bin j = m - i inside the spectrum; A/B of an output bin are
// what sdft_isdft multiplies re / im of that bin with, after the operation.
template <typename FD>
__global__ __launch_bounds__(kBlock) void fold_coeff_kernel(double* alpha, double* beta, SpectralOp<FD> op, const cx<FD>* syn,
                                                            unsigned nbins, int lat1, FD h0, FD h1, FD h2)
{
  const long r = (long)blockIdx.x * kBlock + threadIdx.x, N = (long)nbins;
  if (r >= N) return;
  // one coefficient vector per gain vector (blockIdx.y): alpha / beta [rows][N]
  if (op.rows > 1 && op.gain) op.gain += (size_t)blockIdx.y * (size_t)nbins * (op.kind == OP_CGAIN ? 2u : 1u);
  alpha += (size_t)blockIdx.y * nbins; beta += (size_t)blockIdx.y * nbins;
  const double h[5] = {(double)h2, (double)h1, (double)h0, (double)h1, (double)h2};
  double al = 0.0, be = 0.0;
  auto add_position = [&](long m, bool flip)
  {
    for (int i = -2; i <= 2; ++i)
    {
      const long j = m - i;                                 // Y[j] takes tap i from position j + i = m
      if (j < 0 || j >= N) continue;
      const long ko = j + (op.kind == OP_SHIFT ? op.shift : 0);
      if (ko < 0 || ko >= N) continue;                      // shifted out of the spectrum
      double A, B;
      if (lat1) { A = (ko & 1) ? -1.0 : +1.0; B = 0.0; }                     // sdft.h:643
      else { A = (double)syn[ko].re; B = -(double)syn[ko].im; }              // re(Y * twiddle), :650
      if (op.kind == OP_GAIN) { A *= (double)op.gain[j]; B *= (double)op.gain[j]; }
      else if (op.kind == OP_CGAIN)
      {
        // term = re(Y * g * (A - iB)): the factors of re Y and im Y after the complex gain
        const cx<FD> g = reinterpret_cast<const cx<FD>*>(op.gain)[j];
        const double cr = (double)g.re * A + (double)g.im * B, ci = (double)g.im * A - (double)g.re * B;
        A = cr; B = -ci;
      }
      al += h[i + 2] * A;
      be += (flip ? -(h[i + 2] * B) : h[i + 2] * B);        // the mirror image is the conjugate
    }
  };
  add_position(r, false);
  if (r >= 1 && r <= 2) add_position(-r, true);
  const long mr = 2 * (N - 1) - r;
  if (mr >= N && mr <= N + 1) add_position(mr, true);
  alpha[r] = al;
  beta[r] = be;
}

#ifndef SDFT_PROC_RING
#define SDFT_PROC_RING 4
#endif
constexpr int kProcGroup = 8;            // samples per group
constexpr int kProcRow = 72;             // row stride of the transpose tile: 64 + 8, see the bank note in the kernel
constexpr int kProcRing = SDFT_PROC_RING;             // groups whose per-wave sums are in flight (a ring of tables)
constexpr int kProcSync = SDFT_PROC_RING / 2;         // groups per workgroup barrier (kProcRing >= 2 * kProcSync)

// (one bin per lane: two 16-wave workgroups share a CU -- 64 registers per lane, asked for by name)
template <typename TD, typename FD, int J, bool FUSED, bool HASB, bool SELF = false>
__global__ __launch_bounds__(kWave * kRowWavesMax, J == 1 ? 8 : 4) void process_rows_kernel(ForwardArgs<FD> a, ProcArgs<TD, FD> pz, SelfArgs<TD, FD> sa)
{
  constexpr int G = kProcGroup;
  constexpr int R = kProcRing, K = kProcSync;
  using AT = double;                                        // arithmetic type of everything after the recurrence
  // dynamic LDS: the waves' transpose tiles [waves][G * kProcRow] (the launch has as many waves as the row needs, so that
  // several workgroups share a CU), then the staged differences of a self-carried chunk
  extern __shared__ __align__(16) unsigned char proc_dyn_lds[];
  AT* const tiles = reinterpret_cast<AT*>(proc_dyn_lds);
  __shared__ AT part[R][kRowWavesMax][G];                  // [group][wave][sample]: eight lanes write eight neighbours

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  unsigned chunk; size_t ch;
  flow_position(a, chunk, ch);
  if (!flow_wait(a, chunk, ch)) return;                    // flow mode: the chunk's carries (a time-out ends the workgroup)

  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  const size_t t0 = chunk ? (size_t)chunk * a.chunk_len - a.chunk_shift : 0;
  const size_t tn = (size_t)(chunk + 1) * a.chunk_len - a.chunk_shift;
  const size_t t1 = tn < a.n ? tn : a.n;
  unsigned c = (unsigned)(((size_t)a.cursor0 + t0) % span);

  for (int i = threadIdx.x; i < R * G * kRowWavesMax; i += blockDim.x) (&part[0][0][0])[i] = (AT)0;   // waves that do not exist add 0
  // SELF: the chunk's differences, formed once by the workgroup (this kernel is bound by vector-instruction issue: formed in
  // the time loop from scalar loads they cost every wave 16 of its 74 instructions per 8 samples)
  FD* const dl_lds = reinterpret_cast<FD*>(tiles + (size_t)(blockDim.x >> 6) * G * kProcRow);
  const bool staged = SELF && sa.lds_deltas != 0 && (t1 - t0) <= (size_t)sa.lds_deltas;
  if constexpr (SELF)
  {
    if (staged)
    {
      const TD* xs = sa.x + ch * sa.x_stride;
      const TD* hs = sa.hist_in + ch * (size_t)span;
      for (size_t i = threadIdx.x; i < t1 - t0; i += blockDim.x)
      {
        const size_t tt = t0 + i;
        const TD dd = xs[tt] - (tt < span ? hs[tt] : xs[tt - span]);          // TD precision (sdft.h:564)
        dl_lds[i] = (FD)dd;
      }
    }
  }

  // SELF: carry-in by fold + FFT of everything before this chunk; the 2N cells borrow the transpose tiles, which
  // the time loop does not touch before the barrier below
  // (the host launches the self-carried form only where 2N cells fit the tiles: Plan::launch_process)
  cx<FD>* cells = reinterpret_cast<cx<FD>*>(tiles);
  cx<FD>* dft = nullptr;
  if constexpr (SELF) dft = self_carry<1, 8>(sa, a, cells, chunk, ch, t0);

  BinState<FD> s[J];
  AT al[J], be[J];
  bool live[J];
  const size_t cbase = (ch * a.chunks + chunk) * a.nbins;
#pragma unroll
  for (int j = 0; j < J; ++j)
  {
    const unsigned k = (unsigned)((j * nwaves + wave) * kWave + lane);         // strided: coalesced loads
    live[j] = k < a.nbins;
    const unsigned kk = live[j] ? k : 0u;
    s[j].tw = a.tw[kk];
    if constexpr (SELF)
    {
      s[j].acc = sa.acc_in[ch * a.nbins + kk];
      if (dft) s[j].acc = cadd(s[j].acc, dft[self_slot(sa, kk)]);
      s[j].fid = a.wtab[(size_t)(((unsigned long long)kk * c) % span)];
    }
    else
    {
    s[j].acc = a.carry[cbase + kk];
    s[j].fid = a.fseed ? fid_from_table(a.fseed, a.fseed_L, a.nbins, (long)kk, c, s[j].tw)
             : a.seed  ? a.seed[cbase + kk] : a.wtab[(size_t)(((unsigned long long)kk * c) % span)];
    }
    if (!live[j]) { s[j].tw = cmake<FD>((FD)0, (FD)0); s[j].acc = s[j].tw; s[j].fid = s[j].tw; }
    if constexpr (FUSED)
    {
      // chunk-parallel FD double path: carry the demodulated bin (see step_all)
      s[j].acc = cmake<FD>(__builtin_fma(s[j].acc.re, s[j].fid.re, s[j].acc.im * s[j].fid.im),
                           __builtin_fma(s[j].acc.im, s[j].fid.re, -(s[j].acc.re * s[j].fid.im)));      // X = acc * conj(fid)
      s[j].tw.im = -s[j].tw.im;
    }
  }
  // coefficients: one vector for the call, or (time-varying gains) vector r for the samples [r*hop, (r+1)*hop)
  size_t coeff_row = 0, coeff_next = ~(size_t)0;
  if (pz.rows > 1)
  {
    coeff_row = t0 / pz.hop;
    if (coeff_row >= pz.rows) coeff_row = pz.rows - 1;
    if (coeff_row + 1 < pz.rows) coeff_next = (coeff_row + 1) * pz.hop;
  }
  auto load_coeff = [&]()
  {
#pragma unroll
    for (int j = 0; j < J; ++j)
    {
      const unsigned k = (unsigned)((j * nwaves + wave) * kWave + lane);
      al[j] = live[j] ? pz.alpha[coeff_row * a.nbins + k] : (AT)0;
      be[j] = live[j] ? pz.beta[coeff_row * a.nbins + k] : (AT)0;
    }
  };
  load_coeff();
  __syncthreads();

  // one sample: the recurrence (sdft.h:566-587) for this lane's bins, then their share of the output sample
  auto step_all = [&](FD dl, bool wrap) -> AT
  {
    AT vv = (AT)0;
#pragma unroll
    for (int j = 0; j < J; ++j)
    {
      BinState<FD>& b = s[j];
      if constexpr (FUSED)
      {
        // The demodulated bin itself is carried through the chunk (b.acc holds X, b.tw holds conj(tw)):
        //   X' = (acc + fid*d) * conj(fid*tw) = (X + |fid|^2 d) * conj(tw) = (X + d) * conj(tw),
        // 1 addition + 1 complex multiplication = 5 instructions where acc, fid and the demodulation take 8 (sdft.h:583-585;
        // at the roll-over, :572-574, fid*tw is W[2N*k] = 1 and the same line holds).  What the modulated form is for -- no
        // error growth over an endless stream (sdft.h:6-16) -- is served by the chunk: X starts from (acc, fid) and runs
        // for at most a few thousand multiplications by a unit-modulus constant, 1e-16 relative each.
        (void)wrap;
        const FD xr0 = b.acc.re + dl;
        const FD nr = __builtin_fma(xr0, b.tw.re, -(b.acc.im * b.tw.im));
        const FD ni = __builtin_fma(xr0, b.tw.im, b.acc.im * b.tw.re);
        b.acc.re = nr; b.acc.im = ni;
        vv = __builtin_fma(al[j], nr, vv);
        if constexpr (HASB) vv = __builtin_fma(be[j], ni, vv);
      }
      else
      {
        if (wrap) advance_wrap(b, dl); else advance_normal(b, dl);                     // the stream state stays exact
        const AT ar = (AT)b.acc.re, ai = (AT)b.acc.im, fr = (AT)b.fid.re, fi = (AT)b.fid.im;
        const AT xr = ar * fr + ai * fi;
        vv += al[j] * xr;
        if constexpr (HASB)
        {
          const AT xi = ai * fr - ar * fi;
          vv += be[j] * xi;
        }
      }
    }
    return vv;
  };

  const SDFT_CONSTANT FD* d = as_uniform(a.delta + ch * a.n);
  const SDFT_CONSTANT TD* xin = SELF ? as_uniform(sa.x + ch * sa.x_stride) : nullptr;
  const SDFT_CONSTANT TD* hin = SELF ? as_uniform(sa.hist_in + ch * (size_t)span) : nullptr;
  TD* yo = pz.y + ch * pz.y_stride;
  AT* my = tiles + (size_t)wave * G * kProcRow;
  const int ru = lane >> 3, rs = lane & 7;                 // transposed role: sample of the group, segment of the row
  // the waves' sums of group g wait in part[g % R]; every K groups a barrier, after which K waves add one
  // finished group each (tables K .. 2K-1 groups back are rewritten only after the barrier that follows)
  auto finish_groups = [&](unsigned first, unsigned count)
  {
    for (unsigned g = first + (unsigned)wave; g < first + count; g += (unsigned)nwaves)
    {
      const size_t tg = t0 + (size_t)g * G;
      const int mg = (t1 - tg < (size_t)G) ? (int)(t1 - tg) : G;
      AT p = part[g % R][rs][ru] + part[g % R][rs + 8][ru];
      p = sum_of_eight(p);
      if (rs == 0 && ru < mg) yo[tg + ru] = (TD)(p * (AT)pz.sweight);           // sdft.h:654-656
    }
  };
  unsigned gi = 0;
  size_t t = t0;
  while (t < t1)                       // all waves of the group take identical trip counts
  {
    const int m = (t1 - t < (size_t)G) ? (int)(t1 - t) : G;
    AT v[G];
    if (m == G && c + G <= maxc && t + G <= coeff_next)
    {
      FD dl[G];
      if constexpr (SELF)
      {
        if (staged)
        {
#pragma unroll
          for (int u = 0; u < G; ++u) dl[u] = dl_lds[t - t0 + u];            // broadcast reads
        }
        else self_deltas<G>(dl, xin, hin, t, (size_t)span);
      }
      else
      {
#pragma unroll
        for (int u = 0; u < G; ++u) dl[u] = d[t + u];
      }
#pragma unroll
      for (int u = 0; u < G; ++u) v[u] = step_all(dl[u], false);
      c += G;
    }
    else
    {
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
        v[u] = (AT)0;
        if (u < m)
        {
          if (t + u == coeff_next)                          // the next gain vector takes over (workgroup-uniform)
          {
            ++coeff_row;
            coeff_next = (coeff_row + 1 < pz.rows) ? coeff_next + pz.hop : ~(size_t)0;
            load_coeff();
          }
          const FD dl = SELF ? self_delta1<TD, FD>(xin, hin, t + u, (size_t)span) : d[t + u];
          const bool wrap = (c == maxc);
          v[u] = step_all(dl, wrap);
          c = wrap ? 0 : c + 1;
        }
      }
    }
    // this wave's 64 partial sums per sample -> one sum per sample (LDS operations of one wave execute in order)
    // (rows are written contiguously: 16 consecutive lanes = 32 consecutive banks; lane (u, s) reads its eight
    // columns rotated by s, which with a row stride of 8 mod 32 doublewords puts the 32 lanes of a read on 32
    // different bank pairs)
#pragma unroll
    for (int u = 0; u < G; ++u) my[u * kProcRow + lane] = v[u];
    AT sum = my[ru * kProcRow + rs * 8 + (rs & 7)];
#pragma unroll
    for (int e = 1; e < 8; ++e) sum += my[ru * kProcRow + rs * 8 + ((e + rs) & 7)];
    sum = sum_of_eight(sum);
    if (rs == 0) part[gi % R][wave][ru] = sum;
    t += m;
    ++gi;
    if (gi % K == 0)
    {
      __syncthreads();
      finish_groups(gi - K, K);
    }
  }
  if (gi % K != 0)
  {
    __syncthreads();
    finish_groups(gi - gi % K, gi % K);
  }

  if (chunk + 1 == a.chunks)
  {
#pragma unroll
    for (int j = 0; j < J; ++j)
      if (live[j])
      {
        const size_t k = (size_t)((j * nwaves + wave) * kWave + lane);
        if constexpr (FUSED)
        {
          // back to the stream's state: fid at the cursor the call ends on (closed form, as the chunks were seeded), acc = X * fid
          const cx<FD> f = a.wtab[(size_t)(((unsigned long long)k * c) % span)];
          a.acc_state[ch * a.nbins + k] = cmul(s[j].acc, f);
          a.fid_state[ch * a.nbins + k] = f;
        }
        else
        {
          a.acc_state[ch * a.nbins + k] = s[j].acc;
          a.fid_state[ch * a.nbins + k] = s[j].fid;
        }
      }
  }
  signal_done_workgroup(a.done);
}

// ------------------------------------------------------------------------------------------
// K3h (folded form, calls of one time chunk)  a hop of the reference's streaming driver through the fused
// call in ONE launch: like forward_hop_kernel every 64 bins are one wave and one workgroup (the tiles land
// on different CUs), differences are formed from the input and the delay line by scalar loads, the state is
// double-buffered; like process_rows_kernel a bin contributes alpha * re X + beta * im X.  A wave leaves its
// sum per sample in partial[ch][tile][t]; the workgroup that takes the channel's last ticket (agent-scope
// acquire/release on a counter) adds the tiles in ascending order and writes the samples.  The recurrence is
// the unfused one: the state a call leaves behind is bit-identical to the reference's.
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD> struct ProcHopArgs
{
  const TD* x;                // [channels][n]
  size_t x_stride;
  TD* y;                      // [channels][n]
  size_t y_stride;
  const TD* hist_in;          // [channels][2N] delay line in time order
  TD* hist_out;
  const cx<FD>* tw;           // [N]
  const cx<FD>* acc_in;       // [channels][N]
  const cx<FD>* fid_in;
  cx<FD>* acc_out;
  cx<FD>* fid_out;
  const double* alpha;        // [N]
  const double* beta;
  double* partial;            // [channels][tiles][n]
  unsigned* tickets;          // [channels], zero between calls
  DoneSignal done;            // total = channels: every channel's last workgroup reports
  size_t n;
  unsigned nbins, tiles, cursor0;
  FD sweight;
};

template <typename TD, typename FD, bool HASB>
__global__ __launch_bounds__(kWave) void process_hop_kernel(ProcHopArgs<TD, FD> a)
{
  constexpr int G = kProcGroup;
  using AT = double;
  __shared__ AT tile_lds[G * kProcRow];
  __shared__ TD diff_lds[kHopMax + G];
  __shared__ unsigned last_flag;

  const int lane = threadIdx.x;
#ifdef SDFT_HOP_STAMPS
  unsigned long long stamp[6]; stamp[0] = __builtin_amdgcn_s_memrealtime();
#define SDFT_HOP_STAMP(i) stamp[i] = __builtin_amdgcn_s_memrealtime()
#else
#define SDFT_HOP_STAMP(i)
#endif
  const unsigned tile = blockIdx.x % a.tiles;
  const size_t ch = blockIdx.x / a.tiles;
  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  unsigned c = a.cursor0;

  const unsigned k = tile * kWave + (unsigned)lane;
  const bool live = k < a.nbins;
  const unsigned kk = live ? k : 0u;
  const size_t sbase = ch * a.nbins;
  BinState<FD> s;
  s.tw = a.tw[kk]; s.acc = a.acc_in[sbase + kk]; s.fid = a.fid_in[sbase + kk];
  AT al = a.alpha[kk], be = a.beta[kk];
  if (!live) { s.tw = cmake<FD>((FD)0, (FD)0); s.acc = s.tw; s.fid = s.tw; al = (AT)0; be = (AT)0; }

  // delay line for the next call: element i of the last 2N samples of (hist ++ x)
  {
    const TD* xv = a.x + ch * a.x_stride;
    const TD* hv = a.hist_in + ch * span;
    TD* ho = a.hist_out + ch * span;
    for (size_t i = (size_t)tile * kWave + lane; i < span; i += (size_t)a.tiles * kWave)
    {
      const size_t j = a.n + i;
      ho[i] = (j >= span) ? xv[j - span] : hv[j];
    }
  }

  // differences of the whole call (sdft.h:564; the old sample comes from the delay line while t < 2N, from the
  // call's own input afterwards; the subtraction in TD precision) staged in LDS up front: one round of vector
  // loads instead of a scalar-load latency (~1 us for a lone wave) in front of every group of samples
  {
    const TD* xv = a.x + ch * a.x_stride;
    const TD* hv = a.hist_in + ch * span;
    for (size_t tt = lane; tt < a.n; tt += kWave)
    {
      const TD cur = xv[tt];
      const TD old = (tt < span) ? hv[tt] : xv[tt - span];
      diff_lds[tt] = cur - old;
    }
    for (size_t tt = a.n + lane; tt < ((a.n + G - 1) / G) * G; tt += kWave) diff_lds[tt] = (TD)0;
  }
  double* mine = a.partial + (ch * a.tiles + tile) * a.n;
  const int ru = lane >> 3, rs = lane & 7;
  SDFT_HOP_STAMP(1);

  auto step = [&](FD dl, bool wrap) -> AT
  {
    if (wrap) advance_wrap(s, dl); else advance_normal(s, dl);                         // sdft.h:566-587, unfused
    const AT ar = (AT)s.acc.re, ai = (AT)s.acc.im, fr = (AT)s.fid.re, fi = (AT)s.fid.im;
    AT vv = al * (ar * fr + ai * fi);
    if constexpr (HASB) vv += be * (ai * fr - ar * fi);
    return vv;
  };

  for (size_t t = 0; t < a.n; t += G)
  {
    const int m = (a.n - t < (size_t)G) ? (int)(a.n - t) : G;
    TD dd[G];
#pragma unroll
    for (int u = 0; u < G; ++u) dd[u] = diff_lds[t + u];                                // broadcast reads
    AT v[G];
    if (m == G && c + G <= maxc)
    {
#pragma unroll
      for (int u = 0; u < G; ++u) v[u] = step((FD)dd[u], false);
      c += G;
    }
    else
    {
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
        v[u] = (AT)0;
        if (u < m)
        {
          const bool wrap = (c == maxc);
          v[u] = step((FD)dd[u], wrap);
          c = wrap ? 0 : c + 1;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) tile_lds[u * kProcRow + lane] = v[u];
    AT sum = tile_lds[ru * kProcRow + rs * 8 + (rs & 7)];
#pragma unroll
    for (int e = 1; e < 8; ++e) sum += tile_lds[ru * kProcRow + rs * 8 + ((e + rs) & 7)];
    sum = sum_of_eight(sum);
    if (rs == 0 && ru < m) mine[t + ru] = sum;
  }

  SDFT_HOP_STAMP(2);
  if (live)
  {
    a.acc_out[sbase + k] = s.acc;
    a.fid_out[sbase + k] = s.fid;
  }

  // the channel's last workgroup adds the tiles (release: this wave's stores; acquire: everybody else's)
  if (lane == 0)
  {
    const unsigned ticket = __hip_atomic_fetch_add(a.tickets + ch, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (ticket + 1u == a.tiles) ? 1u : 0u;
  }
  __syncthreads();
  SDFT_HOP_STAMP(3);
  if (!last_flag) return;
  if (lane == 0) __hip_atomic_store(a.tickets + ch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  const double* all = a.partial + ch * a.tiles * a.n;
  TD* yo = a.y + ch * a.y_stride;
  // (all loads of a batch -- 16 tiles x 2 samples per lane -- are requested before the first is added: the
  // partial sums come from other XCDs' L2s through memory, a microsecond per dependent round trip)
  for (size_t tb = 0; tb < a.n; tb += 2 * kWave)
  {
    const size_t t0 = tb + lane, t1 = tb + kWave + lane;
    AT p0 = (AT)0, p1 = (AT)0;
    for (unsigned q0 = 0; q0 < a.tiles; q0 += 16)
    {
      AT pv0[16], pv1[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
      {
        const bool tq = q0 + (unsigned)i < a.tiles;
        pv0[i] = (tq && t0 < a.n) ? all[(size_t)(q0 + i) * a.n + t0] : (AT)0;
        pv1[i] = (tq && t1 < a.n) ? all[(size_t)(q0 + i) * a.n + t1] : (AT)0;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { p0 += pv0[i]; p1 += pv1[i]; }
    }
    if (t0 < a.n) yo[t0] = (TD)(p0 * (AT)a.sweight);                                    // sdft.h:654-656
    if (t1 < a.n) yo[t1] = (TD)(p1 * (AT)a.sweight);
  }
  // completion word for a synchronous host: the launch's last channel publishes it after its samples (a kernel's
  // end reaches the host ~6 us later than a store to pinned memory does)
  if (lane == 0) signal_done(a.done);
#ifdef SDFT_HOP_STAMPS
  SDFT_HOP_STAMP(4);
  if (lane == 0) for (int i = 0; i < 5; ++i) reinterpret_cast<unsigned long long*>(a.partial + (size_t)gridDim.x * a.n)[i] = stamp[i];
#endif
}
#undef SDFT_HOP_STAMP

// ------------------------------------------------------------------------------------------
// K3h, two waves per tile: the lone-wave lesson of forward_hop2_kernel applied to the fused hop.  Wave 0 runs the
// recurrence and parks (acc, fid) of an 8-sample group in a double-buffered LDS image; wave 1 takes the group one
// barrier later, forms alpha * re X + beta * im X, transposes and adds its 64 lanes and writes the per-sample sums
// of the tile.  Ticket, combine and completion word as in process_hop_kernel (the combine by both waves).
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD, bool HASB>
__global__ __launch_bounds__(2 * kWave) void process_hop2_kernel(ProcHopArgs<TD, FD> a)
{
  constexpr int G = kProcGroup;
  using AT = double;
  __shared__ cx<FD> image[2][G][2][kWave];                 // [buffer][sample][acc | fid][lane]
  __shared__ AT tile_lds[G * kProcRow];
  __shared__ TD diff_lds[kHopMax + G];
  __shared__ unsigned last_flag;

  const int lane = threadIdx.x & (kWave - 1);
  const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0 recurrence, 1 coefficients + sums
  const unsigned tile = blockIdx.x % a.tiles;
  const size_t ch = blockIdx.x / a.tiles;
  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  const unsigned k = tile * kWave + (unsigned)lane;
  const bool live = k < a.nbins;
  const unsigned kk = live ? k : 0u;
  const size_t sbase = ch * a.nbins;

  {
    const TD* xv = a.x + ch * a.x_stride;
    const TD* hv = a.hist_in + ch * span;
    TD* ho = a.hist_out + ch * span;
    // delay line for the next call: element i of the last 2N samples of (hist ++ x)
    for (size_t i = (size_t)tile * (2 * kWave) + threadIdx.x; i < span; i += (size_t)a.tiles * (2 * kWave))
    {
      const size_t j = a.n + i;
      ho[i] = (j >= span) ? xv[j - span] : hv[j];
    }
    // differences of the whole call (sdft.h:564), the subtraction in TD precision
    for (size_t tt = threadIdx.x; tt < a.n; tt += 2 * kWave)
    {
      const TD cur = xv[tt];
      const TD old = (tt < span) ? hv[tt] : xv[tt - span];
      diff_lds[tt] = cur - old;
    }
  }

  const size_t groups = (a.n + G - 1) / G;
  if (role == 0)
  {
    // ---------------- recurrence (unfused: the state stays the reference's) ----------------
    BinState<FD> s;
    s.tw = a.tw[kk]; s.acc = a.acc_in[sbase + kk]; s.fid = a.fid_in[sbase + kk];
    if (!live) { s.tw = cmake<FD>((FD)0, (FD)0); s.acc = s.tw; s.fid = s.tw; }
    __syncthreads();                                         // the differences are staged
    unsigned c = a.cursor0;
    int buf = 0;
    for (size_t g = 0; g < groups; ++g)
    {
      const size_t t = g * G;
      const int m = (a.n - t < (size_t)G) ? (int)(a.n - t) : G;
      TD dd[G];
#pragma unroll
      for (int u = 0; u < G; ++u) dd[u] = diff_lds[t + u];   // broadcast reads
      if (m == G && c + G <= maxc)
      {
#pragma unroll
        for (int u = 0; u < G; ++u)
        {
          advance_normal(s, (FD)dd[u]);
          image[buf][u][0][lane] = s.acc;
          image[buf][u][1][lane] = s.fid;
        }
        c += G;
      }
      else
      {
#pragma unroll
        for (int u = 0; u < G; ++u)
        {
          if (u < m)
          {
            const bool wrap = (c == maxc);                   // wave-uniform
            if (wrap) advance_wrap(s, (FD)dd[u]); else advance_normal(s, (FD)dd[u]);
            image[buf][u][0][lane] = s.acc;
            image[buf][u][1][lane] = s.fid;
            c = wrap ? 0 : c + 1;
          }
        }
      }
      __syncthreads();                                       // group g is in the image
      buf ^= 1;
    }
    if (live)
    {
      a.acc_out[sbase + k] = s.acc;
      a.fid_out[sbase + k] = s.fid;
    }
  }
  else
  {
    // ---------------- coefficients and the sum over the tile's bins ----------------
    AT al = a.alpha[kk], be = a.beta[kk];
    if (!live) { al = (AT)0; be = (AT)0; }
    double* mine = a.partial + (ch * a.tiles + tile) * a.n;
    const int ru = lane >> 3, rs = lane & 7;
    __syncthreads();                                         // (pairs with the barrier after the staging)
    int buf = 0;
    for (size_t g = 0; g < groups; ++g)
    {
      const size_t t = g * G;
      const int m = (a.n - t < (size_t)G) ? (int)(a.n - t) : G;
      __syncthreads();                                       // group g is in the image
      cx<FD> ac[G], fi[G];
#pragma unroll
      for (int u = 0; u < G; ++u) { ac[u] = image[buf][u][0][lane]; fi[u] = image[buf][u][1][lane]; }   // all reads first
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
        const AT ar = (AT)ac[u].re, ai = (AT)ac[u].im, fr = (AT)fi[u].re, fm = (AT)fi[u].im;
        AT vv = al * (ar * fr + ai * fm);
        if constexpr (HASB) vv += be * (ai * fr - ar * fm);
        tile_lds[u * kProcRow + lane] = (m == G || u < m) ? vv : (AT)0;        // samples past the call's end hold stale bins
      }
      AT sum = tile_lds[ru * kProcRow + rs * 8 + (rs & 7)];
#pragma unroll
      for (int e = 1; e < 8; ++e) sum += tile_lds[ru * kProcRow + rs * 8 + ((e + rs) & 7)];
      sum = sum_of_eight(sum);
      if (rs == 0 && ru < m) mine[t + ru] = sum;
      buf ^= 1;
    }
  }

  // the channel's last workgroup adds the tiles (release: both waves' stores; acquire: everybody else's)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const unsigned ticket = __hip_atomic_fetch_add(a.tickets + ch, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (ticket + 1u == a.tiles) ? 1u : 0u;
  }
  __syncthreads();
  if (!last_flag) return;
  if (threadIdx.x == 0) __hip_atomic_store(a.tickets + ch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  // (only thread 0 has executed the acquire; the other lanes' loads go to the same coherent level explicitly)
  const double* all = a.partial + ch * a.tiles * a.n;
  TD* yo = a.y + ch * a.y_stride;
  for (size_t tb = 0; tb < a.n; tb += 2 * kWave)
  {
    const size_t t0 = tb + threadIdx.x;
    AT p0 = (AT)0;
    for (unsigned q0 = 0; q0 < a.tiles; q0 += 16)
    {
      AT pv0[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        pv0[i] = (q0 + (unsigned)i < a.tiles && t0 < a.n) ? __hip_atomic_load(all + (size_t)(q0 + i) * a.n + t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (AT)0;
#pragma unroll
      for (int i = 0; i < 16; ++i) p0 += pv0[i];
    }
    if (t0 < a.n) yo[t0] = (TD)(p0 * (AT)a.sweight);                                    // sdft.h:654-656
  }
  signal_done_workgroup(a.done);
}

// rows[ch][t][k] = op(rows[ch][t][k]) for the operations that change a bin in place (the processed copy of the spectrum
// on the two-pass path of sdft_hip_process_n)
template <typename FD>
__global__ __launch_bounds__(kBlock) void scale_rows_kernel(cx<FD>* mat, size_t stride, size_t rows, unsigned nbins, unsigned channels, SpectralOp<FD> op)
{
  const size_t per = rows * nbins, total = per * channels;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock)
  {
    const size_t ch = i / per, r = i - ch * per;
    const size_t t = r / nbins, k = r - t * nbins;
    cx<FD>* p = mat + ch * stride + r;
    const FD* g = gain_row(op, t, nbins);
    if (op.kind == OP_GAIN) *p = cscale(*p, g[k]);
    else if (op.kind == OP_CGAIN) *p = cmul(*p, reinterpret_cast<const cx<FD>*>(g)[k]);
    else if (op.kind >= OP_GATE) *p = op_pointwise(*p, op, op.kind);
  }
}

template <typename TD, typename FD> struct InverseArgs
{
  const cx<FD>* in;           // rows: in + ch*in_stride + t*N
  size_t in_stride;
  const cx<FD>* const* in_rows;   // optional row-pointer table (sdft_isdft_nd)
  const cx<FD>* syn;          // [N]
  TD* y;                      // y + ch*y_stride + t
  size_t y_stride;
  size_t n;
  unsigned nbins, channels;
  FD sweight;
  SpectralOp<FD> op;          // applied to every bin on the way in (identity for sdft_isdft_n)
  DoneSignal done;            // inverse_row_kernel only: total = rows
};

// VERIFY (float samples from double bins): the reference's bits from the tree sum -- the rounding-interval test of
// forward_rows_kernel<SYN = 2>; a row whose interval straddles a rounding boundary of the float is read again (it is in
// cache) and added in ascending bin order, lane by lane.
template <typename TD, typename FD, bool LAT1, bool OPS = false, bool VERIFY = false>
__global__ __launch_bounds__(kBlock) void inverse_kernel(InverseArgs<TD, FD> a)
{
  static_assert(!VERIFY || (sizeof(TD) == 4 && sizeof(FD) == 8), "the interval test needs a rounding to hide behind");
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  const size_t rows = (size_t)a.channels * a.n;
  // rows are taken from the END of the matrix first: a round trip calls this right after the analysis
  // has written the matrix, whose tail is what still sits in the 256 MiB Infinity Cache (measured:
  // -12 % at 197 MB, -2 % at 16 GB, nothing at 786 MB)
  for (size_t ri = (size_t)blockIdx.x * kWavesPerBlock + wib; ri < rows; ri += nwaves)
  {
    const size_t r = rows - 1 - ri;
    const size_t ch = r / a.n, t = r - ch * a.n;
    const cx<FD>* row = a.in_rows ? a.in_rows[r] : a.in + ch * a.in_stride + t * (size_t)a.nbins;
    FD part = (FD)0, mag = (FD)0;
    const FD* grow = OPS ? gain_row(a.op, t, a.nbins) : nullptr;
#pragma unroll 4
    for (unsigned k = lane; k < a.nbins; k += kWave)
    {
      const FD tv = synth_term<FD, LAT1, OPS>(row[k], k, a.op, a.syn, a.nbins, grow);
      part += tv;
      if constexpr (VERIFY) mag += __builtin_fabs(tv);
    }
    const FD sum = wave_sum(part);
    TD out = (TD)(sum * a.sweight);
    if constexpr (VERIFY)
    {
      const FD all = wave_sum(mag);
      const FD e = all * ((FD)2.5e-16 * (FD)(a.nbins + kWave));
      const TD ylo = (TD)((sum - e) * a.sweight), yhi = (TD)((sum + e) * a.sweight);
      out = ylo;
      if (!(ylo == yhi))                                   // wave-uniform (every lane holds the wave's sums)
      {
        FD ordered = (FD)0;
        for (unsigned k0 = 0; k0 < a.nbins; k0 += kWave)
        {
          const unsigned k = k0 + (unsigned)lane;
          const FD tv = k < a.nbins ? synth_term<FD, LAT1, OPS>(row[k], k, a.op, a.syn, a.nbins, grow) : (FD)0;
          const int lo = __double2loint(tv), hi = __double2hiint(tv);
          const unsigned cnt = a.nbins - k0 < (unsigned)kWave ? a.nbins - k0 : (unsigned)kWave;
          for (unsigned j = 0; j < cnt; ++j)               // sdft.h:641-651: one accumulator, ascending bins
            ordered += __hiloint2double(__builtin_amdgcn_readlane(hi, (int)j), __builtin_amdgcn_readlane(lo, (int)j));
        }
        out = (TD)(ordered * a.sweight);
      }
    }
    if (lane == 0) a.y[ch * a.y_stride + t] = out;
  }
}

// ------------------------------------------------------------------------------------------
// K2 (exact order)  inverse with the reference's summation order (sdft.h:641-651: one accumulator
// per row, bins added in ascending order), at streaming bandwidth: a wave owns RW consecutive rows
// and, in the summation phase, lane r adds row r's terms strictly in bin order.  Tiles of RW rows x
// 256 bytes are fetched with 16-byte loads (one instruction = four 256-byte row segments), the
// scalar each bin contributes -- re(X)*(+-1) for latency 1, re(X * twiddle) otherwise -- goes to a
// padded LDS tile, and the next tile's loads are in flight while the current one is summed.
// Result: bit-identical to the reference for every type.  RW = 32 (one tile ahead) for long FD
// double calls, 16 for FD float and medium calls, 4 with an 8-deep ring for short calls (a hop of
// 100 rows has too few rows to hide latency with row-parallelism alone).
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD, bool LAT1, int RW, int DEPTH, bool OPS = false>
__global__ __launch_bounds__(kBlock) void inverse_exact_kernel(InverseArgs<TD, FD> a)
{
  constexpr int BPL = 16 / (int)sizeof(cx<FD>);          // bins per 16-byte load (1 for f64, 2 for f32)
  constexpr int C = 16 * BPL;                            // bins per tile row = 256 bytes
  constexpr int RPI = 4;                                 // rows per load instruction (16 lanes each)
  constexpr int NI = RW / RPI;                           // load instructions per tile
  using V = typename StoreVec<FD, (sizeof(cx<FD>) == 16 ? 1 : 2)>::type;   // 16-byte vector
  __shared__ FD tile[kWavesPerBlock][RW][C + 1];

  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t ngroups_per_ch = (a.n + RW - 1) / RW;
  const size_t ngroups = ngroups_per_ch * a.channels;
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  const int sub = lane >> 4, seg = lane & 15;            // load phase: row within the instruction, 16-byte slot
  const bool vec_ok = (BPL == 1) || ((a.nbins % 2 == 0) && !a.in_rows && (a.in_stride % 2 == 0));

  // row groups are taken from the END of the matrix first: a round trip calls this right after the
  // analysis has written the matrix, whose tail is what still sits in the 256 MiB Infinity Cache (and
  // is dirty there: reading the head first makes the cache write the tail back while HBM is being read)
  for (size_t gi = (size_t)blockIdx.x * kWavesPerBlock + wib; gi < ngroups; gi += nwaves)
  {
    const size_t g = ngroups - 1 - gi;
    const size_t ch = g / ngroups_per_ch;
    const size_t r0 = (g - ch * ngroups_per_ch) * RW;
    const cx<FD>* base = a.in + ch * a.in_stride;
    const FD* grow[NI];                                    // OPS: the gain vector of each row this lane stages
#pragma unroll
    for (int i = 0; i < NI; ++i) grow[i] = OPS ? gain_row(a.op, r0 + (size_t)(RPI * i + sub), a.nbins) : nullptr;

    auto fetch = [&](unsigned k0, cx<FD> (&v)[NI][BPL])
    {
#pragma unroll
      for (int i = 0; i < NI; ++i)
      {
        const size_t r = r0 + (size_t)(RPI * i + sub);
        const unsigned k = k0 + (unsigned)seg * BPL;
#pragma unroll
        for (int b = 0; b < BPL; ++b) v[i][b] = cmake<FD>((FD)0, (FD)0);
        if (r < a.n && k < a.nbins)
        {
          const cx<FD>* rowp = a.in_rows ? a.in_rows[ch * a.n + r] : base + r * (size_t)a.nbins;
          if (BPL == 2 && vec_ok && k + 1 < a.nbins)
          {
            const V q = *reinterpret_cast<const V*>(rowp + k);
            v[i][0] = cmake<FD>((FD)q[0], (FD)q[1]);
            if constexpr (BPL == 2) v[i][1] = cmake<FD>((FD)q[2], (FD)q[3]);
          }
          else
          {
#pragma unroll
            for (int b = 0; b < BPL; ++b)
              if (k + b < a.nbins) v[i][b] = rowp[k + b];
          }
        }
      }
    };
    auto stage = [&](unsigned k0, const cx<FD> (&v)[NI][BPL])
    {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int b = 0; b < BPL; ++b)
        {
          const unsigned k = k0 + (unsigned)seg * BPL + b;
          tile[wib][RPI * i + sub][seg * BPL + b] = synth_term<FD, LAT1, OPS>(v[i][b], k, a.op, a.syn, a.nbins, grow[i]);
        }
    };

    FD sum = (FD)0;
    // ring of DEPTH tiles in registers: tile t is consumed while tiles t+1 .. t+DEPTH are in flight
    cx<FD> ring[DEPTH][NI][BPL];
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) fetch((unsigned)dd * C, ring[dd]);
    for (unsigned kb = 0; kb < a.nbins; kb += DEPTH * C)
    {
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd)
      {
        const unsigned k0 = kb + (unsigned)dd * C;
        if (k0 < a.nbins)                                // wave-uniform
        {
          stage(k0, ring[dd]);
          fetch(k0 + DEPTH * C, ring[dd]);               // past the row end: predicated off, zeros
          __builtin_amdgcn_wave_barrier();
          const unsigned cnt = (a.nbins - k0 < (unsigned)C) ? a.nbins - k0 : (unsigned)C;
          if (lane < RW)
          {
            if (cnt == (unsigned)C)
            {
#pragma unroll
              for (int c = 0; c < C; ++c) sum += tile[wib][lane][c];
            }
            else
            {
              for (unsigned c = 0; c < cnt; ++c) sum += tile[wib][lane][c];
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    const size_t r = r0 + lane;
    if (lane < RW && r < a.n) a.y[ch * a.y_stride + r] = (TD)(sum * a.sweight);     // sdft.h:654-656
  }
}

// ------------------------------------------------------------------------------------------
// K2 (row form, short calls)  exact-order synthesis for calls with few rows (a 100-row hop): one
// wave per row.  The lanes fetch the whole row with every load in flight at once, turn bins into
// the scalars the reference adds (sdft.h:643 / :650) and park them in LDS in bin order; then all
// lanes walk the LDS block with broadcast reads and add the terms strictly in ascending bin
// order (every lane holds the same sum: no exec masking, same cost as one lane).  What remains is
// the chain of N dependent additions the reference's summation order dictates.
// ------------------------------------------------------------------------------------------
template <typename TD, typename FD, bool LAT1, bool OPS = false>
__global__ __launch_bounds__(kWave) void inverse_row_kernel(InverseArgs<TD, FD> a)
{
  constexpr int BPL = 16 / (int)sizeof(cx<FD>);          // bins per 16-byte load
  constexpr int NL = 16;                                 // loads in flight per lane
  constexpr int TB = kWave * NL * BPL;                   // bins per LDS block (1024 f64 / 2048 f32: 8 KiB)
  using V = typename StoreVec<FD, (sizeof(cx<FD>) == 16 ? 1 : 2)>::type;
  __shared__ __align__(16) FD terms[TB];

  const int lane = threadIdx.x;
  const size_t r = (size_t)gridDim.x - 1 - blockIdx.x;       // last rows first (what the analysis wrote last is still in cache)
  const size_t ch = r / a.n, t = r - ch * a.n;
  const cx<FD>* row = a.in_rows ? a.in_rows[r] : a.in + ch * a.in_stride + t * (size_t)a.nbins;
  const FD* grow = OPS ? gain_row(a.op, t, a.nbins) : nullptr;
  const bool vec_ok = (BPL == 1) || ((a.nbins % 2 == 0) && (((uintptr_t)row & 15) == 0));

  FD sum = (FD)0;
  // float samples from double bins, rows of one LDS block: the rounding-interval test of forward_rows_kernel<SYN = 2> --
  // the tree sum and 2*n*2^-53*sum|term| bound the reference's ordered sum; when both ends of the interval round to the
  // same float the N dependent additions are not needed (most rows), else they are made as before.  Same bits either way.
  constexpr bool kInterval = sizeof(TD) == 4 && sizeof(FD) == 8;
  bool decided = false;
  TD decided_y = (TD)0;
  for (unsigned k0 = 0; k0 < a.nbins; k0 += TB)
  {
    cx<FD> v[NL][BPL];
#pragma unroll
    for (int i = 0; i < NL; ++i)
    {
      const unsigned k = k0 + (unsigned)(i * kWave + lane) * BPL;
#pragma unroll
      for (int b = 0; b < BPL; ++b) v[i][b] = cmake<FD>((FD)0, (FD)0);
      if (k < a.nbins)
      {
        if (BPL == 2 && vec_ok && k + 1 < a.nbins)
        {
          const V q = *reinterpret_cast<const V*>(row + k);
          v[i][0] = cmake<FD>((FD)q[0], (FD)q[1]);
          if constexpr (BPL == 2) v[i][1] = cmake<FD>((FD)q[2], (FD)q[3]);
        }
        else
        {
#pragma unroll
          for (int b = 0; b < BPL; ++b)
            if (k + b < a.nbins) v[i][b] = row[k + b];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int b = 0; b < BPL; ++b)
      {
        const unsigned kl = (unsigned)(i * kWave + lane) * BPL + b;
        const unsigned k = k0 + kl;
#ifdef SDFT_USER_EXPR
        // (run-time compilation for the host's own statements: they act on the bin before the synthesis term is formed)
        if constexpr (OPS) { if (k < a.nbins) v[i][b] = user_op(v[i][b], k, a.nbins, a.op.t0 + t, ch, a.op); }
#endif
        terms[kl] = synth_term<FD, LAT1, OPS>(v[i][b], k, a.op, a.syn, a.nbins, grow);
      }
    if constexpr (kInterval)
    {
      if (a.nbins <= (unsigned)TB)                           // (wave-uniform; bins past N-1 park +0)
      {
        FD part = (FD)0, mag = (FD)0;
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int b = 0; b < BPL; ++b) { const FD tv = terms[(unsigned)(i * kWave + lane) * BPL + b]; part += tv; mag += __builtin_fabs(tv); }
        const FD tree = wave_sum_f(part), all = wave_sum_f(mag);
        const FD e = all * ((FD)2.5e-16 * (FD)TB);
        const TD ylo = (TD)((tree - e) * a.sweight), yhi = (TD)((tree + e) * a.sweight);
        if (ylo == yhi) { decided = true; decided_y = ylo; break; }
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const unsigned cnt = (a.nbins - k0 < (unsigned)TB) ? a.nbins - k0 : (unsigned)TB;
    if (cnt == (unsigned)TB)
    {
#pragma unroll 32
      for (int cix = 0; cix < TB; ++cix) sum += terms[cix];
    }
    else
    {
      unsigned cix = 0;
      for (; cix + 16 <= cnt; cix += 16)
      {
        FD tt[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) tt[q] = terms[cix + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += tt[q];
      }
      for (; cix < cnt; ++cix) sum += terms[cix];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0)
  {
    a.y[ch * a.y_stride + t] = decided ? decided_y : (TD)(sum * a.sweight);           // sdft.h:654-656
    signal_done(a.done);
  }
}

}  // namespace sdfthip
