// sdft_base.hpp -- what every stage shares: complex helpers (sdft.h:265-331), cross-lane neighbour fetch by DPP
// whole-wave shifts, index reflection for the halo (sdft.h:589-595), K0 delta_kernel (differences in TD precision +
// delay line, sdft.h:186-191 / :564), the recurrence step (sdft.h:572-585) and the completion word of short calls.
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

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

// the rounding-interval proof of the tree sums accepts a result only when both ends of the interval round to the SAME BITS:
// -0.0f == +0.0f compares equal, and the reference's ordered sum may land on either zero (results below the smallest subnormal)
SDFT_D bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
SDFT_D bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

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


// the same for a workgroup of several waves: every wave waits for its own stores, the workgroup meets, one lane reports
SDFT_D void signal_done_workgroup(const DoneSignal& d)
{
  if (!d.flag) return;                                      // workgroup-uniform
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) signal_done(d);
}

}  // namespace sdfthip
