// sdft_carry_fast.hpp -- K1a/K1b: carries of the chunk-parallel FD double path (partial sums per chunk: direct, FFT, mixed-radix FFT; scan over chunks)
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_base.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

// ------------------------------------------------------------------------------------------
// K1a (fast carry, FD double): per (chunk, bin) partial sums of delta*fid over one chunk,
// written to carry[ch][chunk+1][k]; K1b turns them into carry-ins by an exclusive scan over
// chunks.  fid is seeded from the table W[j] = exp(-i*pi*j/N), j = k*cursor mod 2N, and then
// advanced exactly like the main kernel does, so both see the same rotation sequence.
// ------------------------------------------------------------------------------------------
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

}  // namespace sdfthip
