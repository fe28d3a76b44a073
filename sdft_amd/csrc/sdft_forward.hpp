// sdft_forward.hpp -- K1: window, forward arguments, flow-mode waits, self-carried chunks, the independent-tile kernel
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_carry_exact.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

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
  // Which (channel, chunk) a workgroup takes.  The dispatcher deals consecutive workgroups to different XCDs (b and b + 8 share
  // one); with xcd_map every XCD takes a CONTIGUOUS eighth of the launch's (channel, chunk) sequence -- of the matrix -- instead
  // of every eighth region: the store stream of the headline workload runs 4-9 % faster that way (store-only probes,
  // profiles/r05_store_ceiling_study.txt).  Placement is for speed only: any block-to-XCD assignment gives the same results.
  unsigned xcd_map;           // 0, or the launch's number of (channel, chunk) workgroups
  unsigned inv_chunks, inv_channels;   // floor(2^32 / launch_chunks) + 1 and floor(2^32 / ready_channels) + 1 (0 for a divisor of 1): flow_position
};
// the b-th workgroup's position in the launch's (channel, chunk) sequence: a bijection of [0, total) for any total
SDFT_D unsigned xcd_contiguous(unsigned b, unsigned total)
{
  const unsigned q = total >> 3, r = total & 7u, x = b & 7u;
  return x * q + (x < r ? x : r) + (b >> 3);
}

// Flow mode: which (chunk, channel) a workgroup takes, and the wait for the chunk's carries.  The relay kernel stores
// carries write-through (sc1), waits for them, then stores the flag (sc1); here: relaxed agent-scope polls of the flags,
// one agent-scope acquire, then plain loads (MI355X_MICROARCH.md, inter-workgroup visibility, form R1).
// block / d and block % d for a wave-uniform block and a divisor the host knows (inv = floor(2^32 / d) + 1, 0 for d = 1): scalar
// integer instructions only -- the compiler's own sequence for a division by a run-time value goes through the vector unit's
// reciprocal and keeps vector registers alive for it (these kernels have none to spare)
SDFT_D void uniform_divmod(unsigned block, unsigned d, unsigned inv, unsigned& q, unsigned& r)
{
  if (inv == 0) { q = block; r = 0; return; }
  q = __builtin_amdgcn_readfirstlane((unsigned)(((unsigned long long)block * inv) >> 32));
  r = block - q * d;
  if (r >= d) { --q; r += d; }                               // (floor(2^32 / d) + 1 over-estimates by at most one)
}
template <typename FD> SDFT_D void flow_position(const ForwardArgs<FD>& a, unsigned& chunk, size_t& ch, unsigned block)
{
  unsigned q, r;
  if (a.ready) { uniform_divmod(block, a.ready_channels, a.inv_channels, q, r); chunk = a.chunk0 + q; ch = r; }
  else
  {
    if (a.xcd_map) block = xcd_contiguous(block, a.xcd_map);
    uniform_divmod(block, a.launch_chunks, a.inv_chunks, q, r);
    chunk = a.chunk0 + r; ch = q;
  }
}
template <typename FD> SDFT_D void flow_position(const ForwardArgs<FD>& a, unsigned& chunk, size_t& ch) { flow_position(a, chunk, ch, blockIdx.x); }
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
  if (chunk + 1 == a.chunks && sa.hist_out)
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
// no effect on a pure write stream -- and removed; round 5, store-only kernels at 7 TB/s: 1-3 % slower,
// profiles/r05_nt_stores.txt)
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

}  // namespace sdfthip
