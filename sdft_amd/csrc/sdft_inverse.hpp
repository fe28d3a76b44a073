// sdft_inverse.hpp -- K2: synthesis (exact order streaming, row form, tree sum with the rounding-interval proof) and row rewriting
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_fused.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

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

// a bin of the matrix; nt: non-temporal load (the matrix is read once: it should not push what the analysis left in the
// Infinity Cache -- dirty lines of the same matrix's tail -- out to HBM)
template <typename FD> SDFT_D cx<FD> load_bin(const cx<FD>* p, int nt)
{
  if (nt)
  {
    using V = typename StoreVec<FD, 1>::type;
    const V q = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    return cmake<FD>(q.x, q.y);
  }
  return *p;
}
template <typename V> SDFT_D V load_vec(const V* p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }

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
  int nt;                     // loads of the matrix are non-temporal (streamed past the caches: see Plan::opt_inverse_nt)
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
      const FD tv = synth_term<FD, LAT1, OPS>(load_bin(row + k, a.nt), k, a.op, a.syn, a.nbins, grow);
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
// RPI: rows per load instruction = 4 (a row segment of 256 bytes per instruction), 2 (512 bytes) or 1 (a whole KiB of one row)
template <typename TD, typename FD, bool LAT1, int RW, int DEPTH, bool OPS = false, int RPI = 4>
__global__ __launch_bounds__(kBlock) void inverse_exact_kernel(InverseArgs<TD, FD> a)
{
  constexpr int BPL = 16 / (int)sizeof(cx<FD>);          // bins per 16-byte load (1 for f64, 2 for f32)
  constexpr int LPR = kWave / RPI;                       // lanes per row segment
  constexpr int C = LPR * BPL;                           // bins per tile row = 256 bytes (RPI = 4) ... 1 KiB (RPI = 1)
  constexpr int NI = RW / RPI;                           // load instructions per tile
  using V = typename StoreVec<FD, (sizeof(cx<FD>) == 16 ? 1 : 2)>::type;   // 16-byte vector
  __shared__ FD tile[kWavesPerBlock][RW][C + 1];

  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t ngroups_per_ch = (a.n + RW - 1) / RW;
  const size_t ngroups = ngroups_per_ch * a.channels;
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  const int sub = lane / LPR, seg = lane % LPR;          // load phase: row within the instruction, 16-byte slot
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
            const V q = load_vec(reinterpret_cast<const V*>(rowp + k), a.nt);
            v[i][0] = cmake<FD>((FD)q[0], (FD)q[1]);
            if constexpr (BPL == 2) v[i][1] = cmake<FD>((FD)q[2], (FD)q[3]);
          }
          else
          {
#pragma unroll
            for (int b = 0; b < BPL; ++b)
              if (k + b < a.nbins) v[i][b] = load_bin(rowp + k + b, a.nt);
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
