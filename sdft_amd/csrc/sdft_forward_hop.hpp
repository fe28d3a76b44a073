// sdft_forward_hop.hpp -- K1h: calls of one time chunk (hop-wise streaming, single samples): differences + analysis in one launch
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_forward.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

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
  unsigned parts, part_len;   // forward_hop2_kernel: the call's samples in `parts` time parts of part_len samples (1, n: no split)
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
// Time parts (round 5).  The 100 samples of a hop were one dependent sequence per tile: 100 x 22 instructions of the slower
// wave at the 5-7 cycles a lone wave pays per fp64 instruction.  The call's samples are now cut into up to 8 parts, every
// (tile, part) a workgroup on a CU of its own; the recurrence wave of part p first runs the stream state through the
// samples before its part -- the same acc / fid operations in the same order as the reference (advance_normal / advance_wrap:
// 10 instructions per sample, no demodulation, no window, no store) -- and so starts its part with exactly the state the
// serial pass would have there.  Nothing is approximated and nothing is exchanged: every row is still the reference's, bit
// for bit, and the last part's workgroups write the state.  Critical path per tile: n x 22 -> about n x (10 + 12 / parts).
// ------------------------------------------------------------------------------------------
// (the body is a device function of (arguments, workgroup number, workgroups of the call): forward_hop2_kernel runs it once per launch, the resident
// kernel of sdft_resident.hpp once per doorbell)
template <typename TD, typename FD, int BPL, int WIN, bool ROWS>
SDFT_D void forward_hop2_body(const HopArgs<TD, FD>& a, const unsigned block, const unsigned blocks)
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
  const unsigned tile = block % a.tiles;
  const unsigned part = (block / a.tiles) % a.parts;
  const size_t ch = block / (a.tiles * a.parts);
  const size_t p0 = (size_t)part * a.part_len;                                   // this workgroup's samples: [p0, p1)
  const size_t p1 = (p0 + a.part_len < a.n) ? p0 + a.part_len : a.n;

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
    for (size_t i = ((size_t)part * a.tiles + tile) * (2 * kWave) + threadIdx.x; i < span; i += (size_t)a.tiles * a.parts * (2 * kWave))
    {
      const size_t j = a.n + i;
      ho[i] = (j >= span) ? xv[j - span] : hv[j];
    }
    // differences of the call's samples up to the end of this part (sdft.h:564), the subtraction in TD precision
    for (size_t tt = threadIdx.x; tt < p1; tt += 2 * kWave)
    {
      const TD cur = xv[tt];
      const TD old = (tt < span) ? hv[tt] : xv[tt - span];
      diff_lds[tt] = cur - old;
    }
  }

  const size_t groups = (p1 - p0 + G - 1) / G;
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
    // the state at the start of this part: the reference's own acc / fid operations over the samples before it
    for (size_t tt = 0; tt < p0; tt += G)
    {
      TD dd[G];
#pragma unroll
      for (int u = 0; u < G; ++u) dd[u] = diff_lds[tt + u];  // (a group's reads in one round trip; cells past p0 are staged too: p0 < p1)
      if (tt + G <= p0 && c + G <= maxc)
      {
#pragma unroll
        for (int u = 0; u < G; ++u)
        {
#pragma unroll
          for (int b = 0; b < BPL; ++b) advance_normal(s[b], (FD)dd[u]);
        }
        c += G;
      }
      else
      {
#pragma unroll
        for (int u = 0; u < G; ++u)
          if (tt + u < p0)
          {
            const bool wrap = (c == maxc);                   // wave-uniform
#pragma unroll
            for (int b = 0; b < BPL; ++b) { if (wrap) advance_wrap(s[b], (FD)dd[u]); else advance_normal(s[b], (FD)dd[u]); }
            c = wrap ? 0 : c + 1;
          }
      }
    }
    SDFT_HOP2_STAMP(3);
    int buf = 0;
    for (size_t g = 0; g < groups; ++g)
    {
      const size_t t = p0 + g * G;
      const int m = (p1 - t < (size_t)G) ? (int)(p1 - t) : G;
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
    if (part + 1 == a.parts)                                 // the state after the call's last sample
    {
#pragma unroll
      for (int b = 0; b < BPL; ++b)
        if (keep[b])
        {
          a.acc_out[sbase + kfirst + b] = s[b].acc;
          a.fid_out[sbase + kfirst + b] = s[b].fid;
        }
    }
#ifdef SDFT_HOP_STAMPS
    if (a.stamps && block == blocks - 1 && lane == 0) for (int i = 0; i < 4; ++i) a.stamps[i] = stamp[i];       // (the last part: the longest way)
#endif
  }
  else
  {
    // ---------------- window + stores ----------------
    bool keep[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b) { const long k = kfirst + b; keep[b] = owner && k >= 0 && k < nbins; }
    const FD w = a.wscale;
    cx<FD>* row = a.out + ch * a.out_stride + p0 * (size_t)a.nbins;
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
      const size_t t = p0 + g * G;
      const int m = (p1 - t < (size_t)G) ? (int)(p1 - t) : G;
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
    if (a.stamps && block == blocks - 1 && lane == 0) for (int i = 0; i < 3; ++i) a.stamps[4 + i] = stamp[i];
#endif
  }
  // completion word: both waves' stores are out before one lane reports
  signal_done_workgroup(a.done);
  (void)blocks;
}
template <typename TD, typename FD, int BPL, int WIN, bool ROWS>
__global__ __launch_bounds__(2 * kWave) void forward_hop2_kernel(HopArgs<TD, FD> a)
{
  forward_hop2_body<TD, FD, BPL, WIN, ROWS>(a, blockIdx.x, gridDim.x);
}

}  // namespace sdfthip
