// sdft_forward_rows.hpp -- K1 (row-group form): the dominant kernel; SYN != 0: fused synthesis on the windowed rows
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_ops.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

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
  // (double samples: 8 rows of the fold in flight instead of 16 -- with 16 the 128-register kernel spills 48 of them)
  if constexpr (SELF) dft = self_carry<2, (sizeof(TD) == 8 ? 8 : 16)>(sa, a, cells, chunk, ch, t0);

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
        s[q][b].fid = a.wtab[((unsigned)kk * c) % span];
      }
      else
      {
      s[q][b].acc = a.carry[cbase + kk];
      s[q][b].fid = a.fseed ? fid_from_table(a.fseed, a.fseed_L, a.nbins, kk, c, s[q][b].tw)
                  : a.seed  ? a.seed[cbase + kk] : a.wtab[((unsigned)kk * c) % span];
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
          if (!same_bits(ylo, yhi))                                         // wave-uniform: every lane holds the same sums
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

  if (chunk + 1 == a.chunks && a.acc_state)                // (pipelined calls: self_state_kernel has written the state already)
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

// Pipelined calls: the plan's state AFTER a whole call, computed ahead of the call's rows -- the carry a chunk that started
// at sample n would derive for itself (fold of all n samples + one FFT in LDS, self_carry), the rotation from the table,
// the delay line.  One workgroup per channel, a few microseconds; with it the row kernels of consecutive calls depend on
// this chain of small kernels only, not on each other, and run on two streams: the next call's workgroups take the CUs
// the moment the previous call's leave them (Plan::forward_self).
template <typename TD, typename FD>
__global__ __launch_bounds__(kWave * kRowWavesMax) void self_state_kernel(ForwardArgs<FD> a, SelfArgs<TD, FD> sa)
{
  extern __shared__ __align__(16) unsigned char rows_dyn_lds[];
  cx<FD>* cells = reinterpret_cast<cx<FD>*>(rows_dyn_lds);
  const size_t ch = blockIdx.x;
  const unsigned span = 2u * a.nbins;
  // (as the last chunk's workgroup: that one writes the delay line; "its chunk" starts where the call ends)
  const cx<FD>* dft = self_carry<2, (sizeof(TD) == 8 ? 8 : 16)>(sa, a, cells, a.chunks - 1u, ch, a.n);
  const unsigned c = (unsigned)(((size_t)a.cursor0 + a.n) % span);
  for (unsigned k = threadIdx.x; k < a.nbins; k += blockDim.x)
  {
    cx<FD> acc = sa.acc_in[ch * a.nbins + k];
    if (dft) acc = cadd(acc, dft[self_slot(sa, k)]);
    a.acc_state[ch * a.nbins + k] = acc;
    a.fid_state[ch * a.nbins + k] = a.wtab[(size_t)(((unsigned long long)k * c) % span)];
  }
}

// One wave that stays for `ticks` of the 100 MHz wall clock: Plan::ensure_pipe launches one on each of its streams and looks
// whether they ran at the same time -- streams that share a hardware queue run one after the other.
template <typename FD> __global__ __launch_bounds__(kWave) void queue_probe_kernel(unsigned long long ticks)   // (a template: one definition per translation unit)
{
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace sdfthip
