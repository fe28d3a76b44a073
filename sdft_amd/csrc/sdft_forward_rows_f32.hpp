// sdft_forward_rows_f32.hpp -- K1 (row-group form) for 8-byte bins: the analysis kernel of FD float plans (round 4)
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.
//
// forward_rows_kernel<float, 2, ...> is bound by vector-instruction issue, not by HBM: the compiler packs the (re, im) of ONE
// bin into v_pk_* operands, so every product of the complex recurrence computes a half nobody uses and every operand is
// assembled by moves -- 253 vector instructions per wave and sample at N = 4096 (Blackman), 24 GB/s of rows per CU.  That
// is what kept config 3 (m = 4096, FD float, exact carries) at 48 % of the HBM peak: while the relay kernel holds half
// the CUs, the other half cannot store faster than 3 TB/s.
//
// Here the lane's two ADJACENT BINS are the two halves of every packed operand (structure of arrays over the bin pair):
// acc, fid and tw are (re_b0, re_b1), (im_b0, im_b1) register pairs, so
//     acc += fid * delta            2 v_pk_mul + 2 v_pk_add       (sdft.h:583)
//     fid  = fid * tw               4 v_pk_mul + 2 v_pk_add       (sdft.h:584, cmul :283-289)
//     X    = acc * conj(fid)        4 v_pk_mul + 2 v_pk_add       (sdft.h:585)
// is 16 packed instructions per bin pair and sample with no half wasted and no move -- the same operations on the same
// operands in the same order as the scalar formulas, every one rounded on its own (no contraction): bit-identical.
// The window (sdft.h:350-402) runs on the same pairs: the neighbours of the pair (b0, b1) are
//     m2 = (below.b0, below.b1)   m1 = (below.b1, b0)   p1 = (b1, above.b0)   p2 = (above.b0, above.b1)
// so two pairs come from the neighbouring lanes by whole-wave DPP shifts whose fill value -- what lane 0 / lane 63 receive
// -- is the neighbouring wave's edge pair, read from LDS straight into the shifted registers (the edge slots hold
// (re, re, im, im) quads in exactly that layout: the lane that owns an edge pair publishes it with ONE 16-byte write).
// Same row-lockstep geometry, edge-slot protocol, flow-mode waits and chunk grid as forward_rows_kernel.
// Rows of N = 128 * waves * S bins (N a multiple of 128: no mirror lanes inside a wave), dense 16-byte aligned output.

#pragma once

#include "sdft_forward_rows.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

typedef float sdft_f2 __attribute__((ext_vector_type(2)));

// a bin pair of one lane: (b0, b1) halves
struct PairF { sdft_f2 re, im; };

SDFT_D sdft_f2 f2_make(float lo, float hi) { sdft_f2 v; v.x = lo; v.y = hi; return v; }
SDFT_D sdft_f2 f2_splat(float s) { return f2_make(s, s); }

// whole-wave shifts of a pair; `fill` is what the lane without a source lane keeps (the neighbouring wave's edge pair)
SDFT_D sdft_f2 pair_from_below(sdft_f2 fill, sdft_f2 v) { return f2_make(from_below_fill(fill.x, v.x), from_below_fill(fill.y, v.y)); }
SDFT_D sdft_f2 pair_from_above(sdft_f2 fill, sdft_f2 v) { return f2_make(from_above_fill(fill.x, v.x), from_above_fill(fill.y, v.y)); }

// keeps a value out of the reach of the SLP vectoriser (which would re-pack scalar operations into v_pk_* plus the
// v_mov pairs that assemble their operands: a packed instruction costs what two plain ones cost on this VALU --
// profiles/r04_valu_issue_rates.txt -- so packing pays only where no operand has to be moved)
SDFT_D float keep_scalar(float v) { asm volatile("" : "+v"(v)); return v; }

// the window on a bin pair, as the 16 bytes the lane stores (b0.re, b0.im, b1.re, b1.im): c = the pair, m2 = (below.b0,
// below.b1), p2 = (above.b0, above.b1); m1 = (below.b1, b0) and p1 = (b1, above.b0) are read from those.  Operation
// order of window_tap (sdft.h:366-399).  m1 + p1 and the final scaling are plain operations (their operands / results
// are not register pairs), everything between runs packed on the pair.
template <int WIN> SDFT_D sdft_v4f32 window_quad(PairF c, PairF m2, PairF p2, float w)
{
  // m1 + p1 per bin: b0: below.b1 + b1,  b1: b0 + above.b0
  const sdft_f2 s1re = f2_make(keep_scalar(m2.re.y + c.re.y), keep_scalar(c.re.x + p2.re.x));
  const sdft_f2 s1im = f2_make(keep_scalar(m2.im.y + c.im.y), keep_scalar(c.im.x + p2.im.x));
  sdft_f2 tre, tim;
  if constexpr (WIN == WIN_HANN)
  {
    tre = (c.re + c.re) - s1re;
    tim = (c.im + c.im) - s1im;
  }
  else if constexpr (WIN == WIN_HAMMING)
  {
    tre = c.re * 0.54f - s1re * 0.23f;
    tim = c.im * 0.54f - s1im * 0.23f;
  }
  else if constexpr (WIN == WIN_BLACKMAN)
  {
    const sdft_f2 s2re = m2.re + p2.re, s2im = m2.im + p2.im;
    tre = (c.re * 0.42f - s1re * 0.25f) + s2re * 0.04f;
    tim = (c.im * 0.42f - s1im * 0.25f) + s2im * 0.04f;
  }
  else
  {
    tre = c.re; tim = c.im;
  }
  sdft_v4f32 y;
  y.x = keep_scalar(tre.x * w); y.y = keep_scalar(tim.x * w); y.z = keep_scalar(tre.y * w); y.w = keep_scalar(tim.y * w);
  return y;
}

// G: samples per lockstep group (one barrier each)
template <int WIN, int S, int G>
__global__ __launch_bounds__(kWave * kRowWavesMax) void forward_rows_f32_kernel(ForwardArgs<float> a)
{
  constexpr int H = win_halo<WIN>::value;
  constexpr int VW = kRowWavesMax * S;                    // virtual waves
  // edge[side][buf][u][v] = (re(e0), re(e1), im(e0), im(e1)):
  //   side 0 (left):  e0 = first bin of virtual wave v - 2, e1 = first - 1     (what lane 0 receives as `below`)
  //   side 1 (right): e0 = last bin + 1, e1 = last + 2                          (what lane 63 receives as `above`)
  __shared__ sdft_v4f32 edge[2][2][G][VW];

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int nv = nwaves * S;
  unsigned chunk; size_t ch;
  flow_position(a, chunk, ch, blockIdx.x);
  if (!flow_wait(a, chunk, ch)) return;                    // flow mode: the chunk's carries (a time-out ends the workgroup)
  const unsigned kbase = 0u;                               // first bin of this workgroup

  const unsigned span = 2u * a.nbins, maxc = span - 1u;
  const size_t t0 = chunk ? (size_t)chunk * a.chunk_len - a.chunk_shift : 0;
  const size_t tn = (size_t)(chunk + 1) * a.chunk_len - a.chunk_shift;
  const size_t t1 = tn < a.n ? tn : a.n;
  unsigned c = (unsigned)(((size_t)a.cursor0 + t0) % span);

  PairF acc[S], fid[S], tw[S];
  unsigned off_elems[S];                                   // first bin of the lane's pair in slot q
  const size_t cbase = (ch * a.chunks + chunk) * a.nbins;
  // the state a pair starts the chunk with: rotation constant, carry-in, rotation at cursor c
  auto load_pair = [&](unsigned k, PairF& A, PairF& F, PairF& T)
  {
    const sdft_v4f32 t4 = *reinterpret_cast<const sdft_v4f32*>(a.tw + k);
    T.re = f2_make(t4.x, t4.z); T.im = f2_make(t4.y, t4.w);
    const sdft_v4f32 c4 = *reinterpret_cast<const sdft_v4f32*>(a.carry + cbase + k);
    A.re = f2_make(c4.x, c4.z); A.im = f2_make(c4.y, c4.w);
    cx<float> f0, f1;
    if (a.fseed)
    {
      f0 = fid_from_table(a.fseed, a.fseed_L, a.nbins, (long)k, c, cmake<float>(t4.x, t4.y));
      f1 = fid_from_table(a.fseed, a.fseed_L, a.nbins, (long)k + 1, c, cmake<float>(t4.z, t4.w));
    }
    else if (a.seed) { f0 = a.seed[cbase + k]; f1 = a.seed[cbase + k + 1]; }
    else
    {
      f0 = a.wtab[(size_t)(((unsigned long long)k * c) % span)];
      f1 = a.wtab[(size_t)(((unsigned long long)(k + 1) * c) % span)];
    }
    F.re = f2_make(f0.re, f1.re); F.im = f2_make(f0.im, f1.im);
  };
#pragma unroll
  for (int q = 0; q < S; ++q)
  {
    const unsigned k = kbase + (unsigned)(((q * nwaves + wave) * kWave + lane) * 2);
    off_elems[q] = k;
    __builtin_assume(off_elems[q] < (1u << 20));
    load_pair(k, acc[q], fid[q], tw[q]);
  }

  const SDFT_CONSTANT float* d = as_uniform(a.delta + ch * a.n);
  const float w = a.wscale;
  cx<float>* row = a.out + ch * a.out_stride + t0 * (size_t)a.nbins;     // wave-uniform row base

  // one sample of the recurrence on a pair; returns the demodulated pair (sdft.h:583-585 / :572-574)
  auto step = [&](PairF& A, PairF& F, const PairF& T, float dl, bool wrap) -> PairF
  {
    const sdft_f2 dd = f2_splat(dl);
    A.re = A.re + F.re * dd;                                // acc + fid * delta (cscale, cadd)
    A.im = A.im + F.im * dd;
    if (wrap)
    {
      F.re = f2_splat(1.0f); F.im = f2_splat(0.0f);        // :573
      return A;                                             // :574
    }
    const sdft_f2 nr = F.re * T.re - F.im * T.im;           // cmul(fid, tw)
    const sdft_f2 ni = F.re * T.im + F.im * T.re;
    F.re = nr; F.im = ni;
    PairF X;
    // acc * conj(fid): (ar*fr - ai*(-fi), ar*(-fi) + ai*fr) -- the negations are exact, so the same bits as written here
    X.re = A.re * nr + A.im * ni;
    X.im = A.im * nr - A.re * ni;
    return X;
  };

  // edge slots: lane 63 owns the next virtual wave's left edge pair, lane 0 the previous one's right edge pair (one 16-byte
  // write of the lane's own pair); at the two ends of the spectrum the conjugate mirror images (sdft.h:589-595):
  // X[-1] = conj X[1], X[-2] = conj X[2] come from lane 0's b1 and lane 1's b0 of virtual wave 0; X[N] = conj X[N-2],
  // X[N+1] = conj X[N-3] from lane 63's b0 and lane 62's b1 of the last virtual wave.  A lane has at most one role per slot;
  // the destinations are lane constants (slab [buffer 0][sample 0]), so a sample's publish is one or two exec-masked writes.
  constexpr int kSlabU = VW;                               // quads between consecutive samples of a group
  constexpr int kSlabBuf = G * VW;                         // ... between the two buffers
  sdft_v4f32* quad_dst[S];
  float* mir_dst[S];
  bool mir_hi[S];
#pragma unroll
  for (int q = 0; q < S; ++q)
  {
    const int v = q * nwaves + wave;
    quad_dst[q] = nullptr; mir_dst[q] = nullptr; mir_hi[q] = false;
    if constexpr (H >= 1)
    {
      if (lane == kWave - 1 && v + 1 < nv) quad_dst[q] = &edge[0][0][0][v + 1];
      if (lane == 0 && v > 0) quad_dst[q] = &edge[1][0][0][v - 1];
      float* el = reinterpret_cast<float*>(&edge[0][0][0][0]);
      float* er = reinterpret_cast<float*>(&edge[1][0][0][nv - 1]);
      if (v == 0 && lane == 0) { mir_dst[q] = el + 1; mir_hi[q] = true; }                  // X[-1] = conj X[1]
      if (v == 0 && lane == 1) { mir_dst[q] = el + 0; mir_hi[q] = false; }                 // X[-2] = conj X[2]
      if (v == nv - 1 && lane == kWave - 1) { mir_dst[q] = er + 0; mir_hi[q] = false; }    // X[N]   = conj X[N-2]
      if (v == nv - 1 && lane == kWave - 2) { mir_dst[q] = er + 1; mir_hi[q] = true; }     // X[N+1] = conj X[N-3]
    }
  }
  auto publish = [&](const PairF (&x)[S], int buf, int u)
  {
    if constexpr (H >= 1)
    {
      const int slab = buf * kSlabBuf + u * kSlabU;
#pragma unroll
      for (int q = 0; q < S; ++q)
      {
        if (quad_dst[q])
        {
          sdft_v4f32 quad; quad.x = x[q].re.x; quad.y = x[q].re.y; quad.z = x[q].im.x; quad.w = x[q].im.y;
          quad_dst[q][slab] = quad;
        }
        if (mir_dst[q])
        {
          float* m = mir_dst[q] + 4 * slab;
          m[0] = mir_hi[q] ? x[q].re.y : x[q].re.x;
          m[2] = -(mir_hi[q] ? x[q].im.y : x[q].im.x);
        }
      }
    }
  };

  // phase B of one group: window and store, pair by pair; the edge quads of the NEXT pair are requested before the current
  // one is worked on (after the barrier every wave of the SIMD would otherwise sit through the same LDS latency together)
  auto left_quad = [&](int buf, int u, int v, size_t) -> sdft_v4f32 { return edge[0][buf][u][v]; };
  auto right_quad = [&](int buf, int u, int v, size_t) -> sdft_v4f32 { return edge[1][buf][u][v]; };
  auto finish_group = [&](const PairF (&x)[G][S], int buf, int m, size_t tl)
  {
    sdft_v4f32 l = {}, r = {};
    if constexpr (H >= 1) { l = left_quad(buf, 0, wave, tl); r = right_quad(buf, 0, wave, tl); }
#pragma unroll
    for (int u = 0; u < G; ++u)
    {
      if (u < m)
      {
#pragma unroll
        for (int q = 0; q < S; ++q)
        {
          sdft_v4f32 ln = {}, rn = {};
          if constexpr (H >= 1)
          {
            const int un = (q + 1 < S) ? u : u + 1, qn = (q + 1 < S) ? q + 1 : 0;
            if (un < G) { ln = left_quad(buf, un, qn * nwaves + wave, tl); rn = right_quad(buf, un, qn * nwaves + wave, tl); }   // (past m: stale, unused)
          }
          PairF m2 = x[u][q], p2 = x[u][q];
          if constexpr (H >= 1)
          {
            m2.re = pair_from_below(f2_make(l.x, l.y), x[u][q].re);
            m2.im = pair_from_below(f2_make(l.z, l.w), x[u][q].im);
            p2.re = pair_from_above(f2_make(r.x, r.y), x[u][q].re);
            p2.im = pair_from_above(f2_make(r.z, r.w), x[u][q].im);
          }
          store_vec(reinterpret_cast<sdft_v4f32*>(row + off_elems[q]), window_quad<WIN>(x[u][q], m2, p2, w));
          l = ln; r = rn;
        }
        row += a.nbins;
      }
    }
  };

  int buf = 0;
  size_t t = t0;
  // the differences of the next group are requested a group ahead (past the call's end: the workspace has slack)
  float dnext[G];
#pragma unroll
  for (int u = 0; u < G; ++u) dnext[u] = d[t + u];
  while (t < t1)                       // all waves of the group take identical trip counts
  {
    const int m = (t1 - t < (size_t)G) ? (int)(t1 - t) : G;
    PairF xs[G][S];
    float dl[G];
#pragma unroll
    for (int u = 0; u < G; ++u) dl[u] = dnext[u];
#pragma unroll
    for (int u = 0; u < G; ++u) dnext[u] = d[t + G + u];
    if (m == G && c + G <= maxc)
    {
#pragma unroll
      for (int u = 0; u < G; ++u)
      {
#pragma unroll
        for (int q = 0; q < S; ++q) xs[u][q] = step(acc[q], fid[q], tw[q], dl[u], false);
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
          const bool wrap = (c == maxc);
#pragma unroll
          for (int q = 0; q < S; ++q) xs[u][q] = step(acc[q], fid[q], tw[q], dl[u], wrap);
          c = wrap ? 0 : c + 1;
          publish(xs[u], buf, u);
        }
      }
    }
    __syncthreads();
    finish_group(xs, buf, m, t - t0);
    t += m;
    buf ^= 1;
  }

  if (chunk + 1 == a.chunks)
  {
#pragma unroll
    for (int q = 0; q < S; ++q)
    {
      sdft_v4f32 sa, sf;
      sa.x = acc[q].re.x; sa.y = acc[q].im.x; sa.z = acc[q].re.y; sa.w = acc[q].im.y;
      sf.x = fid[q].re.x; sf.y = fid[q].im.x; sf.z = fid[q].re.y; sf.w = fid[q].im.y;
      *reinterpret_cast<sdft_v4f32*>(a.acc_state + ch * a.nbins + off_elems[q]) = sa;
      *reinterpret_cast<sdft_v4f32*>(a.fid_state + ch * a.nbins + off_elems[q]) = sf;
    }
  }
  signal_done_workgroup(a.done);
}

}  // namespace sdfthip
