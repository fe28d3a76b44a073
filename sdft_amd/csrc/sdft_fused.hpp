// sdft_fused.hpp -- K3: fused analysis -> operation -> synthesis in the folded form (long calls and hops)
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_forward_rows.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

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

// ------------------------------------------------------------------------------------------
// K3h, two waves per tile: the lone-wave lesson of forward_hop2_kernel applied to the fused hop (a lone wave pays 5-8
// cycles per fp64 instruction whatever its dependencies are).  Wave 0 runs the recurrence and parks (acc, fid) of an
// 8-sample group in a double-buffered LDS image; wave 1 takes the group one barrier later, forms
// alpha * re X + beta * im X, transposes and adds its 64 lanes and writes the per-sample sums of the tile; the workgroup
// that draws the channel's last ticket combines the tiles (both waves) and sets the completion word.
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

}  // namespace sdfthip
