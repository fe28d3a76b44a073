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
// (through the row-pointer table or the dense base, the pointer is GENERIC and these are FLAT loads, which count as LDS operations too: a wave
// of inverse_exact_kernel that waits for its LDS tile also waits for the matrix loads it has on its way.  Round 6 tried global loads here
// (an address-space cast): the tiles' forms got SLOWER, 6.3 -> 5.8 TB/s at n = 1e6 x 1024 double -- fewer loads on their way per wave is what
// these forms want (scripts/inverse_forms_probe.hip) -- so the loads stay as they are.)
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
// (nt a run-time value: two loads behind a branch.  With a CONSTANT nt the select folds to an ordinary load -- inverse_rows_ordered_kernel
// calls the builtin itself)
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
  size_t nt_skip;             // ... but the first nt_skip rows READ (the matrix' end: what an analysis wrote last) take ordinary loads all the same:
                              // non-temporal loads of lines that sit dirty in the Infinity Cache are slow, ordinary loads of them are not and
                              // push the rest of what is dirty there out on the way (Plan::opt_inverse_nt_skip_mb)
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
    const int nt = (a.nt && ri >= a.nt_skip) ? 1 : 0;      // (wave-uniform)
    const FD* grow = OPS ? gain_row(a.op, t, a.nbins) : nullptr;
#pragma unroll 4
    for (unsigned k = lane; k < a.nbins; k += kWave)
    {
      const FD tv = synth_term<FD, LAT1, OPS>(load_bin(row + k, nt), k, a.op, a.syn, a.nbins, grow);
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
      if (!same_bits(ylo, yhi))                            // wave-uniform (every lane holds the wave's sums)
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
// K2 (rows in step; round 5)  float samples from double bins, long calls.  The streaming forms read the matrix at 6.1 TB/s;
// a load-only kernel whose workgroups each read WHOLE ROWS of a contiguous chunk of the matrix, one after the other, with
// non-temporal 16-byte loads reaches 6.7-7.0 (scripts/load_forms_probe.hip, profiles/r05_load_forms.txt) -- the shape the
// analysis writes in.  So here a workgroup of N/64 waves owns a chunk of consecutive rows, every lane ONE bin of every row
// (its synthesis twiddle stays in registers), G rows in flight; the terms of a row meet in a tree sum -- lanes, then the
// eight sums of a group (G sums, G sums of magnitudes) transposed through the wave in 10 exchanges instead of 48, then
// the waves' partial sums through LDS -- and the rounding-interval proof of inverse_kernel<VERIFY> decides the sample:
// the tree sum and 2.5e-16 * (N + 64) * sum|term| bound the reference's ordered sum (sdft.h:641-651) whatever the tree; when
// both ends of the interval round to the same float that float is the reference's, else the row is read again and added
// in ascending bin order by one wave.  Same bits as every other form.
// ------------------------------------------------------------------------------------------
// J: bins per lane (1: rows of up to 1024 bins, 64 registers, two workgroups of 16 waves to a CU; 2: up to 2048 bins)
template <typename TD, typename FD, bool LAT1, int J, int G = 4>
SDFT_D void inverse_rows_body(const InverseArgs<TD, FD>& a, unsigned chunk_len)
{
  static_assert(sizeof(TD) == 4 && sizeof(FD) == 8, "the interval test needs a rounding to hide behind");
  static_assert(LAT1, "latency 1 only: the term of a bin is +-re, one register per bin in flight");
  static_assert(G == 4, "the transposed reduction below is written for four rows");
  static_assert(J == 1 || J == 2, "one or two bins per lane");
  constexpr int SG = 16;                                   // groups between two barriers (64 rows: the waves drift that far apart)
  __shared__ FD red[2][SG * G][2][kRowWavesMax];           // [buffer][row of the super-group][sum | sum of magnitudes][wave]
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const size_t rows = (size_t)a.channels * a.n;
  // chunks are taken from the END of the matrix first (what the analysis wrote last still sits in the Infinity Cache)
  const size_t nchunks = (rows + chunk_len - 1) / chunk_len;
  const size_t cidx = nchunks - 1 - blockIdx.x;
  const int nt = (a.nt && (size_t)blockIdx.x * chunk_len >= a.nt_skip) ? 1 : 0;
  const size_t r0 = cidx * (size_t)chunk_len;
  const size_t r1 = r0 + chunk_len < rows ? r0 + chunk_len : rows;

  unsigned kb[J]; bool live[J];
  FD sgn[J];
#pragma unroll
  for (int j = 0; j < J; ++j)
  {
    kb[j] = threadIdx.x + (unsigned)j * blockDim.x;
    live[j] = kb[j] < a.nbins;
    sgn[j] = (kb[j] & 1u) ? (FD)(-1) : (FD)(+1);
  }
  auto row_of = [&](size_t r) -> const cx<FD>* { const size_t ch = r / a.n, t = r - ch * a.n; return a.in + ch * a.in_stride + t * (size_t)a.nbins; };
  auto fetch = [&](size_t r, cx<FD> (&x)[G][J])
  {
#pragma unroll
    for (int g = 0; g < G; ++g)
    {
      const size_t rr = r + g < r1 ? r + g : r1 - 1;       // (past the chunk: the last row again, its sum is not used)
      const cx<FD>* row = row_of(rr);
#pragma unroll
      for (int j = 0; j < J; ++j) x[g][j] = live[j] ? load_bin(row + kb[j], nt) : cmake<FD>((FD)0, (FD)0);
    }
  };
  auto term = [&](const cx<FD>& v, int j) -> FD { return v.re * sgn[j]; };             // sdft.h:643

  cx<FD> xa[G][J], xb[G][J], xc[G][J];                     // a ring of three groups: two in flight while one is summed
  int buf = 0;
  unsigned slot = 0;                                       // group of the super-group
  // the sums of the group at row r (in x) and the samples they decide
  auto finish = [&](size_t r, const cx<FD> (&x)[G][J])
  {
    FD v[2 * G];
#pragma unroll
    for (int g = 0; g < G; ++g)
    {
      const FD t0 = term(x[g][0], 0);                     // (a bin past N-1 holds +0)
      v[g] = t0; v[G + g] = __builtin_fabs(t0);
      if constexpr (J == 2) { const FD t1 = term(x[g][1], 1); v[g] += t1; v[G + g] += __builtin_fabs(t1); }
    }
    // eight sums over the 64 lanes, transposed: after the exchange with lane ^ 32 a lane keeps four of them, after ^ 16 two,
    // after ^ 8 one -- value (lane >> 3) -- which the last three exchanges complete
    FD w4[4], w2[2], w1;
    {
      const bool hi = (lane & 32) != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const FD send = hi ? v[i] : v[i + 4], keep = hi ? v[i + 4] : v[i]; w4[i] = keep + __shfl_xor(send, 32, 64); }
    }
    {
      const bool hi = (lane & 16) != 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) { const FD send = hi ? w4[i] : w4[i + 2], keep = hi ? w4[i + 2] : w4[i]; w2[i] = keep + __shfl_xor(send, 16, 64); }
    }
    {
      const bool hi = (lane & 8) != 0;
      const FD send = hi ? w2[0] : w2[1], keep = hi ? w2[1] : w2[0];
      w1 = keep + __shfl_xor(send, 8, 64);
    }
    w1 += __shfl_xor(w1, 4, 64);
    w1 += __shfl_xor(w1, 2, 64);
    w1 += __shfl_xor(w1, 1, 64);
    // value index = lane >> 3 (bits 5, 4, 3 chose i of 4, i of 2, i of 1): 0 .. 3 the sums of rows 0 .. 3, 4 .. 7 their magnitudes
    if ((lane & 7) == 0) red[buf][slot * G + ((lane >> 3) & 3)][lane >> 5][wave] = w1;
    ++slot;
    if (slot < (unsigned)SG && r + G < r1) return;
    // ---- every SG groups (and at the chunk's end): the waves' partial sums meet ----
    __syncthreads();
    const unsigned nrows = slot * G;                       // rows of this super-group (the last group may reach past the chunk)
    const size_t rs = r + G - nrows;                       // its first row
    for (unsigned q = (unsigned)wave; q < nrows; q += (unsigned)nwaves)
    {
      const size_t rr = rs + q;
      if (rr >= r1) break;
      // lanes 0 .. 15: the waves' sums of the row, lanes 16 .. 31: their magnitudes (waves past nwaves: +0)
      const int wsrc = lane & 15;
      FD qv = (lane < 32 && wsrc < nwaves) ? red[buf][q][lane >> 4][wsrc] : (FD)0;
      qv += __shfl_xor(qv, 8, 64); qv += __shfl_xor(qv, 4, 64); qv += __shfl_xor(qv, 2, 64); qv += __shfl_xor(qv, 1, 64);
      const FD sum = __shfl(qv, 0, 64), all = __shfl(qv, 16, 64);
      const FD e = all * ((FD)2.5e-16 * (FD)(a.nbins + 2 * kWave));
      const TD ylo = (TD)((sum - e) * a.sweight), yhi = (TD)((sum + e) * a.sweight);
      TD out = ylo;
      if (!same_bits(ylo, yhi))                            // wave-uniform
      {
        const cx<FD>* row = row_of(rr);
        FD ordered = (FD)0;
        for (unsigned k0 = 0; k0 < a.nbins; k0 += kWave)
        {
          const unsigned k = k0 + (unsigned)lane;
          FD tv = (FD)0;
          if (k < a.nbins)
          {
            tv = row[k].re * ((k & 1u) ? (FD)(-1) : (FD)(+1));
          }
          const int lo = __double2loint(tv), hi2 = __double2hiint(tv);
          const unsigned cnt = a.nbins - k0 < (unsigned)kWave ? a.nbins - k0 : (unsigned)kWave;
          for (unsigned jj = 0; jj < cnt; ++jj)            // sdft.h:641-651: one accumulator, ascending bins
            ordered += __hiloint2double(__builtin_amdgcn_readlane(hi2, (int)jj), __builtin_amdgcn_readlane(lo, (int)jj));
        }
        out = (TD)(ordered * a.sweight);
      }
      if (lane == 0) { const size_t ch = rr / a.n, t = rr - ch * a.n; a.y[ch * a.y_stride + t] = out; }
    }
    slot = 0;
    buf ^= 1;                                              // (the other buffer is written next: its readers pass the next barrier first)
  };
  fetch(r0, xa);
  if (r0 + G < r1) fetch(r0 + G, xb);
  for (size_t r = r0; r < r1; r += 3 * G)                  // (all quantities wave-uniform and the same in every wave)
  {
    if (r + 2 * G < r1) fetch(r + 2 * G, xc);
    finish(r, xa);
    if (r + G >= r1) break;
    if (r + 3 * G < r1) fetch(r + 3 * G, xa);
    finish(r + G, xb);
    if (r + 2 * G >= r1) break;
    if (r + 4 * G < r1) fetch(r + 4 * G, xb);
    finish(r + 2 * G, xc);
  }
}

// (one bin per lane: held to 64 registers, so that two workgroups of 16 waves share a CU -- 8 waves per SIMD)
template <typename TD, typename FD, bool LAT1>
__global__ __launch_bounds__(kWave * kRowWavesMax) __attribute__((amdgpu_waves_per_eu(8, 8))) void inverse_rows1_kernel(InverseArgs<TD, FD> a, unsigned chunk_len)
{
  inverse_rows_body<TD, FD, LAT1, 1>(a, chunk_len);
}
template <typename TD, typename FD, bool LAT1>
__global__ __launch_bounds__(kWave * kRowWavesMax) void inverse_rows2_kernel(InverseArgs<TD, FD> a, unsigned chunk_len)
{
  inverse_rows_body<TD, FD, LAT1, 2>(a, chunk_len);
}

// ------------------------------------------------------------------------------------------
// K2 (whole rows, ordered sum; round 6)  the reference's summation order (sdft.h:641-651) for EVERY type pair -- no rounding-interval proof,
// no second pass -- read the way inverse_rows_body reads: a workgroup owns a chunk of consecutive rows and fetches WHOLE rows, 1 KiB per load
// instruction, non-temporal.  The tiles of inverse_exact_kernel below read 256-byte pieces of 16 ... 32 rows per wave: 5.8 ... 6.35 TB/s at
// n = 1e6 x 1024 where this kernel reads 6.8 ... 6.9 for every type pair (rows in step with the tree sum and the proof: 6.54;
// profiles/r06_synthesis_forms.txt).  Eleven LOADER waves turn the bins into the scalars the reference adds and park them in LDS, row by row
// (row slots of N terms; NG groups of G slots); ONE wave adds: lane j owns row slot j and walks it in ascending bin order, 128 bins per pass,
// every group of slots at its own place in its rows -- a group is handed over when its last loader has written (counter `filled`) and handed
// back when its sums are stored (counter `freed`), so the loaders run up to NG groups ahead in LDS and three more in registers while the
// chains of N dependent additions run 16 ... 64 abreast.  Rows of up to 8 KiB of terms (1024 double bins, 2048 float bins): with fewer than
// 16 slots the adding wave cannot keep up with HBM (measured: 12 ... 14 cycles of the nominal clock per term in the kernel, 6.5 float / 9 double for
// a lone wave on an idle chip -- scripts/adding_wave_probe.hip; giving the adding wave its SIMD to itself changes nothing: 2048 float bins, 16 slots,
// run at 5.6 ... 5.8 TB/s either way, at the adding wave's pace).
// What the first versions lost, in the order found (each is a comment at its place below): a load whose other half nobody reads gives that
// half's registers away and is waited for at once; a load behind a condition, a polling loop inside the ring's loop or a fetch at the loop's
// top make the compiler wait for ALL loads that are on their way; a 64-bit division per load; and `nt ? __builtin_nontemporal_load(p) : *p`
// with a constant nt compiles to an ORDINARY load -- 6.2 TB/s instead of 6.85.
// ------------------------------------------------------------------------------------------
template <typename FD, int NLOAD = 11> struct ordered_rows_geometry   // (NLOAD: scripts/inverse_forms_probe.hip tries 7, 8, 11 and 15 loader waves)
{
  static constexpr int BPL = 16 / (int)sizeof(cx<FD>);     // bins per 16-byte load
  static constexpr int PIECE = kWave * BPL;                // bins per load instruction of a wave (1 KiB of a row)
  static constexpr int LOADERS = NLOAD;                    // waves 1 .. 11 (wave 0 adds); twelve waves: 168 registers each (sixteen: 128, and the adding wave spills)
  static constexpr int MAXP = (64 + NLOAD - 1) / NLOAD;    // pieces of a group per loader wave (G * ppr <= 64 <= LOADERS * MAXP)
  static constexpr unsigned kFlagBytes = 512;              // filled[64], freed[64]
  static constexpr int kBlockBins = 128;                   // the adding wave's pass: a row slot is a whole number of these (bins past N-1 hold +0)
  unsigned ppr, stride_bytes, G, NG;                       // pieces per row, bytes from a row slot to the next, rows per group, groups
  size_t lds_bytes;
  SDFT_HD bool make(unsigned nbins, size_t lds_budget)
  {
    ppr = (nbins + PIECE - 1) / PIECE;
    while ((ppr * PIECE) % kBlockBins) ++ppr;
    const size_t row_bytes = (size_t)ppr * PIECE * sizeof(FD);
    if (nbins < 1 || row_bytes > 8192) return false;       // (16 KiB rows, 8 slots: the adding wave sets the pace -- 3.1 TB/s at N = 4096 float, 5.5 at 2048 double)
    stride_bytes = (unsigned)row_bytes + 16;               // (lanes of the adding wave read the same bin of different rows: 16 bytes apart in the banks)
    size_t slots = (lds_budget - kFlagBytes - 256) / stride_bytes;
    if (slots > 64) slots = 64;
    if (slots < 16) return false;
    // rows per group: at most 64 pieces to a group, at least four groups, as many slots in use as can be (19 fit beside 8 KiB rows: 6 x 3)
    unsigned gmax = 64 / ppr; if (gmax > slots / 4) gmax = (unsigned)(slots / 4);
    G = gmax;
    for (unsigned g = gmax; g >= 1 && 4 * g >= 3 * gmax; --g) if ((slots / g) * g > (slots / G) * G) G = g;   // (small groups cost more than slots gain: 17 x 1 ran at 4.6 TB/s)
    NG = (unsigned)(slots / G);
    if (NG > 64) NG = 64;
    lds_bytes = kFlagBytes + (size_t)NG * G * stride_bytes + 256;   // (slack: the adding wave fetches one stage past a row's end)
    return true;
  }
};

template <typename TD, typename FD, bool LAT1, int NLOAD = 11>
__global__ __launch_bounds__(kWave * (NLOAD + 1)) void inverse_rows_ordered_kernel(InverseArgs<TD, FD> a, unsigned chunk_len, ordered_rows_geometry<FD, NLOAD> geo)
{
  using GEO = ordered_rows_geometry<FD, NLOAD>;
  constexpr int BPL = GEO::BPL, PIECE = GEO::PIECE, LOADERS = GEO::LOADERS, MAXP = GEO::MAXP;
  using V = typename StoreVec<FD, (sizeof(cx<FD>) == 16 ? 1 : 2)>::type;            // 16 bytes of the matrix
  using TV = typename StoreVec<FD, (sizeof(FD) == 8 ? 1 : 2)>::type;               // 16 bytes of terms (2 double / 4 float)
  extern __shared__ __align__(16) unsigned char ordered_lds[];
  unsigned* filled = reinterpret_cast<unsigned*>(ordered_lds);
  unsigned* freed = filled + 64;
  unsigned char* slots = ordered_lds + GEO::kFlagBytes;

  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t rows = (size_t)a.channels * a.n;
  // chunks are taken from the END of the matrix first (what the analysis wrote last still sits in the Infinity Cache)
  const size_t nchunks = (rows + chunk_len - 1) / chunk_len;
  const size_t cidx = nchunks - 1 - blockIdx.x;
  const int nt = (a.nt && (size_t)blockIdx.x * chunk_len >= a.nt_skip) ? 1 : 0;
  const size_t r0 = cidx * (size_t)chunk_len;
  const size_t r1 = r0 + chunk_len < rows ? r0 + chunk_len : rows;
  const unsigned G = geo.G, NG = geo.NG, ppr = geo.ppr, stride = geo.stride_bytes;
  const unsigned ngroups = (unsigned)((r1 - r0 + G - 1) / G);
  if (threadIdx.x < 128) filled[threadIdx.x] = 0u;
  __syncthreads();

  if (wave == 0)
  {
    // ---- the adding wave: lane j owns row slot j ----
    // One pass of the loop adds a block of 128 bins in stages of 128 bytes of terms (16 double / 32 float); while a stage is added the next one
    // is on its way from LDS into the other of two register buffers (the last stage fetches the first one of the row's next block).
    __builtin_amdgcn_s_setprio(3);
    constexpr int SV = 8, PER = 16 / (int)sizeof(FD);      // 16-byte vectors per stage (the statement below names eight), terms per vector
    constexpr int BLK = GEO::kBlockBins, NS = BLK / (SV * PER);
    static_assert(NS % 2 == 0, "the stages alternate between two buffers");
    const bool mine = (unsigned)lane < NG * G;
    const unsigned q = (unsigned)lane / G, rin = (unsigned)lane - q * G;
    const unsigned blocks = ppr * (unsigned)PIECE / (unsigned)BLK;
    const unsigned char* my = slots + (size_t)lane * stride;
    unsigned it = 0, pos = 0;                              // occupancy number of my group of slots (its rows: group it * NG + q of the chunk), block within the row
    bool active = false;
    FD sum = (FD)0;
    unsigned remaining = ngroups;
    TV b0[SV], b1[SV];
    while (remaining)
    {
      if (mine && !active && it * NG + q < ngroups)
      {
        if (__hip_atomic_load(&filled[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)LOADERS * (it + 1u))
        {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          active = true; pos = 0; sum = (FD)0;
          const TV* t = reinterpret_cast<const TV*>(my);
#pragma unroll
          for (int c = 0; c < SV; ++c) b0[c] = t[c];
        }
      }
      if (__ballot(active) == 0ull) { __builtin_amdgcn_s_sleep(2); continue; }
      bool finished = false;
      if (active)
      {
        const TV* t = reinterpret_cast<const TV*>(my + (size_t)pos * (BLK * sizeof(FD)));
#pragma unroll
        for (int st = 0; st < NS; ++st)
        {
          // (past the row's last block: the next slot's first terms, or the slack after the last slot -- fetched, never added)
#pragma unroll
          for (int c = 0; c < SV; ++c) { if ((st & 1) == 0) b1[c] = t[(st + 1) * SV + c]; else b0[c] = t[(st + 1) * SV + c]; }
          __builtin_amdgcn_sched_barrier(0);
          // ONE wait for the whole stage: the empty statement reads all of the stage's registers, so the compiler waits here once (LDS answers
          // in order: when at most the SV reads just issued are open, the stage before them has arrived) and not again before every second
          // addition -- s_waitcnt takes an issue slot like any instruction, and a lone wave issues one instruction in 4.5 cycles
          if ((st & 1) == 0) asm volatile("" : : "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b0[4]), "v"(b0[5]), "v"(b0[6]), "v"(b0[7]) : "memory");
          else asm volatile("" : : "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]), "v"(b1[4]), "v"(b1[5]), "v"(b1[6]), "v"(b1[7]) : "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < SV; ++c)
#pragma unroll
            for (int e = 0; e < PER; ++e) sum += ((st & 1) == 0 ? b0[c][e] : b1[c][e]);   // sdft.h:641-651: one accumulator, ascending bins
          __builtin_amdgcn_sched_barrier(0);
        }
        ++pos;
        finished = pos == blocks;
      }
      if (finished)
      {
        const size_t r = r0 + (size_t)(it * NG + q) * G + rin;
        if (r < r1) a.y[r] = (TD)(sum * a.sweight);      // sdft.h:654-656 (the samples of the channels follow each other as the rows do)
        active = false; ++it;
        if (rin == 0) __hip_atomic_fetch_add(&freed[q], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      remaining -= (unsigned)__popcll(__ballot(finished && rin == 0));
    }
    return;
  }

  // ---- the loaders: the pieces of a group (G rows x ppr pieces of 1 KiB) dealt round over the loader waves ----
  const unsigned lw = wave - 1u;
  const unsigned np = G * ppr;
  unsigned prow[MAXP], kbin[MAXP]; bool live[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p)
  {
    const unsigned i = lw + (unsigned)p * LOADERS;
    live[p] = i < np;
    prow[p] = i / ppr;
    kbin[p] = (i - prow[p] * ppr) * PIECE + (unsigned)lane * BPL;
  }
  // (16-byte loads only, and the matrix is one run of rows: the launch sees to it that FD float rows have an even number of bins and a 16-byte
  // aligned start, and that channels follow each other without a gap -- a load costs a handful of instructions, not a 64-bit division)
  bool livek[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) livek[p] = live[p] && kbin[p] < a.nbins;
  // a piece of group g this wave does not have (past the group, the chunk or the row) is loaded all the same, from the chunk's first bytes (not from one
  // address for the whole chip: 3 % of the loads on one line of one channel cost 5 % of the speed), and
  // parked as +0: EVERY fetch issues MAXP loads, so the wait before a group is parked is for that group's loads alone (behind conditional
  // loads the compiler waits for all that are on their way -- the ring would drain every third group)
  auto have = [&](unsigned g, int p) -> bool { return livek[p] && g < ngroups && r0 + (size_t)g * G + prow[p] < r1; };
  auto loaders = [&](auto nt_tag)
  {
    constexpr int NT = decltype(nt_tag)::value;
    auto fetch = [&](unsigned g, V (&x)[MAXP])
    {
#pragma unroll
      for (int p = 0; p < MAXP; ++p)
      {
        const cx<FD>* src = a.in + (have(g, p) ? (r0 + (size_t)g * G + prow[p]) * (size_t)a.nbins + kbin[p] : r0 * (size_t)a.nbins);
        // (not load_vec(src, NT): the select of two loads with a constant condition folds to an ordinary load)
        if constexpr (NT != 0) x[p] = __builtin_nontemporal_load(reinterpret_cast<const V*>(src)); else x[p] = *reinterpret_cast<const V*>(src);
      }
    };
    auto put = [&](unsigned g, const V (&x)[MAXP])
    {
      const unsigned it = g / NG, q = g - it * NG;
      // the slots' rows before these have been added and stored
      // (the wait is one opaque statement: around a polling LOOP the compiler waits for every load that is on its way at the top of the
      // ring's loop, and the ring drains every third group)
      {
        unsigned seen;
        const unsigned at = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)(&freed[q]);
        asm volatile("1:\n\t"
                     "ds_read_b32 %0, %1\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_cmp_ge_u32 vcc, %0, %2\n\t"
                     "s_cbranch_vccnz 2f\n\t"
                     "s_sleep 1\n\t"
                     "s_branch 1b\n\t"
                     "2:"
                     : "=&v"(seen) : "v"(at), "v"(it) : "vcc", "memory");
      }
#pragma unroll
      for (int p = 0; p < MAXP; ++p)
      {
        // (every load is waited for here, in the order of issue, whether the wave has the piece or not: a load nobody consumes stays "on its way"
        // in the compiler's count and costs a wait for everything at the top of the ring's loop; latency 1 needs the real halves only, and
        // without this the other halves' registers are handed out while the load is on its way)
        if constexpr (sizeof(FD) == 8) asm volatile("" : : "v"(x[p][0]), "v"(x[p][1]));
        else asm volatile("" : : "v"(x[p][0]), "v"(x[p][1]), "v"(x[p][2]), "v"(x[p][3]));
        if (live[p])
        {
          const bool h = have(g, p);
          FD tv[BPL];
#pragma unroll
          for (int b = 0; b < BPL; ++b)
          {
            const unsigned k = kbin[p] + b;                // (bins past N-1 park +0)
            const cx<FD> v = cmake<FD>((FD)x[p][2 * b], (FD)x[p][2 * b + 1]);
            tv[b] = (h && k < a.nbins) ? synth_term<FD, LAT1, false>(v, k, a.op, a.syn, a.nbins) : (FD)0;
          }
          FD* dst = reinterpret_cast<FD*>(slots + (size_t)(q * G + prow[p]) * stride) + kbin[p];
#pragma unroll
          for (int b = 0; b < BPL; ++b) dst[b] = tv[b];
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(&filled[q], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    V xa[MAXP], xb[MAXP], xc[MAXP];                        // a ring of three groups: two on their way while one is parked
    // (every pass begins by parking what was fetched first: the same eighteen loads are on their way, in the same order, whether the loop is
    // entered or repeated -- with a fetch at the top the compiler waits for ALL of them there)
    fetch(0, xa);
    fetch(1, xb);
    fetch(2, xc);
    for (unsigned g = 0; g < ngroups; g += 3)
    {
      put(g, xa);
      fetch(g + 3, xa);
      if (g + 1 >= ngroups) break;
      put(g + 1, xb);
      fetch(g + 4, xb);
      if (g + 2 >= ngroups) break;
      put(g + 2, xc);
      fetch(g + 5, xc);
    }
  };
  if (nt) loaders(OpTag<1>{}); else loaders(OpTag<0>{});
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
    const int nt = (a.nt && gi * (size_t)RW >= a.nt_skip) ? 1 : 0;     // (wave-uniform)
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
            const V q = load_vec(reinterpret_cast<const V*>(rowp + k), nt);
            v[i][0] = cmake<FD>((FD)q[0], (FD)q[1]);
            if constexpr (BPL == 2) v[i][1] = cmake<FD>((FD)q[2], (FD)q[3]);
          }
          else
          {
#pragma unroll
            for (int b = 0; b < BPL; ++b)
              if (k + b < a.nbins) v[i][b] = load_bin(rowp + k + b, nt);
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
// (the body is a device function of (arguments, the wave's row, its block of LDS): inverse_row_kernel runs it once per launch, the resident kernel of
// sdft_resident.hpp once per row a wave takes)
template <typename FD> struct inverse_row_geometry
{
  static constexpr int BPL = 16 / (int)sizeof(cx<FD>);   // bins per 16-byte load
  static constexpr int NL = 16;                          // loads in flight per lane
  static constexpr int TB = kWave * NL * BPL;            // bins per LDS block (1024 f64 / 2048 f32: 8 KiB)
};
template <typename TD, typename FD, bool LAT1, bool OPS = false>
SDFT_D void inverse_row_body(const InverseArgs<TD, FD>& a, const size_t r, FD* terms)
{
  constexpr int BPL = inverse_row_geometry<FD>::BPL, NL = inverse_row_geometry<FD>::NL, TB = inverse_row_geometry<FD>::TB;
  using V = typename StoreVec<FD, (sizeof(cx<FD>) == 16 ? 1 : 2)>::type;

  const int lane = threadIdx.x & (kWave - 1);
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
        if (same_bits(ylo, yhi)) { decided = true; decided_y = ylo; break; }
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
template <typename TD, typename FD, bool LAT1, bool OPS = false>
__global__ __launch_bounds__(kWave) void inverse_row_kernel(InverseArgs<TD, FD> a)
{
  __shared__ __align__(16) FD terms[inverse_row_geometry<FD>::TB];
  // last rows first (what the analysis wrote last is still in cache)
  inverse_row_body<TD, FD, LAT1, OPS>(a, (size_t)gridDim.x - 1 - blockIdx.x, terms);
}

}  // namespace sdfthip
