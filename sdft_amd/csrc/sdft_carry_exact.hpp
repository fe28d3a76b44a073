// sdft_carry_exact.hpp -- K1a': exact carries, the reference's rounding sequence (serial pass; relay form with its seed table)
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_carry_fast.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

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

// ------------------------------------------------------------------------------------------
// words of the inter-wave protocols in LDS, accessed as workgroup-scope atomics on the __shared__ objects themselves
// (a volatile access through a generic pointer compiles to flat_load/flat_store sc0 sc1 and drags a full
// s_waitcnt behind it)
// ------------------------------------------------------------------------------------------
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

// Holds the forward launch back until the relay workgroups of the call are resident (ChainArgs::started counts them): one
// wave on the forward stream, a poll of one device word per microsecond, bounded (about 20 ms: the forward kernel's own
// waits are bounded too, and a call whose relays never came is re-run with the serial pass).
template <typename FD> __global__ __launch_bounds__(kWave) void relay_gate_kernel(const unsigned* started, unsigned target)   // (a template: one definition per translation unit)
{
  if (threadIdx.x != 0) return;
  for (unsigned polls = 0; polls < (1u << 15); ++polls)
  {
    const unsigned now = __hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(now - target) >= 0) return;
    __builtin_amdgcn_s_sleep(16);
  }
}

// STATS: measurement build (ChainArgs::stats; instantiated for the longest block only)
// A workgroup may hold TWO relays (ChainArgs::P waves each, 32 bins each): the pass then occupies half as many CUs, and
// the forward launches of earlier segments -- whose 16-wave workgroups cannot share a CU with it -- keep three quarters
// of the chip instead of half (config 3: 128 relays).  FD float only: 12 waves of <= 168 registers fit a CU, FD double's
// 185 registers allow 8.
// Round 6: blocks of up to 64 steps keep the products in 64 registers, so SIXTEEN waves fit a CU (4 per SIMD: two relays of eight waves, test
// hook "relay_groups" = 2) -- the form the round-5 review asked for; measured in profiles/r06_relay_on_64_cus.txt.
template <typename FD, int L = 128> struct relay_limits
{
  static constexpr int waves = sizeof(FD) == 4 ? (L <= 64 ? 16 : 12) : 8;
  static constexpr int groups = sizeof(FD) == 4 ? 2 : 1;
};
template <typename FD, int L, bool STATS = false>
__global__ __launch_bounds__((kWave * relay_limits<FD, L>::waves)) void carry_relay_kernel(ChainArgs<FD> a)
{
  constexpr int DV = (L + 15) / 16;                        // difference vectors per block (16 steps each)
  using token = RelayToken<FD>;
  using raw_t = typename token::raw_t;
  __shared__ __align__(16) raw_t mails[relay_limits<FD, L>::groups][kWave];
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

}  // namespace sdfthip
