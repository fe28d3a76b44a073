// sdft_resident.hpp -- an opt-in RESIDENT kernel for the reference driver's hop loop (/root/reference/test/test.c:69-83:
// sdft_sdft_n + sdft_isdft_n per hop of 100 samples, synchronous calls on device pointers).
// Not part of the run-time-compiled text (sdft_kernels.hpp): plain analysis and synthesis only.
//
// A synchronous call costs a launch and a completion whatever its kernel does: 13.5 us per call where the two kernels of a hop need 11 us
// together (profiles/r05_hop_time_parts.txt).  With option "resident" = 1 the first such call starts ONE kernel of kResidentBlocks small
// workgroups that stays on the chip and serves the calls that follow: the host writes a call into one cache line of pinned memory (pointers,
// length, cursor, state slots; the doorbell -- a sequence number -- last), workgroup 0 polls that line over PCIe (one 64-byte read per poll),
// copies it to device memory with one 64-byte store (two slots, by the call's number); the other workgroups poll their slot there.  The work of a call is the body of
// forward_hop2_kernel (analysis: every (tile of bins, time part) one workgroup) or of inverse_row_kernel (synthesis: a wave per row) -- the
// same device functions, bit for bit the same results -- and completion is the kernels' own completion word in pinned memory.
// What orders a call after the one before it: every workgroup ends a call with an agent-scope release (the ticket of the completion word)
// and begins the next with an agent-scope acquire (the poll of the device word), and the host rings the doorbell only after it has seen the
// completion word -- a synthesis reads the rows the analysis before it wrote, on whichever XCD.
// The kernel LEAVES by itself when no call has come for `idle_ticks` of the 100 MHz clock (200 us): a blocking-stream hipMemcpy of the
// host, which waits for the plan's stream, waits that long at most.  Every other entry point of the plan retires it first (a QUIT call).
// On leaving it writes the number of the last call it served to pinned memory: a host whose call raced with the time-out sees that its
// call was not served and launches the ordinary kernel.  All polls are bounded.

#pragma once

#include "sdft_forward_hop.hpp"
#include "sdft_inverse.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

constexpr unsigned kResidentBlocks = 256;                  // workgroups of the resident kernel (two waves each): one per CU
enum { RES_ANALYSIS = 1, RES_SYNTHESIS = 2, RES_QUIT = 3, RES_ANALYSIS_SAMPLE = 4 };     // (4: sdft_sdft -- ONE sample, carried by the call itself)

// one call = one cache line of pinned host memory; the host writes `seq` LAST (the line is read with one 64-byte request: a snapshot)
struct __attribute__((aligned(64))) ResidentCall
{
  unsigned long long x, out, y;                            // device pointers: analysis x -> out, synthesis out -> y (RES_ANALYSIS_SAMPLE: y holds the sample's bits)
  unsigned n, cursor0, parts, part_len;
  unsigned slots;                                          // bits 0-1: the state slot the call reads, bits 2-3: the delay-line slot (it writes slot ^ 1)
  unsigned op, flag_seq, blocks;                           // blocks: workgroups (analysis) / rows (synthesis) that report to the completion word
  unsigned check;                                          // ~(seq ^ xor of words 0..13): a line whose words do not belong together is polled again
  unsigned seq;                                            // the doorbell
};
static_assert(sizeof(ResidentCall) == 64, "one cache line");

template <typename TD, typename FD> struct ResidentArgs
{
  HopArgs<TD, FD> ha;                                      // what does not change from call to call (tables, geometry, window scale)
  InverseArgs<TD, FD> ia;
  cx<FD>* acc[4];                                          // the plan's state slots
  cx<FD>* fid[4];
  TD* hist[4];
  const unsigned* host_call;                               // the line the host writes (pinned, mapped), as 16 words
  unsigned* dev_call;                                      // two device copies of it (by call number & 1), published by workgroup 0
  unsigned* done_flag;                                     // completion word (pinned) and its ticket counter (device): DoneSignal
  unsigned* done_count;
  unsigned* exit_word;                                     // pinned: the last call served, written when the kernel leaves
  unsigned first_seq;                                      // the first call this launch serves
  unsigned idle_ticks;                                     // leave when no call has come for that long (100 MHz clock)
};

template <typename TD, typename FD, int BPL, int WIN, bool LAT1>
__global__ __launch_bounds__(2 * kWave) void resident_hop_kernel(ResidentArgs<TD, FD> ra)
{
  __shared__ __align__(16) unsigned cur[16];               // the call being served (ResidentCall as words)
  __shared__ __align__(16) FD terms[2][inverse_row_geometry<FD>::TB];
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned next = ra.first_seq;                            // the call this workgroup waits for
  unsigned served = ra.first_seq - 1u;                     // (workgroup 0: the last call taken from the host)
  for (;;)
  {
    if (wave == 0)
    {
      // xor of the line's words 0..13: part of the check word, so that a line whose words do not belong together never passes for a call
      auto line_ok = [&](unsigned word, unsigned want_seq, bool exact) -> bool
      {
        unsigned f = lane < 14 ? word : 0u;
        f ^= (unsigned)__shfl_xor((int)f, 8, 16); f ^= (unsigned)__shfl_xor((int)f, 4, 16);
        f ^= (unsigned)__shfl_xor((int)f, 2, 16); f ^= (unsigned)__shfl_xor((int)f, 1, 16);
        const unsigned fold = (unsigned)__builtin_amdgcn_readlane((int)f, 0);
        const unsigned seq = (unsigned)__builtin_amdgcn_readlane((int)word, 15), chk = (unsigned)__builtin_amdgcn_readlane((int)word, 14);
        if (chk != ~(seq ^ fold)) return false;
        // exact: this very call; else this call or a later one of the same slot (a workgroup that fell behind had no work in the ones it skips:
        // the host rings only after every workgroup with work has reported)
        return exact ? seq == want_seq : ((int)(seq - want_seq) >= 0 && ((seq - want_seq) & 1u) == 0u);
      };
      if (blockIdx.x == 0)
      {
        // ---- workgroup 0: the host's line, over PCIe ----
        const unsigned long long t0 = wall_clock64();
        unsigned word = 0;
        bool quit = false;
        for (;;)
        {
          word = lane < 16 ? __hip_atomic_load(const_cast<unsigned*>(ra.host_call) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
          if (line_ok(word, next, true)) break;
          if (wall_clock64() - t0 > (unsigned long long)ra.idle_ticks) { quit = true; break; }
        }
        if (quit) word = lane == 11 ? (unsigned)RES_QUIT : (lane == 15 ? next : (lane == 14 ? ~(next ^ (unsigned)RES_QUIT) : 0u));
        else served = next;
        // one 64-byte store publishes the call to the other workgroups (the line carries its own number and check word: nothing else to order)
        if (lane < 16) { __hip_atomic_store(ra.dev_call + 16u * (next & 1u) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cur[lane] = word; }
      }
      else
      {
        // ---- the others: the call's slot in device memory, one 64-byte read per poll ----
        // (relaxed polls: an acquire per poll would invalidate the XCD's L2 under the workgroups that are working; the one acquire a call
        // needs is made below, by the workgroups that have work in it)
        const unsigned long long t0 = wall_clock64();
        unsigned word = 0;
        for (;;)
        {
          word = lane < 16 ? __hip_atomic_load(ra.dev_call + 16u * (next & 1u) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
          if (line_ok(word, next, false)) break;
          // (four idle periods: workgroup 0 has long said QUIT if it lives)
          if (wall_clock64() - t0 > 4ull * (unsigned long long)ra.idle_ticks) { word = lane == 11 ? (unsigned)RES_QUIT : 0u; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane < 16) cur[lane] = word;
      }
    }
    __syncthreads();
    const unsigned op = cur[11];
    if (op != (unsigned)RES_ANALYSIS && op != (unsigned)RES_SYNTHESIS && op != (unsigned)RES_ANALYSIS_SAMPLE) break;      // QUIT (or a line nobody should have written)
    const unsigned long long px = ((unsigned long long)cur[1] << 32) | cur[0], pout = ((unsigned long long)cur[3] << 32) | cur[2],
                             py = ((unsigned long long)cur[5] << 32) | cur[4];
    const unsigned n = cur[6], blocks = cur[13];
    DoneSignal done; done.flag = ra.done_flag; done.count = ra.done_count; done.seq = cur[12]; done.total = blocks;
    // a workgroup with work in this call sees what the call before it wrote, on any XCD (its own release was the completion ticket)
    const bool analysis = op != (unsigned)RES_SYNTHESIS;
    const bool has_work = analysis ? blockIdx.x < blocks : blockIdx.x < n;
    if (has_work) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (analysis)
    {
      if (blockIdx.x < blocks)
      {
        HopArgs<TD, FD> a = ra.ha;
        const unsigned ss = cur[10] & 3u, hs = (cur[10] >> 2) & 3u;
        // (a single sample travels in the call's line: the workgroup's own copy of it in LDS is the sample "array")
        a.x = op == (unsigned)RES_ANALYSIS_SAMPLE ? reinterpret_cast<const TD*>(&cur[4]) : reinterpret_cast<const TD*>(px); a.x_stride = n;
        a.out = reinterpret_cast<cx<FD>*>(pout); a.out_stride = (size_t)n * a.nbins; a.out_rows = nullptr;
        a.n = n; a.cursor0 = cur[7]; a.parts = cur[8]; a.part_len = cur[9];
        a.acc_in = ra.acc[ss]; a.fid_in = ra.fid[ss]; a.acc_out = ra.acc[ss ^ 1u]; a.fid_out = ra.fid[ss ^ 1u];
        a.hist_in = ra.hist[hs]; a.hist_out = ra.hist[hs ^ 1u];
        a.vec_store = (BPL == 2 && (a.nbins % 2u) == 0u && (pout % 16ull) == 0ull) ? 1 : 0;
        a.done = done;
        forward_hop2_body<TD, FD, BPL, WIN, false>(a, blockIdx.x, blocks);
      }
    }
    else
    {
      InverseArgs<TD, FD> ia = ra.ia;
      ia.in = reinterpret_cast<const cx<FD>*>(pout); ia.in_stride = (size_t)n * ia.nbins; ia.in_rows = nullptr;
      ia.y = reinterpret_cast<TD*>(py); ia.y_stride = n; ia.n = n;
      ia.done = done;
      // a wave per row, the rows spread over the workgroups (a hop of 100 rows: one wave on each of 100 CUs), last rows first
      for (size_t r = (size_t)blockIdx.x + (size_t)wave * gridDim.x; r < (size_t)n; r += 2u * (size_t)gridDim.x)
        inverse_row_body<TD, FD, LAT1, false>(ia, (size_t)n - 1u - r, terms[wave]);
    }
    __syncthreads();                                         // everybody is through with `cur`
    next = cur[15] + 1u;
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(ra.exit_word, served, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace sdfthip
