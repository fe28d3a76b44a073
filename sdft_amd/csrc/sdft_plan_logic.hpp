// sdft_plan_logic.hpp -- the host-side DECISIONS of the engine, free of any HIP dependency: launch geometry (lanes, tiles,
// row slots), time chunking, the block length of the exact-carry relay, which calls leave the plan's stream (call pattern,
// row-stream ring, address-range overlap), time parts of a hop, rows per wave of the synthesis, the wait of a synchronous
// call, and the slot ring of the host-copy engine.  sdft_plan.hpp (Plan<TD, FD>) asks these functions and does the HIP calls;
// tests/cpp/plan_logic_test.cpp compiles this header alone with g++ -fsanitize=address,undefined (and the ring under
// -fsanitize=thread) in the `-m "not gpu"` suite (sanitizers on the host side).
// Citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include <stddef.h>
#include <stdint.h>

#include <algorithm>

namespace sdfthip {
namespace logic {

// the kernels' constants this logic is written against (sdft_plan.hpp asserts that they are the kernels')
constexpr int kLanes = 64;              // lanes of a wave (kWave)
constexpr int kRowWaves = 16;           // waves of a row-group workgroup (kRowWavesMax)
constexpr int kRowSlots = 2;            // bins-per-lane slots of the row-group kernel (kRowSlotsMax)
constexpr int kTimeGroup = 8;           // samples per scalar-load burst of the time loops (kGroup)
constexpr int kSumBlockLen = 8;         // samples per block of the direct partial sums (kSumBlock)
constexpr int kHopSamples = 512;        // calls of one time chunk are shorter than this (kHopMax)
constexpr int kWindowHann = 1, kWindowBlackman = 3, kWindowBoxcar = 0;   // sdft.h:127-133

// ---- lanes, tiles, row groups ---------------------------------------------------------------------------------------------
inline int bins_per_lane(size_t fdx_bytes) { return fdx_bytes == 16 ? 1 : 2; }          // every lane stores 16 bytes
inline int halo_bins(int window) { return window == kWindowBlackman ? 2 : (window == kWindowBoxcar ? 0 : 1); }   // :350-402
inline int halo_lanes(int window, size_t fdx_bytes) { const int b = bins_per_lane(fdx_bytes); return (halo_bins(window) + b - 1) / b; }
// Bin-owning lanes of an independent-tile wave: a tile of 8*j lanes starts and ends on 128-byte lines, so no line is shared
// between two waves (n = 1e6, N = 1024, f64: 62 lanes 4.2 TB/s, 60 lanes 5.4, 56 lanes 5.6)
inline long interior_lanes(int window, size_t fdx_bytes, long forced)
{
  const long mx = kLanes - 2 * halo_lanes(window, fdx_bytes);
  const long v = forced > 0 ? std::min(forced, mx) : (mx / 8) * 8;
  return std::max(v, 1L);
}
inline long tiles(size_t nbins, int window, size_t fdx_bytes, long forced_interior)
{
  const long per = interior_lanes(window, fdx_bytes, forced_interior) * bins_per_lane(fdx_bytes);
  return (long)((nbins + (size_t)per - 1) / (size_t)per);
}
inline bool rows_kernel_ok(size_t nbins, size_t fdx_bytes, bool row_pointers, bool enabled, long slots_max)
{
  const size_t per = (size_t)(kLanes * kRowWaves * bins_per_lane(fdx_bytes));
  return enabled && !row_pointers && nbins >= 8 && nbins <= per * (size_t)std::min<long>(kRowSlots, std::max<long>(1, slots_max));
}
inline long row_slots(size_t nbins, size_t fdx_bytes)
{
  const size_t per = (size_t)(kLanes * kRowWaves * bins_per_lane(fdx_bytes));
  return (long)((nbins + per - 1) / per) <= 1 ? 1 : 2;
}
inline long row_waves(size_t nbins, size_t fdx_bytes)
{
  const size_t per = (size_t)(kLanes * bins_per_lane(fdx_bytes)) * (size_t)row_slots(nbins, fdx_bytes);
  return (long)((nbins + per - 1) / per);
}

// ---- time chunking ------------------------------------------------------------------------------------------------------------
struct ChunkQuery
{
  size_t n = 0, channels = 1, nbins = 0;
  bool rows_kernel = false;      // the row-group kernel takes the call: one workgroup per (channel, chunk)
  bool exact = false;            // exact carries (the reference's rounding sequence)
  bool pipelined = false;        // the call's rows run beside the neighbouring calls' (two row streams)
  long forced_chunk = 0;         // option "chunk"
  long target_waves = 0;         // option "target_waves"
  long row_waves = 1, tiles = 1; // of the plan's geometry
  int compute_units = 256;
};
struct Chunking { long chunks = 1, len = 0; };

// Enough workgroups to fill the chip, chunks not shorter than a minimum.  Row-group kernel: two rounds of the CUs (the carry
// pre-pass gets cheaper with fewer chunks: 1017 chunks 0.044 ms, 511 chunks 0.029 ms at n = 1e6); exact carries: 2048 chunks
// (8 overlap segments of >= 256 workgroups).  Calls below 36 000 samples are bound by the serial samples of one chunk
// (~0.36 us each), not by HBM: about 190 chunks of >= 32 samples.  WHOLE ROUNDS of the chip: a row group is one workgroup per
// CU, so 260 ... 500 workgroups on 256 CUs are one full round plus a partly filled one that takes just as long (n = 52 000:
// 260 chunks of 200 rows 226 us, 250 chunks of 208 rows 168 us): between one and two rounds the call takes ONE round of
// longer chunks.  Pipelined calls are the opposite case (the next call's workgroups fill whatever a launch leaves free, and a
// launch that fills the chip exactly keeps all workgroups in step): about 300 chunks of >= 160 rows whatever the length.
inline Chunking choose_chunks(const ChunkQuery& q)
{
  Chunking r;
  const long ch = (long)std::max<size_t>(q.channels, 1);
  const size_t n = q.n;
  if (q.rows_kernel && q.forced_chunk <= 0 && n >= (size_t)kHopSamples)
  {
    const long target_blocks = q.target_waves > 0 ? std::max(1L, q.target_waves / std::max(1L, q.row_waves)) : (q.exact ? 2048 : 512);
    long want = std::max(1L, (target_blocks + ch - 1) / ch);
    const bool mid = q.target_waves <= 0 && q.channels * n < 36000;
    if (mid) want = std::max(1L, std::min((190L + ch - 1) / ch, (long)(n / 32)));
    else want = std::max(1L, std::min(want, (long)(n / (q.exact ? 128 : 192))));
    if (!mid && q.pipelined && q.target_waves <= 0)
    {
      const long total = std::max(1L, (300L + ch - 1) / ch);
      want = std::max(1L, std::min(total, (long)(n / 160)));
    }
    else if (!mid && !q.exact && q.target_waves <= 0)
    {
      const long round = (long)q.compute_units;
      const long blocks = want * ch;
      if (blocks > round && blocks < 2L * round) want = std::max(1L, round / ch);
    }
    long len = (long)((n + (size_t)want - 1) / (size_t)want);
    len = ((len + kTimeGroup - 1) / kTimeGroup) * kTimeGroup;
    if (q.exact)
    {
      len = ((len + 31) / 32) * 32;                          // whole trips of the exact pass's inner loop
      if (mid && len > 32) len = ((len + 63) / 64) * 64;
      if (!mid && len > 64) len = ((len + 127) / 128) * 128; // whole blocks of the relay form
    }
    len = std::max(1L, std::min(len, (long)n));
    r.len = len; r.chunks = (long)((n + (size_t)len - 1) / (size_t)len);
    return r;
  }
  const long target = q.target_waves > 0 ? q.target_waves : 16384;
  const long min_len = 64;
  if (q.forced_chunk <= 0 && n < (size_t)kHopSamples) { r.chunks = 1; r.len = (long)n; return r; }   // short hops stay serial (and bit-exact)
  const long per = std::max(1L, ch * std::max(1L, q.tiles));
  long want = (target + per - 1) / per;
  long len;
  if (q.forced_chunk > 0) len = !q.exact ? ((q.forced_chunk + kSumBlockLen - 1) / kSumBlockLen) * kSumBlockLen : q.forced_chunk;
  else
  {
    want = std::max(1L, std::min(want, (long)(n / (size_t)min_len)));
    len = (long)((n + (size_t)want - 1) / (size_t)want);
    len = ((len + kTimeGroup - 1) / kTimeGroup) * kTimeGroup;
    if (q.exact) len = ((len + 31) / 32) * 32;
  }
  len = std::max(1L, std::min(len, (long)std::max<size_t>(n, 1)));
  r.len = len; r.chunks = (long)((n + (size_t)len - 1) / (size_t)len);
  return r;
}

// ---- exact carries, relay form: block length = seed distance -----------------------------------------------------------
// divides 2N and the chunk length; L products live in L registers per lane (128 at FD float, 64 register pairs at FD double);
// the seed table (fid at every L-th cursor) stays below 256 MiB.  0: no block length fits (serial pass).
inline unsigned relay_block(size_t nbins, long chunk_len, size_t fd_bytes, size_t fdx_bytes, long forced)
{
  const size_t span = 2 * nbins;
  const unsigned top = fd_bytes == 4 ? 128u : 64u;
  for (unsigned cand : {128u, 64u, 32u, 16u, 8u})
  {
    if (cand > top) continue;
    if (forced > 0 && (unsigned)forced != cand) continue;
    if (span % cand == 0 && chunk_len > 0 && (size_t)chunk_len % cand == 0 && ((span / cand) * nbins * fdx_bytes) <= ((size_t)256 << 20)) return cand;
  }
  return 0;
}

// ---- radices of the mixed-radix FFT of `span` points (4, 2, 3, 5); count == 0: span has other prime factors --------------
struct Radices { unsigned char count = 0; unsigned char r[15] = {}; };
inline Radices smooth_radices(size_t span)
{
  Radices rl;
  size_t rem = span;
  if (rem == 0) return rl;
  for (unsigned f : {4u, 2u, 3u, 5u})
    while (rem % f == 0 && rl.count < 15) { rl.r[rl.count++] = (unsigned char)f; rem /= f; }
  if (rem != 1) rl.count = 0;
  return rl;
}
// LDS cells the in-kernel DFT of a self-carried chunk works in: 2N in place for powers of two, two buffers of 2N for the
// 2/3/5-smooth sizes (Stockham); 0: this 2N has neither form
inline size_t self_cells(size_t nbins, bool enabled, size_t fdx_bytes)
{
  const size_t span = 2 * nbins;
  if (span < 16 || span > 4096) return 0;
  if ((span & (span - 1)) == 0) return span;
  return (enabled && smooth_radices(span).count > 0 && 2 * span * fdx_bytes <= (size_t)80 * 1024) ? 2 * span : 0;
}

// ---- address ranges -------------------------------------------------------------------------------------------------------------
struct Range { uintptr_t lo = 0, hi = 0; };
inline bool overlap(uintptr_t alo, uintptr_t ahi, const Range& b) { return alo < b.hi && b.lo < ahi; }
inline bool overlap(const Range& a, const Range& b) { return a.lo < b.hi && b.lo < a.hi; }

// ---- which kind of host is calling: learnt from the calls ------------------------------------------------------------------
// Pipelining pays for a host that analyses call after call (or synthesises call after call); a host that alternates the two
// (the reference's loop, test/test.c:69-83) would pay an event wait between streams per call and gain nothing.  So a mode is
// on once two of a kind have come in a row and off again when the other kind follows a lone one.
struct CallPattern
{
  bool prev_was_inverse = false, inverse_batch = false;
  int inverse_run = 0;
  bool prev_was_analysis = false, analysis_batch = false;
  int analysis_run = 0;
  // an analysis call is about to be launched (fused = the fused analysis -> synthesis call, which is neither kind)
  void on_analysis(bool fused)
  {
    if (prev_was_inverse && inverse_run == 1) inverse_batch = false;
    prev_was_inverse = false; inverse_run = 0;
    if (prev_was_analysis && !fused) analysis_batch = true;
    prev_was_analysis = !fused;
    if (!fused) ++analysis_run;
  }
  // a synthesis call is about to be launched
  void on_synthesis_begin()
  {
    if (prev_was_inverse) inverse_batch = true;
    if (prev_was_analysis && analysis_run == 1) analysis_batch = false;
    prev_was_analysis = false; analysis_run = 0;
  }
  void on_synthesis_launched() { prev_was_inverse = true; ++inverse_run; }
};

// ---- pipelined analyses: the two row streams and the ring of four outstanding launches ------------------------------------
// Which row stream a launch goes to: the other one than the previous launch's -- unless the call's matrix overlaps what an
// outstanding launch writes: then the stream of the latest such launch (whose order costs nothing), and if launches on the
// other stream overlap as well, the latest of those is waited for by an event.
struct RowRing
{
  unsigned long long seq = 0;              // launches since the ring was last joined
  Range out[4];                            // what the outstanding launches write (by launch number & 3)
  int stream_of[4] = {0, 0, 0, 0};
  bool open = false;
  struct Pick { int stream = 0; bool behind = false; int wait_launch = -1; /* ring slot of a launch on the other stream to wait for */ };
  // do the call's SAMPLES lie in a matrix an outstanding launch is still writing?  (then everything joins first)
  bool samples_overlap(uintptr_t xlo, uintptr_t xhi) const
  {
    for (unsigned long long back = 1; back <= 3 && back <= seq; ++back)
      if (overlap(xlo, xhi, out[(seq - back) & 3])) return true;
    return false;
  }
  Pick pick(uintptr_t olo, uintptr_t ohi) const
  {
    Pick p;
    p.stream = seq ? (stream_of[(seq - 1) & 3] ^ 1) : 0;
    for (unsigned long long back = 1; back <= 3 && back <= seq; ++back)
      if (overlap(olo, ohi, out[(seq - back) & 3])) { p.stream = stream_of[(seq - back) & 3]; p.behind = true; break; }
    if (p.behind)
      for (unsigned long long back = 1; back <= 3 && back <= seq; ++back)
      {
        const int q = (int)((seq - back) & 3);
        if (stream_of[q] != p.stream && overlap(olo, ohi, out[q])) { p.wait_launch = q; break; }
      }
    return p;
  }
  // the state slot a launch's state kernel writes was read by the rows of three launches ago: their ring slot, or -1
  int state_reader() const { return seq >= 3 ? (int)((seq + 1) & 3) : -1; }
  int slot() const { return (int)(seq & 3); }
  void launched(uintptr_t olo, uintptr_t ohi, int stream) { out[seq & 3] = Range{olo, ohi}; stream_of[seq & 3] = stream; ++seq; open = true; }
  // the ring slots of the last launch on each row stream (older ones are ordered before them); returns how many
  int last_per_stream(int slots[2]) const
  {
    int count = 0;
    bool seen[2] = {false, false};
    for (unsigned long long back = 1; back <= 4 && back <= seq; ++back)
    {
      const int q = (int)((seq - back) & 3), rsi = stream_of[q];
      if (!seen[rsi]) { seen[rsi] = true; slots[count++] = q; }
    }
    return count;
  }
  void joined() { open = false; seq = 0; }
};

// ---- pipelined syntheses (stateless): two streams in turn ------------------------------------------------------------------
struct InverseStreams
{
  Range y[2];                              // the samples the last synthesis on each row stream writes
  bool used[2] = {false, false};
  int last = 1;
  struct Pick { int stream = 0; bool wait_other = false; };
  // the other stream than the previous synthesis -- the same one if the two write overlapping samples; what the OTHER row
  // stream still has outstanding (samples this call overwrites, or a matrix reinterpreted from them) is waited for
  Pick pick(const Range& yr, const Range& in) const
  {
    Pick p;
    p.stream = overlap(yr, y[last]) ? last : (last ^ 1);
    const int so = p.stream ^ 1;
    p.wait_other = used[so] && (overlap(yr, y[so]) || overlap(in, y[so]));
    return p;
  }
  void launched(int stream, const Range& yr) { used[stream] = true; last = stream; y[stream] = yr; }
};

// ---- calls of one time chunk: time parts -------------------------------------------------------------------------------------
// forward_hop2_kernel: every (tile, part) a workgroup on a CU of its own; as many parts as leave every workgroup a CU, at
// least 12 samples each; a synchronous call waits for the completion word, which the LAST of all workgroups sets after a
// ticket each: at most 4 parts there, 8 otherwise (profiles/r05_hop_time_parts.txt)
struct HopParts { unsigned parts = 1, part_len = 0; };
inline HopParts hop_parts(size_t n, unsigned long long tile_waves, int compute_units, long forced, bool waits_for_word)
{
  HopParts h; h.parts = 1; h.part_len = (unsigned)n;
  if (forced == 1 || n < 24) return h;
  const size_t room = std::max<size_t>(1, (size_t)std::max(compute_units, 1) / (size_t)std::max<unsigned long long>(1, tile_waves));
  const size_t most = waits_for_word ? 4 : 8;
  size_t parts = forced > 1 ? (size_t)forced : std::min<size_t>({most, room, n / 12});
  parts = std::max<size_t>(1, std::min(parts, n));
  const size_t plen = (n + parts - 1) / parts;
  h.part_len = (unsigned)plen; h.parts = (unsigned)((n + plen - 1) / plen);
  return h;
}

// ---- synthesis in the reference's order: rows per wave ---------------------------------------------------------------------
// 32 rows per wave for long FD-double calls, 16 for FD float and medium calls, 4 with an 8-deep ring below 64 Ki rows, one
// wave per row up to 1024 rows; where 4 rows per wave need a second, partly filled round of the chip and 8 rows per wave fit
// in one (capacities from the occupancy API), the 8-row form (profiles/r04_synthesis_rows_per_wave.txt)
inline long inverse_rows_per_wave(size_t total_rows, size_t fd_bytes, long forced, size_t capacity4, size_t capacity8, bool with_operation)
{
  long rw = forced > 0 ? forced : (total_rows <= 1024 ? 1 : total_rows < 65536 ? 4 : ((fd_bytes == 8 && total_rows >= (size_t)32 * 8192) ? 32 : 16));
  if (!with_operation && forced <= 0 && rw == 4 && capacity4 && capacity8)
  {
    const size_t groups4 = (total_rows + 3) / 4, groups8 = (total_rows + 7) / 8;
    if (groups4 > capacity4 && groups8 <= capacity8) rw = 8;
  }
  if (with_operation && rw != 1) rw = 16;                    // one streaming instantiation with the operation built in
  return rw;
}

// ---- which of several bit-identical kernel forms is the fastest HERE: measured on the calls themselves ---------------------
// The streaming synthesis has forms that give the same bits and differ by a few per cent either way from box to box (rows per
// wave, bytes per row segment, tree sum with the rounding-interval proof): round 4's fixed switch points were right on one
// lease and wrong on the next.  A host that repeats a call shape lets the plan find out: the first calls of the shape take
// the candidate forms in turn, each bracketed by a pair of events (two samples per form, the smaller one counts), then the
// fastest form serves the shape.  Until a sample's events have completed no new trial starts (the calls in between take
// form 0, the static choice, untimed); a new shape starts over.
struct FormTuner
{
  static constexpr int kMax = 6, kSamples = 2;
  size_t key = 0;
  int count = 0, chosen = -1;
  int samples[kMax] = {};
  bool inflight[kMax] = {};                                // a timed call of this form has been launched and not yet reported
  float best[kMax] = {};
  void reset(size_t k, int candidates)
  {
    key = k; count = candidates < 1 ? 1 : (candidates > kMax ? kMax : candidates); chosen = count == 1 ? 0 : -1;
    for (int i = 0; i < kMax; ++i) { samples[i] = 0; best[i] = 0.f; inflight[i] = false; }
  }
  // the form this call takes; timed: the caller brackets the launch with the form's pair of events, calls launched() and
  // reports the time once the events have completed (a host that queues calls faster than they run has several trials in
  // flight, one per form)
  int next(bool can_time, bool& timed)
  {
    timed = false;
    if (chosen >= 0) return chosen;
    int want = -1, open = 0;
    for (int i = 0; i < count; ++i)
    {
      if (samples[i] < kSamples) ++open;
      if (samples[i] < kSamples && !inflight[i] && (want < 0 || samples[i] < samples[want])) want = i;
    }
    if (open == 0)
    {
      chosen = 0;
      for (int i = 1; i < count; ++i) if (best[i] < best[chosen]) chosen = i;
      return chosen;
    }
    if (!can_time || want < 0) return 0;                      // (every open form is in flight: the static form, untimed)
    timed = true;
    return want;
  }
  void launched(int form) { if (form >= 0 && form < count) inflight[form] = true; }
  void report(int form, float ms)
  {
    if (form < 0 || form >= count) return;
    inflight[form] = false;
    if (!(ms > 0.f)) return;
    best[form] = samples[form] == 0 ? ms : (ms < best[form] ? ms : best[form]);
    ++samples[form];
  }
};

// A host may alternate between a few call lengths (hops of two sizes; staged host-pointer calls that end in a shorter tail segment): one tuner per
// kind of call would start over at candidate 0 on every change of shape and never settle.  A small table of tuners keyed by the shape, least recently
// used replaced: a shape that has decided stays decided while up to kSlots shapes interleave.
struct TunerTable
{
  static constexpr int kSlots = 3;
  FormTuner slot[kSlots];
  unsigned long stamp[kSlots] = {};                        // 0: never used
  unsigned long clock = 0;
  int find(size_t key, int candidates)
  {
    const int count = candidates < 1 ? 1 : (candidates > FormTuner::kMax ? FormTuner::kMax : candidates);
    int lru = 0;
    for (int i = 0; i < kSlots; ++i)
    {
      if (stamp[i] && slot[i].key == key && slot[i].count == count) { stamp[i] = ++clock; return i; }
      if (stamp[i] < stamp[lru]) lru = i;
    }
    slot[lru].reset(key, candidates);
    stamp[lru] = ++clock;
    return lru;
  }
  void reset_all() { for (int i = 0; i < kSlots; ++i) { slot[i].reset(0, 1); stamp[i] = 0; } clock = 0; }
};

// ---- pipelined analyses: where they pay ------------------------------------------------------------------------------------------
// The next call's workgroups fill what a launch leaves idle -- the launch gap, the prologue of the self-carried chunks, the ragged end: a fixed 20-30 us
// per call.  Interleaved in one process on two equally placed matrices (profiles/r06_pipelined_calls.txt): n = 24 000 +12 %, n = 48 000 +9 % (82 against
// 75 % of the HBM peak), n = 90 000 +6 %, n = 131 072 +11 % (85 against 76 %), n = 1e6 a tie (85 %: 2.4 ms per call amortise what pipelining hides, and the
// call's own cut -- two rounds of shorter chunks -- is as good as the pipelined one).  So: calls of less than 2^29 bins (half a million rows of 1024 bins,
// about a millisecond of store stream) by default.
// option: 0 never, 1 (default) calls below that size, 2 calls of any length.
constexpr size_t kPipelineBinsMax = (size_t)1 << 29;
inline bool pipeline_pays(const ChunkQuery& q, long option)
{
  if (option <= 0) return false;
  if (option >= 2) return true;
  return std::max<size_t>(q.channels, 1) * q.n * q.nbins < kPipelineBinsMax;
}

// ---- synthesis: are the matrix' loads non-temporal? ------------------------------------------------------------------------------
// Beyond 256 MiB (what fits the Infinity Cache reads faster through it).  Round 4 had stopped at 4 GiB, where non-temporal loads of a just-written
// matrix were 5 % slower; round 5 found why -- a non-temporal load of a line that sits DIRTY in the cache is slow -- and reads the rows that may be
// dirty with ordinary loads (inverse_ordinary_rows below): with that, non-temporal loads win on every matrix measured, just written or only read, by
// 3-10 % (profiles/r05_synthesis_streaming_loads_big.txt, r05_after_write_16gb.txt).  forced: the option (-1 = by size).
inline bool inverse_streaming_loads(size_t matrix_bytes, long forced)
{
  if (forced >= 0) return forced != 0;
  return matrix_bytes > ((size_t)256 << 20);
}

// ... and how many of the rows a synthesis reads FIRST (the matrix' end) take ordinary loads whatever the kind of load: a non-temporal load of a line that
// sits dirty in the 256 MiB Infinity Cache (what an analysis wrote last) is slow; ordinary loads of such lines are not, and 1.5 GB of them push everything
// dirty out on the way.  n = 1e6 x 1024 (16.4 GB), rows in step: after the analysis 2.74 (all non-temporal) / 2.75 (all ordinary) -> 2.55 ms, only read 2.45
// either way (profiles/r05_after_write_16gb.txt).  Matrices from 6 GiB on (below, none: 706 MB ... 2.1 GB read best without; around 4 GiB it depends on
// the form, which the tuner finds out); forced_mb: the option (-1 = this rule, 0 = none).
inline size_t inverse_ordinary_rows(size_t matrix_bytes, size_t row_bytes, long forced_mb)
{
  if (row_bytes == 0) return 0;
  const size_t mb = forced_mb >= 0 ? (size_t)forced_mb : (matrix_bytes >= ((size_t)6 << 30) ? (size_t)1536 : 0);
  return mb ? ((mb << 20) + row_bytes - 1) / row_bytes : 0;
}

// ---- fused call: waves of a workgroup and bins per lane (1, 2, 4) -----------------------------------------------------------
struct ProcessGeometry { long waves = 1, slots = 1; };
inline ProcessGeometry process_geometry(size_t nbins, size_t channels, size_t n, bool fused, size_t fd_bytes, long forced_slots)
{
  ProcessGeometry g;
  g.waves = std::min<long>(kRowWaves, (long)((nbins + kLanes - 1) / kLanes));
  const long want = forced_slots > 0 ? forced_slots : ((fused && fd_bytes == 8 && nbins >= 256) ? ((nbins >= 512 && channels * n >= 400000) ? 4 : 2) : 1);
  if (want > 1) g.waves = std::max(1L, std::min(g.waves, (long)((nbins + (size_t)(kLanes * want) - 1) / (size_t)(kLanes * want))));
  g.slots = (long)((nbins + (size_t)g.waves * kLanes - 1) / ((size_t)g.waves * kLanes));
  return g;
}

// ---- synchronous completion -----------------------------------------------------------------------------------------------------
// A call cannot end before its bytes have moved at the chip's peak rate (the spec figure: a LOWER bound of the time, not a
// tuned one): until then the host spins on its own clock, then it polls the stream; never a sleeping wait while the call can
// still be running (it wakes up 9 us late on one box and 45 us late on another, profiles/r05_sync_completion.txt)
constexpr double kPeakBytesPerUs = 8.0e6;                    // HBM3E, 8 TB/s
struct SyncWait { double quiet_us = 0, budget_us = 0; };
inline SyncWait sync_wait(size_t hbm_bytes)
{
  SyncWait w;
  const double floor_us = (double)hbm_bytes / kPeakBytesPerUs;
  w.quiet_us = floor_us > 8.0 ? floor_us : 0.0;
  w.budget_us = std::min(20000.0, 200.0 + 4.0 * floor_us);
  return w;
}

// ---- host-pointer staging: rows per time segment ---------------------------------------------------------------------------
inline size_t stage_rows(size_t n, size_t row_bytes, size_t stage_bytes)
{
  const size_t seg = std::max<size_t>(1, stage_bytes / std::max<size_t>(row_bytes, 1));
  return std::min(seg, std::max<size_t>(n, 1));
}

// ---- host copies through pinned pieces: the slot ring --------------------------------------------------------------------------
// A copy of `bytes` bytes travels in pieces of `piece` bytes through a ring of `slots` pinned slots; piece i uses slot i % slots,
// so piece i may enter its slot once piece i - slots has left it.  Both directions are the same pipeline of two stages per
// piece: to the device  FILL (a host thread copies the caller's bytes into the slot) -> SEND (DMA out of the slot);
//        to the host    SEND (DMA into the slot) -> DRAIN (a host thread copies the slot into the caller's bytes).
struct PieceRing
{
  size_t bytes = 0, piece = 1;
  unsigned slots = 1;
  PieceRing(size_t total, size_t piece_bytes, unsigned ring_slots) : bytes(total), piece(std::max<size_t>(piece_bytes, 1)), slots(std::max(ring_slots, 1u)) {}
  size_t pieces() const { return (bytes + piece - 1) / piece; }
  size_t offset(size_t i) const { return i * piece; }
  size_t length(size_t i) const { return std::min(piece, bytes - std::min(bytes, i * piece)); }
  unsigned slot(size_t i) const { return (unsigned)(i % slots); }
  // the piece that must have left piece i's slot before i may enter it, or (size_t)-1
  size_t predecessor(size_t i) const { return i >= slots ? i - slots : (size_t)-1; }
};

}  // namespace logic
}  // namespace sdfthip
