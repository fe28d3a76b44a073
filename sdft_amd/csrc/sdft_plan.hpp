// sdft_plan.hpp -- host side of the HIP engine: plan tables, device-resident stream state,
// chunk geometry, kernel launches.  One instance per (TD, FD) pair, see sdft_capi.inc.
// Citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_kernels.hpp"
#include "sdft_forward_rows_f32.hpp"      // not part of the run-time-compiled text: plain analysis only
#include "sdft_resident.hpp"              // likewise: the opt-in resident kernel of the hop loop
#include "sdft_plan_logic.hpp"            // every decision that needs no HIP call (unit-tested on the CPU under sanitizers)
#include "sdft_host_io.hpp"               // the caller's host memory: classification, registration, copies through pinned slots

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <chrono>
#include <string>
#include <vector>

#pragma clang fp contract(off)

namespace sdfthip {

// ---- error channel (the reference has none: void returns, sdft.h:413-687) -----------------
void set_error(const char* what, const char* detail);   // sdft_common.hip
void set_warning(const char* what, const char* detail); // sdft_common.hip: something the host may want to know about a call that succeeded
bool lane_selftest();                                   // sdft_common.hip
// run-time compilation of a host's own spectral operation (sdft_common.hip)
bool rtc_kernel(const char* expr, const char* name_expr, int device, hipFunction_t* fn);
bool rtc_compile(const char* expr, const char* name_expr, const char* arch, std::string& lowered_name, std::vector<char>& code);


// ---- plan tables: the reference's own expressions (sdft.h:422-423, :439-446) evaluated on the
// host with the host libm, so the device tables are bit-identical to the oracle's ------------
template <typename FD> static inline FD host_cos(FD a);
template <> inline float  host_cos<float>(float a)   { return cosf(a); }
template <> inline double host_cos<double>(double a) { return cos(a); }
template <typename FD> static inline FD host_acos(FD a);
template <> inline float  host_acos<float>(float a)   { return acosf(a); }
template <> inline double host_acos<double>(double a) { return acos(a); }

// polar(r, t) of sdft.h:333-348.  The canonical oracle build (gcc -O2) merges the
// cos(t)/sin(t) pair into one glibc sincos() call, whose results differ from separate calls in
// the last bit for a few arguments; calling sincos explicitly keeps the tables bit-identical.
template <typename FD> static inline cx<FD> host_polar(FD r, FD t);
template <> inline cx<float> host_polar<float>(float r, float t)
{
  float s, c; sincosf(t, &s, &c); return cmake<float>(r * c, r * s);
}
template <> inline cx<double> host_polar<double>(double r, double t)
{
  double s, c; sincos(t, &s, &c); return cmake<double>(r * c, r * s);
}

template <typename FD>
struct Tables
{
  FD aweight, sweight;
  std::vector<cx<FD>> tw, syn, wtab;

  void build(size_t nbins, double latency)
  {
    aweight = (FD)(1) / (nbins * 2);                                    // :422
    sweight = (FD)(2);                                                  // :423
    tw.resize(nbins); syn.resize(nbins); wtab.resize(nbins * 2);
    const FD omega = (FD)(-2) * host_acos<FD>((FD)(-1)) / (nbins * 2);  // :439
    const FD gain = (FD)(+2) / ((FD)(1) - host_cos<FD>((FD)(omega * nbins * latency)));   // :440
    for (size_t k = 0; k < nbins; ++k)
    {
      const FD a = omega * k;                                           // :444
      const FD s = (FD)(omega * k * nbins * latency);                   // :445 (fd products, then * double)
      tw[k] = host_polar<FD>((FD)(1), a);
      syn[k] = host_polar<FD>(gain, s);
    }
    // W[j] = exp(-i*pi*j/N), j in [0, 2N): closed form of fid after j mod 2N rotations of bin 1
    // (only used to seed time chunks in the fast-carry mode)
    for (size_t j = 0; j < nbins * 2; ++j)
    {
      const double a = -3.14159265358979323846 * (double)j / (double)nbins;
      wtab[j] = (j < nbins) ? tw[j] : cmake<FD>((FD)cos(a), (FD)sin(a));
    }
  }
};

// the constants sdft_plan_logic.hpp is written against are the kernels'
static_assert(logic::kLanes == kWave && logic::kRowWaves == kRowWavesMax && logic::kRowSlots == kRowSlotsMax && logic::kTimeGroup == kGroup &&
              logic::kSumBlockLen == kSumBlock && logic::kHopSamples == kHopMax && kGroup % kRowGroup == 0, "sdft_plan_logic.hpp and the kernels disagree");
static_assert(logic::kWindowHann == WIN_HANN && logic::kWindowBlackman == WIN_BLACKMAN && logic::kWindowBoxcar == WIN_BOXCAR, "window numbering");
static_assert(sizeof(logic::Radices) == sizeof(RadixList), "radix lists");

enum CarryMode : int { CARRY_FAST = 0, CARRY_EXACT = 1 };

enum ProfileStage : int { ST_DELTA = 0, ST_CARRY = 1, ST_FORWARD = 2, ST_INVERSE = 3, ST_COUNT = 4 };

template <typename TD, typename FD>
class Plan
{
 public:
  using fdx = cx<FD>;

  size_t nbins = 0, channels = 1;
  int window = WIN_HANN;
  double latency = 1.0;
  Tables<FD> tab;

  int device = 0;
  int compute_units = 256;       // CUs of the plan's device (one row-group workgroup each)
  hipStream_t stream = nullptr;
  bool own_stream = false, async = false;

  // options
  int carry_mode = sizeof(FD) == 8 ? CARRY_FAST : CARRY_EXACT;
  static constexpr size_t kFlagMax = (size_t)1 << 24;       // bin-samples up to which a row-group analysis call signals its own completion
#ifdef SDFT_SELF_STAMPS
  long opt_self_stamps = 0;                                // development builds: device address of 8 stamp words
#endif
  long opt_inverse_verify = 1, last_inverse_form = 0;      // launch_inverse
  static constexpr size_t kInverseVerifyMax = 500000;      // rows up to which the tree sum with the rounding-interval proof serves sdft_isdft_n (beyond: the streaming kernel)
  bool rtc_failed = false;                                 // launch_inverse returns void: a failed run-time compilation is reported here
  std::string user_expr;                                   // sdft_hip_process_n with an expression: the statements of the call in flight
  template <typename T> static const char* type_name() { return sizeof(T) == 8 ? "double" : "float"; }
  // FD float plans take the chunk-parallel carries too: 2x faster on long calls and closer to the double-precision
  // result than the reference's float arithmetic is, but not within 1e-4 of it (the float reference itself drifts
  // 2e-4 of the largest bin per 262144 samples: profiles/r03_float_parallel_carries.txt)
  long opt_float_parallel = 0;
  long opt_chunk = 0;            // forced chunk length (0 = heuristic)
  long opt_interior = 0;         // forced interior lanes per wave (0 = maximum)
  static constexpr size_t kDefaultStageBytes = (size_t)1 << 30;
  size_t stage_bytes = kDefaultStageBytes;   // host-pointer path: staging segment size
  int profile = 0;               // 0 off, 1 = events around every stage, 2 = forward/inverse kernels only
  long opt_rows_kernel = 1;      // use the row-group forward kernel when the row fits one workgroup
  long opt_xcd_map = 1;          // every XCD takes a contiguous eighth of a launch's (channel, chunk) workgroups (ForwardArgs::xcd_map)
  long opt_pinned_io = 1;        // small host sample buffers travel through a pinned scratch the kernels access directly
  long opt_pointers = 0;         // 0 = ask the runtime on every call (hipPointerGetAttributes, ~0.1 us), 1 = all device, 2 = all host
  long opt_inverse_rows = 0;     // rows per wave of the exact inverse (0 = heuristic; 4, 8, 16, 32)
  size_t inverse_capacity[2] = {0, 0};   // waves the chip holds at once of the 4-row and the 8-row form (occupancy x CUs), 0: not asked yet
  // Synthesis reads the matrix with non-temporal loads (-1 = by size, 0 / 1).  The host pattern is analysis -> synthesis of the
  // same matrix: the analysis leaves the matrix's last 256 MiB dirty in the Infinity Cache, and ordinary loads of the rest
  // push those lines out to HBM while they read -- streaming loads leave them where the next analysis overwrites them.
  // Measured after a write (profiles/r04_synthesis_streaming_loads.txt): 706 MB f64f64 217 -> 176 us, f32f64 193 -> 153 us,
  // 2.1 GB 461 -> 392 us; a matrix that fits the cache (192 MB) 51 -> 58 us and 8-16 GB +5 %: hence the size window.
  long opt_inverse_nt = -1;
  long opt_inverse_nt_skip_mb = -1;                         // -1: 1536 MB of matrices from 6 GiB on (logic::inverse_ordinary_rows); 0: none
  // long synthesis calls: the fastest of the bit-identical streaming forms is found on the host's own calls (launch_inverse)
  long opt_inverse_tune = 1, last_inverse_tuned = 0;
  // (two tuners: a synthesis that follows an analysis call reads a matrix whose tail is still dirty in the Infinity Cache, one that
  // follows another synthesis does not -- which form and which kind of load is fastest differs between the two, and a host may do both)
  logic::TunerTable inv_tunes[2];                           // (each a few shapes: a host that alternates call lengths keeps what it has decided)
  bool inv_after_write = false;                              // the synthesis call being launched directly follows an analysis call
  hipEvent_t tune_evs[2][logic::TunerTable::kSlots][logic::FormTuner::kMax][2] = {};     // a pair of events per candidate form and tuner
  bool ensure_tune_events()
  {
    if (tune_evs[0][0][0][0]) return true;
    for (auto& kind : tune_evs)
      for (auto& tuner : kind)
        for (auto& pair : tuner)
          for (hipEvent_t& e : pair)
            if (hipEventCreate(&e) != hipSuccess)
            {
              (void)hipGetLastError(); e = nullptr;
              destroy_tune_events();
              return false;
            }
    return true;
  }
  void destroy_tune_events()
  {
    for (auto& kind : tune_evs) for (auto& tuner : kind) for (auto& q : tuner) for (hipEvent_t& f : q) if (f) { (void)hipEventDestroy(f); f = nullptr; }
  }
  long opt_exact_inverse = 1;    // inverse sums bins in the reference's order (bit-identical)
  long opt_row_slots_max = 2;    // largest slots-per-lane the row-group kernel may use (1 = rows <= 1024*BPL only)
  long opt_fused = 1;            // fused multiply-add arithmetic in the chunk-parallel FD double path
  long last_fused = 0;
  long opt_fft_carry = 1;        // FFT form of the chunk partial sums when 2N is a power of two
  long opt_hop_kernel = 1;       // single-chunk calls: fused delta + forward launch (forward_hop_kernel)
  long opt_fused_exact = -1;     // fused analysis->synthesis: 0 = folded / tree sum; 1 = bins summed in the reference's order
                                 // (the fastest route that gives those bits), 2 = in that order by the fused kernel;
                                 // -1 = in order exactly when the host asked for exact carries at FD double (carry = 1)
  long last_fused_exact = 0, last_fused_fold = 0, last_process_path = 0;   // last_process_path: 1 fused kernel, 2 hop pair, 3 two-pass segments
  long opt_spin = 1;             // synchronous short calls poll the stream instead of sleeping on it
  long opt_self = 1;             // chunk-parallel FD double calls, 2N a power of two: self-carried chunks (no pre-pass launches)
  static constexpr size_t kSelfMax = (size_t)1 << 19;      // ... for calls of up to this many samples per channel (the fold of a chunk's past grows with n)
  long last_self = 0;

  long last_kernel = 0;          // 1 = forward_kernel (independent tiles), 2 = forward_rows_kernel

  // device-resident stream state
  DevBuf<fdx> d_tw, d_syn, d_wtab;
  // acc / fid / delay line are double-buffered: single-chunk calls read one set and write the other
  DevBuf<fdx> d_accs[4], d_fids[4];                        // state slots: a call reads [st_cur], writes [st_cur ^ 1] (pipelined calls: the next of the ring of four)
  int st_cur = 0;
  fdx* acc_p() { return d_accs[st_cur].p; }
  fdx* fid_p() { return d_fids[st_cur].p; }
  DevBuf<TD> d_hist[4];
  int hist_cur = 0;
  size_t cursor = 0;             // reference cursor (:153)

  // exact-carry mode overlaps the serial pass with the forward kernel: the pass runs on `aux`
  // in time segments, each segment's forward launch on `stream` waits for its event
  hipStream_t aux = nullptr;
  std::vector<hipEvent_t> seg_events;
  hipEvent_t ev_delta = nullptr;
  DevBuf<fdx> d_run_acc[2], d_run_fid[2];
  long opt_segments = 0;         // 0 = heuristic
  long last_segments = 1;

  // exact carries, chain form: fid at every fseed_L-th cursor of the canonical rotation sequence
  // (built on first use); fid_canonical = the stream's fid state is on that sequence (false after
  // a chunk-parallel call seeded fid from the closed-form table, or after set_state, until the
  // next roll-over puts it back to exactly 1)
  DevBuf<fdx> d_fseed;
  unsigned fseed_L = 0;
  bool fid_canonical = true;
  long opt_chain = 1;            // 0 = always the serial pass (carry_exact_kernel), 1 = heuristic, 2 = chain form whenever possible
  long opt_chain_L = 0, opt_chain_debug = 0;
  long last_hop_pipe = 0;
  long opt_hop_parts = 0, last_hop_parts = 1;   // ... in time parts (0 = by the launch's size, 1 = never, n = that many)
  long opt_fold = 1;             // fused call, tree-sum flavour: window, operation and synthesis folded into per-bin coefficients
  long opt_relay_waves = 0;      // waves per workgroup of the relay form (0 = default)
  DevBuf<unsigned long long> d_chain_stats;
  long last_chain = 0;

  // workspace
  DevBuf<FD> d_delta;
  DevBuf<fdx> d_carry, d_seed;
  DevBuf<TD> d_stage_td;
  DevBuf<fdx> d_stage_fdx;
  DevBuf<fdx*> d_rowptr;

  // profile: HIP events on the plan's stream, one pair per stage launch, collected lazily so
  // that back-to-back asynchronous calls are never serialised by the measurement
  std::vector<hipEvent_t> ev_pool[ST_COUNT];
  size_t ev_used[ST_COUNT] = {};
  double prof_ms[ST_COUNT] = {};
  long prof_calls[ST_COUNT] = {};

  // last launch geometry (introspection for tests / bench)
  long last_chunks = 0, last_chunk_len = 0, last_tiles = 0, last_interior = 0, last_pipelined = 0;

  bool create(size_t dftsize, int win, double lat, size_t nch)
  {
    nbins = dftsize; window = win; latency = lat; channels = nch ? nch : 1;
    if (window < 0 || window > 3) window = WIN_BOXCAR;      // reference: unknown window -> default branch (:394)
    tab.build(nbins, latency);
    SDFT_TRY(hipGetDevice(&device));
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) compute_units = cus; else (void)hipGetLastError(); }
    if (!lane_selftest()) return false;
    SDFT_TRY(hipStreamCreate(&stream));
    own_stream = true;
    if (const char* e = getenv("SDFT_HIP_CARRY"))
      carry_mode = (!strcmp(e, "exact") || !strcmp(e, "1")) ? CARRY_EXACT : CARRY_FAST;
    if (sizeof(FD) == 4) carry_mode = CARRY_EXACT;         // float FD follows the reference's rounding (unless "float_carry_parallel")
    if (const char* e = getenv("SDFT_HIP_CHUNK")) opt_chunk = atol(e);
    if (const char* e = getenv("SDFT_HIP_INTERIOR")) opt_interior = atol(e);
    if (nbins == 0) return true;
    const size_t nb = nbins, span = 2 * nbins;
    if (!d_tw.reserve(nb) || !d_syn.reserve(nb) || !d_wtab.reserve(span)) return false;
    for (int q = 0; q < 4; ++q)
      if (!d_accs[q].reserve(channels * nb) || !d_fids[q].reserve(channels * nb) || !d_hist[q].reserve(channels * span)) return false;
    if (!to_device(d_tw.p, tab.tw.data(), nb * sizeof(fdx))) return false;
    if (!to_device(d_syn.p, tab.syn.data(), nb * sizeof(fdx))) return false;
    if (!to_device(d_wtab.p, tab.wtab.data(), span * sizeof(fdx))) return false;
    return reset();
  }

  void destroy()
  {
    (void)hipSetDevice(device);
    (void)resident_retire();
    resident_release();
    (void)pipe_join();
    if (stream) (void)hipStreamSynchronize(stream);
    d_tw.release(); d_syn.release(); d_wtab.release();
    release_pipe();
    for (int q = 0; q < 4; ++q) { d_accs[q].release(); d_fids[q].release(); d_hist[q].release(); }
    d_delta.release(); d_carry.release(); d_seed.release();
    d_stage_td.release(); d_stage_fdx.release(); d_rowptr.release(); d_fseed.release();
    d_gain.release(); d_stage_y.release(); d_chain_stats.release();
    d_alpha.release(); d_beta.release(); d_partial.release(); d_tickets.release();
    if (h_done_flag) { (void)hipHostFree(h_done_flag); h_done_flag = nullptr; }
    if (h_io) { (void)hipHostFree(h_io); h_io = nullptr; d_io = nullptr; }
    io.release_pin();
    if (h_status) { (void)hipHostFree(h_status); h_status = nullptr; }
    io.forget_host_buffers();
    if (d_started) { (void)hipFree(d_started); d_started = nullptr; }
    d_ready.release();
    d_done_count.release(); d_walked.release();
    d_run_acc[0].release(); d_run_acc[1].release(); d_run_fid[0].release(); d_run_fid[1].release();
    if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); aux = nullptr; }
    for (hipEvent_t e : seg_events) (void)hipEventDestroy(e);
    seg_events.clear();
    if (ev_delta) { (void)hipEventDestroy(ev_delta); ev_delta = nullptr; }
    destroy_tune_events();
    for (int st = 0; st < ST_COUNT; ++st)
    {
      for (hipEvent_t e : ev_pool[st]) (void)hipEventDestroy(e);
      ev_pool[st].clear(); ev_used[st] = 0;
    }
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }

  // a host driving several GPUs from one process may have switched the current device
  std::chrono::steady_clock::time_point call_start;         // when the entry point in progress began (finish() counts from here)
  bool bind()
  {
    flag_pending = false; flag_wanted = false; pipe_allowed = false; call_start = std::chrono::steady_clock::now();
    SDFT_TRY(hipSetDevice(device));
    return res.alive ? resident_retire() : true;             // (option "resident": whatever is not one of its calls ends the kernel first)
  }

  // sdft.h:517-529
  bool reset()
  {
    if (!pipe_join()) return false;
    cursor = 0; hist_cur = 0; st_cur = 0; fid_canonical = true;
    if (nbins == 0) return true;
    if (!bind()) return false;
    const size_t nb = nbins, span = 2 * nbins;
    SDFT_TRY(hipMemsetAsync(d_hist[0].p, 0, channels * span * sizeof(TD), stream));
    SDFT_TRY(hipMemsetAsync(acc_p(), 0, channels * nb * sizeof(fdx), stream));
    std::vector<fdx> ones(channels * nb, cmake<FD>((FD)1, (FD)0));
    if (!to_device(fid_p(), ones.data(), ones.size() * sizeof(fdx))) return false;
    SDFT_TRY(hipStreamSynchronize(stream));
    return true;
  }

  bool set_stream(hipStream_t s)
  {
    if (!pipe_join()) return false;
    if (stream) SDFT_TRY(hipStreamSynchronize(stream));
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    stream = s; own_stream = false;
    return true;
  }

  bool synchronize()
  {
    if (res.alive && !resident_retire()) return false;
    if (!pipe_join()) return false;
    SDFT_TRY(hipStreamSynchronize(stream));
    if (status_armed)
    {
      if (aux) SDFT_TRY(hipStreamSynchronize(aux));
      if (ring_gave_up())
      {
        set_error("carry_ring_kernel", "a poll loop timed out in an asynchronous call: its output and the stream state are invalid (reset the plan or restore a state)");
        (void)collect_profile();
        return false;
      }
    }
    return collect_profile();
  }

  // ---- profiling ---------------------------------------------------------------------------
  bool prof_on(int st) const { return profile == 1 || (profile == 2 && (st == ST_FORWARD || st == ST_INVERSE)); }
  bool prof_begin(int st, hipStream_t on = nullptr)
  {
    if (!prof_on(st)) return true;
    if (ev_used[st] + 2 > ev_pool[st].size())
    {
      hipEvent_t a, b;
      SDFT_TRY(hipEventCreate(&a)); SDFT_TRY(hipEventCreate(&b));
      ev_pool[st].push_back(a); ev_pool[st].push_back(b);
    }
    SDFT_TRY(hipEventRecord(ev_pool[st][ev_used[st]], on ? on : stream));
    return true;
  }
  bool prof_end(int st, hipStream_t on = nullptr)
  {
    if (!prof_on(st)) return true;
    SDFT_TRY(hipEventRecord(ev_pool[st][ev_used[st] + 1], on ? on : stream));
    ev_used[st] += 2;
    return true;
  }
  bool collect_profile()
  {
    for (int st = 0; st < ST_COUNT; ++st)
    {
      for (size_t i = 0; i + 1 < ev_used[st]; i += 2)
      {
        SDFT_TRY(hipEventSynchronize(ev_pool[st][i + 1]));
        float ms = 0.f;
        SDFT_TRY(hipEventElapsedTime(&ms, ev_pool[st][i], ev_pool[st][i + 1]));
        prof_ms[st] += ms; prof_calls[st] += 1;
      }
      ev_used[st] = 0;
    }
    return true;
  }

  // ---- geometry and time chunking: sdft_plan_logic.hpp ---------------------------------------------------------------
  static int bins_per_lane() { return logic::bins_per_lane(sizeof(fdx)); }
  int halo_bins() const { return logic::halo_bins(window); }
  int halo_lanes() const { return logic::halo_lanes(window, sizeof(fdx)); }
  long interior_lanes() const { return logic::interior_lanes(window, sizeof(fdx), opt_interior); }
  long tiles() const { return logic::tiles(nbins, window, sizeof(fdx), opt_interior); }
  bool rows_kernel_ok(bool row_pointers) const { return logic::rows_kernel_ok(nbins, sizeof(fdx), row_pointers, opt_rows_kernel != 0, opt_row_slots_max); }
  long row_slots() const { return logic::row_slots(nbins, sizeof(fdx)); }       // slots per lane (1 or 2) ...
  long row_waves() const { return logic::row_waves(nbins, sizeof(fdx)); }       // ... and physical waves of the row group
  logic::ChunkQuery chunk_query(size_t n, bool rows_kernel) const
  {
    logic::ChunkQuery q;
    q.n = n; q.channels = channels; q.nbins = nbins; q.rows_kernel = rows_kernel; q.exact = carry_mode == CARRY_EXACT; q.pipelined = pipe_this;
    q.forced_chunk = opt_chunk; q.row_waves = row_waves(); q.tiles = tiles(); q.compute_units = compute_units;
    return q;
  }
  void choose_chunks(size_t n, long& chunks, long& len, bool rows_kernel = false) const
  {
    const logic::Chunking c = logic::choose_chunks(chunk_query(n, rows_kernel));
    chunks = c.chunks; len = c.len;
  }

  // floor(2^32 / d) + 1, 0 for d <= 1: what flow_position divides a workgroup number by (ForwardArgs::inv_chunks, inv_channels)
  static unsigned inv32(unsigned d) { return d <= 1 ? 0u : (unsigned)((((unsigned long long)1) << 32) / d) + 1u; }
  // every launch is a 1-D grid (channels ride on grid.x); refuse what would not fit it
  static bool grid_fits(size_t blocks)
  {
    if (blocks <= 0x7fffffffull) return true;
    set_error("launch", "grid too large: channels x chunks x bins exceed 2^31 workgroups");
    return false;
  }

  // ---- exact carries, relay form: block length / seed table -----------------------------------------
  unsigned relay_block(long len) const { return logic::relay_block(nbins, len, sizeof(FD), sizeof(fdx), opt_chain_L); }
  // flow mode of the relay form (see forward_launch)
  long opt_relay_flow = 1, last_flow = 0;
  DevBuf<unsigned> d_ready;
  unsigned ready_seq = 0;
  unsigned* d_started = nullptr;      // device word every relay workgroup bumps when it has started
  unsigned started_target = 0;
  // The forward launch must not take CUs the relays still need (a forward workgroup that waits for a relay which cannot
  // start would be a deadlock until the bounded polls run out): a one-wave gate kernel on the forward stream polls the
  // word until the relays of the call are resident (round 3 used hipStreamWaitValue32 for this: the runtime's wait packet
  // took 135 us to notice -- profiles/r04_relay_gate_trace.txt)
  bool gate_ok()
  {
    if (d_started) return true;
    if (hipMalloc((void**)&d_started, 8) != hipSuccess) { (void)hipGetLastError(); d_started = nullptr; return false; }
    if (hipMemset(d_started, 0, 8) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_started); d_started = nullptr; return false; }
    started_target = 0;
    return true;
  }
  // relays (32 bins of a channel each), waves per relay.  (Two relays per workgroup -- the kernel has the form -- were measured
  // slower, 12 waves of products and two chains on one CU contend for issue: config 3 shape, 1.5 against 1.05 ms; the option is gone)
  static unsigned relay_waves_default() { return 8u; }
  long opt_relay_groups = 1;     // test hook: relays per workgroup (2: half as many CUs; FD float only)
  template <int L> unsigned relay_groups() const { return (unsigned)std::max(1L, std::min<long>(relay_limits<FD, L>::groups, opt_relay_groups)); }
  template <int L> unsigned relay_waves() const
  {
    const long mx = relay_limits<FD, L>::waves / (long)relay_groups<L>();
    return (unsigned)std::max(1L, std::min(mx, opt_relay_waves > 0 ? opt_relay_waves : (long)relay_waves_default()));
  }
  unsigned relay_groups_of(unsigned L) const { return L <= 64 ? relay_groups<64>() : relay_groups<128>(); }
  template <int L> bool launch_relay(ChainArgs<FD> cc, unsigned relays, hipStream_t on)
  {
    const unsigned waves = relay_waves<L>(), groups = relay_groups<L>();
    const unsigned blocks = (relays + groups - 1) / groups;
    cc.P = waves; cc.chunks_channels = (unsigned)channels;
    if constexpr (L * sizeof(FD) == 512)
    {
      // the longest block also exists as a measurement build
      if (cc.stats) { hipLaunchKernelGGL((carry_relay_kernel<FD, L, true>), dim3(blocks), dim3(kWave * waves * groups), 0, on, cc); SDFT_TRY(hipGetLastError()); return true; }
    }
    if constexpr (L * sizeof(FD) <= 512)
      hipLaunchKernelGGL((carry_relay_kernel<FD, L>), dim3(blocks), dim3(kWave * waves * groups), 0, on, cc);
    SDFT_TRY(hipGetLastError());
    return true;
  }
  bool ensure_fseed(unsigned L)
  {
    if (fseed_L == L && d_fseed.p) return true;
    if (!d_fseed.reserve((2 * nbins / L) * nbins)) return false;
    hipLaunchKernelGGL((fid_seed_kernel<FD>), dim3((unsigned)((nbins + kWave - 1) / kWave)), dim3(kWave), 0, stream,
                       (const fdx*)d_tw.p, d_fseed.p, (unsigned)nbins, L);
    SDFT_TRY(hipGetLastError());
    fseed_L = L;
    return true;
  }

  // ---- forward on device-resident buffers ------------------------------------------------
  // x: [channels] x n with stride x_stride; out: rows at out + ch*out_stride + t*N, or the row
  // pointer table `rows` (device array of channels*n device pointers)
  // Checked form (what every entry point calls).  The ring form of the exact carries ends its poll loops after a bounded
  // number of tries instead of hanging the GPU; a wave that ran out says so in a word of pinned host memory
  // (ChainArgs::status).  The state a call starts from stays intact until the next call (acc, fid and the delay line are
  // double-buffered, the cursor lives on the host), so a synchronous call that finds the word changed restores that
  // state and runs again with the serial pass: the outputs are the reference's bits either way, and
  // sdft_hip_last_error() tells the host that it happened.  Asynchronous calls are checked in synchronize(), which can
  // only report (the caller's buffers may have moved on).
  unsigned* h_status = nullptr;
  unsigned* d_status = nullptr;
  unsigned status_seen = 0;
  bool status_armed = false;       // a ring launch is in flight or unchecked
  long ring_recoveries = 0;
  bool ensure_status()
  {
    if (h_status) return true;
    if (hipHostMalloc((void**)&h_status, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); h_status = nullptr; return false; }
    *h_status = 0;
    if (hipHostGetDevicePointer((void**)&d_status, h_status, 0) != hipSuccess)
    {
      (void)hipGetLastError(); (void)hipHostFree(h_status); h_status = nullptr; return false;
    }
    return true;
  }
  // after the stream has drained: did a ring launch give up since the last look?
  bool ring_gave_up()
  {
    if (!status_armed || !h_status) return false;
    status_armed = false;
    const unsigned now = *(volatile unsigned*)h_status;
    if (now == status_seen) return false;
    status_seen = now;
    return true;
  }
  bool forward_device(size_t n, const TD* x, size_t x_stride, fdx* out, size_t out_stride, fdx* const* rows,
                      const FuseArgs<TD, FD>* fuse = nullptr)
  {
    const size_t cursor0 = cursor;
    const int st0 = st_cur, hist0 = hist_cur;
    const bool canon0 = fid_canonical;
    if (!forward_launch(n, x, x_stride, out, out_stride, rows, fuse))
    {
      // a launch that failed half-way (a host expression that does not compile, a grid that does not fit, an allocation)
      // must not leave the stream half-advanced: what was queued wrote the OTHER buffer set and the workspace only
      cursor = cursor0; st_cur = st0; hist_cur = hist0; fid_canonical = canon0;
      return false;
    }
    if (last_chain < 2 || async) return true;
    SDFT_TRY(hipStreamSynchronize(stream));
    if (aux) SDFT_TRY(hipStreamSynchronize(aux));
    if (!ring_gave_up()) return true;
    cursor = cursor0; st_cur = st0; hist_cur = hist0; fid_canonical = canon0;
    const long saved = opt_chain;
    opt_chain = 0;
    const bool ok = forward_launch(n, x, x_stride, out, out_stride, rows, fuse) && (hipStreamSynchronize(stream) == hipSuccess);
    opt_chain = saved;
    ++ring_recoveries;
    // a recovered call is a valid call: it goes to the warning channel (and the counter), not to the error channel, so a
    // host that checks sdft_hip_last_error() after every call does not feed the samples twice
    if (ok) set_warning("carry_relay_kernel", "a poll loop timed out; the call was re-run with the serial carry pass (results are valid)");
    else
    {
      cursor = cursor0; st_cur = st0; hist_cur = hist0; fid_canonical = canon0;
      set_error("carry_relay_kernel", "a poll loop timed out and the re-run with the serial carry pass failed");
    }
    return ok;
  }

  bool forward_launch(size_t n, const TD* x, size_t x_stride, fdx* out, size_t out_stride, fdx* const* rows,
                      const FuseArgs<TD, FD>* fuse = nullptr)
  {
    if (n == 0 || nbins == 0) return true;
    const size_t nb = nbins, span = 2 * nbins;
    SDFT_TRY(hipSetDevice(device));
    flag_pending = false;                                    // only the hop kernel signals its completion

    calls.on_analysis(fuse != nullptr);                      // (which kind of host is calling: logic::CallPattern)
    const bool use_rows = rows_kernel_ok(rows != nullptr);
    long chunks, len;
    // (the folded fused kernel and the row-group forward kernel have the self-carried form)
    const bool folded_fuse = fuse && !wants_reference_order() && !fuse->store && opt_fold && coeff_ready;
    // pipelined calls (forward_self): decided here because they take the self-carried form at any length and cut time differently
    pipe_this = false;
    uintptr_t out_lo = 0, out_hi = 0;
    if (!fuse && !rows && out && use_rows)
    {
      out_lo = reinterpret_cast<uintptr_t>(out); out_hi = out_lo + ((channels - 1) * out_stride + n * nb) * sizeof(fdx);
      // (calls of a few thousand rows gain a microsecond from it and cost the host seven runtime calls instead of one,
      // 19 against 3 us: n = 4096, m = 1024: 25.6 against 26.4 us per call; from n = 8192 on 30.4 against 32.9)
      pipe_this = calls.analysis_batch && pipe_wanted(nullptr) && self_eligible(n, false, true) && n < ((size_t)1 << 31) && channels * n * nb >= ((size_t)6 << 20) &&
                  !logic::overlap(out_lo, out_hi, prev_out) && logic::pipeline_pays(chunk_query(n, true), opt_pipeline);
    }
    bool self_form = self_eligible(n, fuse != nullptr, pipe_this) && (fuse ? folded_fuse : use_rows);
    if (self_form && fuse)
    {
      // the fused kernel folds into its transpose tiles: the 2N cells have to fit them, and it has one or two bins per lane
      long pw, ps;
      process_geometry(opt_fused != 0, pw, ps, n);
      self_form = ps <= 2 && self_cells() * sizeof(fdx) <= process_tiles_bytes((unsigned)(pw * kWave));
    }
    if (!self_form) pipe_this = false;
    if (out_hi) prev_out = logic::Range{out_lo, out_hi};
    choose_chunks(n, chunks, len, use_rows);
    const long ntiles = tiles(), inter = interior_lanes();
    last_kernel = use_rows ? 2 : 1;
    last_chunks = chunks; last_chunk_len = len; last_tiles = ntiles; last_interior = inter;
    last_segments = 1; last_fused = 0; last_self = 0; last_chain = 0;
    if (chunks == 1 && opt_hop_kernel && nbins >= 2 && !fuse) { if (!pipe_join()) return false; return forward_hop(n, x, x_stride, out, out_stride, rows); }

    const bool exact = (carry_mode == CARRY_EXACT);
    // self-carried chunks: every workgroup derives its carry-in from the raw samples (fold + one FFT in LDS) and forms
    // its own differences -- the call is ONE launch.  Kernels that have the form: the row-group forward kernel and the
    // folded fused kernel, FD double, 2N a power of two of at most 4096 cells.
    // (the fold of a chunk's past costs t0 / threads loads: hidden behind the other workgroups' row stores in the
    // analysis, which is bound by HBM -- n = 1e6: 2.885 -> 2.853 ms -- but not in the fused call, which is bound by
    // instruction issue: n = 48000: 45.9 -> 42.3 us, n = 131072: 99 -> 117 us; hence self_eligible's limit for it)
    // (2N = 2/3/5-smooth: five Stockham stages with table look-ups -- n = 12000, N = 1000, chunks of 64 samples: 47 us with
    // the pre-pass, 55 us self-carried; n = 48000, chunks of 192: 192 -> 173 us)
    const bool self = self_form && chunks > 1 && ((span & (span - 1)) == 0 || len > 64);
    last_self = self;
    if (self) return forward_self(n, x, x_stride, out, out_stride, chunks, len, fuse);
    if (!pipe_join()) return false;
    // exact carries: relay form (seed table + identical waves that take the blocks of cL steps in turn, a block's products in
    // registers) while the serial pass would leave most SIMDs idle; the plain serial pass when bins x channels already
    // fill the chip.  The chunk grid is shifted so that every chunk but the first starts on a block boundary of the
    // cursor (chunk j starts at sample j*len - shift; one more chunk may be needed for the tail)
    const size_t serial_waves = ((nb + kWave / 2 - 1) / (kWave / 2)) * channels;
    const bool chain_ok = exact && chunks > 1 && opt_chain && fid_canonical && (opt_chain >= 2 || serial_waves <= 1024);
    const unsigned cL = (chain_ok && n < ((size_t)1 << 31)) ? relay_block(len) : 0u;
    const bool use_chain = cL != 0;
    const unsigned shift = use_chain ? (unsigned)(cursor % cL) : 0u;
    if (shift) { chunks = (long)((n + shift + (size_t)len - 1) / (size_t)len); last_chunks = chunks; }

    if (!d_delta.reserve(channels * n + 128)) return false;     // + slack: the exact pass prefetches bursts past a run
    if (!d_carry.reserve(channels * (size_t)chunks * nb)) return false;
    last_chain = use_chain ? 3 : 0;
    if ((exact || chunks == 1) && !use_chain && !d_seed.reserve(channels * (size_t)chunks * nb)) return false;
    if (use_chain && !ensure_fseed(cL)) return false;

    // form of the chunk-parallel partial sums: FFT when 2N is a power of two or 2/3/5-smooth (and fits LDS) -- but
    // direct sums for chunks of up to 64 samples, which need no barriers (n = 1024 / 4096, N = 1024: 28.0 / 32.1 ->
    // 24.2 / 28.3 us per call even with one launch more) -- and direct sums for every other size
    enum { SUMS_DIRECT = 0, SUMS_FFT2 = 1, SUMS_FFT_MIXED = 2 };
    int sums_form = SUMS_DIRECT;
    RadixList rl; rl.count = 0;
    {
      const size_t span_bytes = span * sizeof(fdx);
      const bool pow2 = (span & (span - 1)) == 0 && span >= 2;
      if (!pow2) rl = smooth_radices(span);                           // other prime factors: direct sums
      const bool short_chunks = len <= 64 && opt_fft_carry != 2;
      if (opt_fft_carry && !short_chunks && pow2 && span_bytes <= (size_t)64 * 1024) sums_form = SUMS_FFT2;
      else if (opt_fft_carry && !short_chunks && rl.count > 0 && 2 * span_bytes <= (size_t)64 * 1024) sums_form = SUMS_FFT_MIXED;
    }
    // K0: differences + delay line -- unless the chunk-parallel carry kernel forms them itself
    const bool delta_in_carry = !exact && chunks > 1;
    if (!delta_in_carry)
    {
    if (!prof_begin(ST_DELTA)) return false;
    {
      const size_t work = std::max(n, span);
      const size_t per_ch = (work + kBlock - 1) / kBlock;
      if (!grid_fits(per_ch * channels)) return false;
      const bool single = (chunks == 1);
      hipLaunchKernelGGL((delta_kernel<TD, FD>), dim3((unsigned)(per_ch * channels)), dim3(kBlock), 0, stream, x, x_stride,
                         d_hist[hist_cur].p, d_hist[hist_cur ^ 1].p, d_delta.p, n, span,
                         (const fdx*)acc_p(), (const fdx*)fid_p(), single ? d_carry.p : (fdx*)nullptr, single ? d_seed.p : (fdx*)nullptr,
                         (unsigned)per_ch);
      SDFT_TRY(hipGetLastError());
    }
    if (!prof_end(ST_DELTA)) return false;
    }
    DeltaIn<TD, FD> din;
    din.x = delta_in_carry ? x : nullptr; din.x_stride = x_stride;
    din.hist_in = d_hist[hist_cur].p; din.hist_out = d_hist[hist_cur ^ 1].p; din.delta_out = d_delta.p;
    hist_cur ^= 1;

    // carries
    long segments = 1;
    bool flow = false;
    if (exact && chunks > 1)
    {
      // time segments: the serial pass of segment s+1 (few waves, latency-bound) runs on `aux`
      // while the forward kernel of segment s streams the matrix on `stream`;
      // up to 8 segments, each forward launch still filling the chip (>= 256 workgroups)
      const long launch_blocks = use_rows ? (long)channels * chunks : (long)channels * chunks * ntiles / kWavesPerBlock;
      segments = opt_segments > 0 ? opt_segments : std::max(1L, std::min(8L, launch_blocks / 256));
      // Relay form, flow mode: ONE relay launch for the whole call on `aux` and ONE forward launch whose workgroups wait
      // for their chunk's carries themselves (ForwardArgs::ready).  With segments and events the two passes barely overlap:
      // a 16-wave forward workgroup fills a CU's registers, so the relays of segment s+1 wait until the forward launch of
      // segment s has drained (config 3: chain 0.71 ms alone + forward 1.55 ms alone = 2.2 ms together).  Here the relays
      // hold their CUs from the start, the forward workgroups take whatever is free, in time order, and the whole chip
      // once the relays are through.  The forward launch is held back (relay_gate_kernel polls a word every relay workgroup
      // bumps at its start) until the relays are resident: a forward workgroup that waits for a relay which cannot start
      // would be a deadlock -- every wait in the kernels is bounded all the same, and a time-out re-runs the call (forward_device).
      flow = use_chain && opt_relay_flow && opt_segments <= 0 && (fuse ? true : use_rows) && gate_ok();
      if (flow) segments = 1;
      if (flow && started_target > (1u << 30))
      {
        // long before the start counter could wrap: drain both streams and begin it again (once per ~8 million calls)
        if (aux) SDFT_TRY(hipStreamSynchronize(aux));
        SDFT_TRY(hipStreamSynchronize(stream));
        SDFT_TRY(hipMemset(d_started, 0, 8));
        started_target = 0;
      }
      segments = std::max(1L, std::min(segments, chunks));
      if (segments > 1 || flow)
      {
        if (!aux) SDFT_TRY(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        if (!ev_delta) SDFT_TRY(hipEventCreateWithFlags(&ev_delta, hipEventDisableTiming));
        while ((long)seg_events.size() < segments)
        {
          hipEvent_t e; SDFT_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          seg_events.push_back(e);
        }
        for (int q = 0; q < 2; ++q)
          if (!d_run_acc[q].reserve(channels * nb) || !d_run_fid[q].reserve(channels * nb)) return false;
        SDFT_TRY(hipEventRecord(ev_delta, stream));                 // delta (and everything before) done
        SDFT_TRY(hipStreamWaitEvent(aux, ev_delta, 0));
      }
    }
    // the stage's events go where its kernels go (the overlapped exact pass runs on `aux`)
    hipStream_t carry_stream = (segments > 1 || flow) ? aux : stream;
    if (!prof_begin(ST_CARRY, carry_stream)) return false;
    CarryArgs<FD> ca;
    ca.acc_next = nullptr; ca.fid_next = nullptr; ca.chunk0 = 0; ca.launch_chunks = (unsigned)chunks;
    ca.delta = d_delta.p; ca.tw = d_tw.p; ca.wtab = d_wtab.p; ca.carry = d_carry.p; ca.seed = d_seed.p;
    ca.acc_state = acc_p(); ca.fid_state = fid_p(); ca.n = n;
    ca.nbins = (unsigned)nb; ca.chunks = (unsigned)chunks; ca.chunk_len = (unsigned)len; ca.cursor0 = (unsigned)cursor;
    const unsigned bin_blocks = (unsigned)((nb + kBlock - 1) / kBlock);
    if (!grid_fits((size_t)bin_blocks * (size_t)chunks * channels) || !grid_fits(((nb + kScanBins - 1) / kScanBins) * channels) ||
        !grid_fits(((nb + kWave / 2 - 1) / (kWave / 2)) * channels)) return false;
    bool use_seed = true;
    if (chunks == 1)
    {
      // single chunk: the stream state is the carry; delta_kernel has already copied it
    }
    else if (exact)
    {
      const unsigned eblocks = (unsigned)((nb + kWave / 2 - 1) / (kWave / 2));
      for (long sg = 0; sg < segments && use_chain; ++sg)
      {
        const long j0 = chunks * sg / segments, j1 = chunks * (sg + 1) / segments;
        ChainArgs<FD> cc{};
        cc.delta = d_delta.p; cc.tw = d_tw.p; cc.fseed = d_fseed.p; cc.carry = d_carry.p;
        cc.acc_state = sg == 0 ? acc_p() : d_run_acc[(sg - 1) & 1].p;
        cc.acc_next = segments > 1 ? d_run_acc[sg & 1].p : nullptr;
        cc.n = n; cc.nbins = (unsigned)nb; cc.chunks = (unsigned)chunks; cc.chunk_len = (unsigned)len; cc.cursor0 = (unsigned)cursor;
        cc.chunk0 = (unsigned)j0; cc.launch_chunks = (unsigned)(j1 - j0); cc.L = cL; cc.P = 0; cc.chunk_shift = shift; cc.debug = (unsigned)opt_chain_debug & (47u | 128u); cc.stats = nullptr;
        if (opt_chain_debug & (16 | 64 | 128)) { if (!d_chain_stats.reserve(64 + 3 * 1024)) return false; cc.stats = d_chain_stats.p; }
        cc.status = ensure_status() ? d_status : nullptr;
        if (cc.status) status_armed = true;
        cc.ready = nullptr; cc.ready_seq = 0; cc.started = nullptr; cc.chunks_channels = (unsigned)channels;
        if (flow)
        {
          const size_t relays_per_channel = (nb + kWave / 2 - 1) / (kWave / 2);
          const size_t words = channels * (size_t)chunks * relays_per_channel;
          if (d_ready.cap < words)
          {
            if (!d_ready.reserve(words)) return false;
            SDFT_TRY(hipMemsetAsync(d_ready.p, 0, d_ready.cap * sizeof(unsigned), carry_stream));
            ready_seq = 0;
          }
          if (++ready_seq == 0) { SDFT_TRY(hipMemsetAsync(d_ready.p, 0, d_ready.cap * sizeof(unsigned), carry_stream)); ready_seq = 1; }
          cc.ready = d_ready.p; cc.ready_seq = ready_seq; cc.started = d_started;
        }
        const unsigned cblocks = eblocks * (unsigned)channels;
        bool ok = true;
        switch (cL)
        {
          case 128: ok = launch_relay<128>(cc, cblocks, carry_stream); break;
          case 64:  ok = launch_relay<64>(cc, cblocks, carry_stream); break;
          case 32:  ok = launch_relay<32>(cc, cblocks, carry_stream); break;
          case 16:  ok = launch_relay<16>(cc, cblocks, carry_stream); break;
          default:  ok = launch_relay<8>(cc, cblocks, carry_stream); break;
        }
        if (!ok) return false;
        if (segments > 1 || flow) SDFT_TRY(hipEventRecord(seg_events[sg], aux));
        if (flow)
        {
          // the forward launch may go once every relay workgroup of this launch is resident
          const unsigned relays = cblocks, groups = relay_groups_of(cL);
          started_target += (relays + groups - 1) / groups;
          hipLaunchKernelGGL((relay_gate_kernel<FD>), dim3(1), dim3(kWave), 0, stream, (const unsigned*)d_started, started_target);
          SDFT_TRY(hipGetLastError());
        }
      }
      for (long sg = 0; sg < segments && !use_chain; ++sg)
      {
        const long j0 = chunks * sg / segments, j1 = chunks * (sg + 1) / segments;
        CarryArgs<FD> cs = ca;
        cs.chunk0 = (unsigned)j0; cs.launch_chunks = (unsigned)(j1 - j0);
        cs.acc_state = sg == 0 ? acc_p() : d_run_acc[(sg - 1) & 1].p;
        cs.fid_state = sg == 0 ? fid_p() : d_run_fid[(sg - 1) & 1].p;
        cs.acc_next = d_run_acc[sg & 1].p; cs.fid_next = d_run_fid[sg & 1].p;
        hipLaunchKernelGGL((carry_exact_kernel<FD>), dim3(eblocks * (unsigned)channels), dim3(kWave), 0,
                           segments > 1 ? aux : stream, cs);
        SDFT_TRY(hipGetLastError());
        if (segments > 1) SDFT_TRY(hipEventRecord(seg_events[sg], aux));
      }
    }
    else
    {
      // partial sums per chunk (form chosen above)
      const size_t span_bytes = span * sizeof(fdx);
      const unsigned sum_chunks = (unsigned)(chunks - (delta_in_carry ? 0 : 1));
      if (sums_form == SUMS_FFT2)
      {
        unsigned lg = 0; while (((size_t)1 << lg) < span) ++lg;
        hipLaunchKernelGGL((chunk_fft_kernel<TD, FD>), dim3((unsigned)(sum_chunks * channels)), dim3(kBlock), span_bytes, stream, ca, lg, din);
      }
      else if (sums_form == SUMS_FFT_MIXED)
        hipLaunchKernelGGL((chunk_fft_mixed_kernel<TD, FD>), dim3((unsigned)(sum_chunks * channels)), dim3(kBlock), 2 * span_bytes,
                           stream, ca, (unsigned)span, rl, din);
      else
      {
        if (!grid_fits((size_t)bin_blocks * sum_chunks * channels)) return false;
        hipLaunchKernelGGL((chunk_sum_kernel<TD, FD>), dim3((unsigned)((size_t)bin_blocks * sum_chunks * channels)), dim3(kBlock), 0, stream, ca, din);
      }
      SDFT_TRY(hipGetLastError());
      hipLaunchKernelGGL((carry_scan_kernel<FD>), dim3((unsigned)(((nb + kScanBins - 1) / kScanBins) * channels)),
                         dim3(kScanBins * kScanSlices), 0, stream, ca);
      SDFT_TRY(hipGetLastError());
      use_seed = false;
    }
    if (!prof_end(ST_CARRY, carry_stream)) return false;

    // K1
    if (!prof_begin(ST_FORWARD)) return false;
    ForwardArgs<FD> fa{};
    fa.delta = d_delta.p; fa.tw = d_tw.p; fa.wtab = d_wtab.p; fa.carry = d_carry.p;
    fa.seed = (use_seed && !use_chain) ? d_seed.p : nullptr;
    fa.fseed = use_chain ? d_fseed.p : nullptr; fa.fseed_L = use_chain ? cL : 0;
    fa.out = out; fa.out_stride = out_stride; fa.out_rows = rows;
    // the new state goes to the other buffer set (the one a call started from survives it, see forward_device)
    fa.acc_state = d_accs[st_cur ^ 1].p; fa.fid_state = d_fids[st_cur ^ 1].p; fa.n = n;
    fa.total_waves = (unsigned long long)channels * (unsigned long long)chunks * (unsigned long long)ntiles;
    fa.nbins = (unsigned)nb; fa.chunks = (unsigned)chunks; fa.chunk_len = (unsigned)len; fa.tiles = (unsigned)ntiles;
    fa.interior_lanes = (unsigned)inter; fa.cursor0 = (unsigned)cursor; fa.chunk_shift = shift;
    fa.vec_store = (bins_per_lane() == 2 && (nb % 2 == 0) && ((uintptr_t)out % 16 == 0) && (out_stride % 2 == 0) && !rows) ? 1 : 0;
    fa.wscale = (window == WIN_HANN) ? (FD)(tab.aweight * (FD)(0.25)) : tab.aweight;   // :371
    fa.done.flag = nullptr; fa.done.count = nullptr; fa.done.seq = 0; fa.done.total = 0;
    fa.ready = nullptr; fa.ready_seq = 0; fa.ready_n = 0; fa.ready_channels = (unsigned)channels; fa.ready_status = nullptr; fa.ready_status_seen = 0;
    fa.inv_channels = inv32((unsigned)channels);
    if (flow)
    {
      fa.ready = d_ready.p; fa.ready_seq = ready_seq; fa.ready_n = (unsigned)((nb + kWave / 2 - 1) / (kWave / 2));
      fa.ready_status = ensure_status() ? d_status : nullptr;
      fa.ready_status_seen = fa.ready_status ? *(volatile unsigned*)h_status : 0u;
      if (fa.ready_status) status_armed = true;
    }
    // short synchronous calls: the row-group kernels report their own completion (a word in pinned host memory
    // reaches the host before the stream does).  Worth it while the kernel has little to write back: n = 4096,
    // N = 1024: 49.7 -> 46.7 us per sdft_sdft_n, 40.4 -> 35.8 us per fused call; nothing at n = 48000.
    if ((use_rows || fuse) && segments == 1 && channels * n * nb <= (fuse ? (size_t)1 << 26 : kFlagMax))
      fa.done = arm_flag((unsigned)(channels * (size_t)chunks));
    last_segments = segments;
    for (long sg = 0; sg < segments; ++sg)
    {
      const long j0 = chunks * sg / segments, j1 = chunks * (sg + 1) / segments;
      fa.chunk0 = (unsigned)j0; fa.launch_chunks = (unsigned)(j1 - j0); fa.inv_chunks = inv32(fa.launch_chunks);
      fa.xcd_map = (opt_xcd_map && !flow && channels * (size_t)(j1 - j0) >= 16) ? (unsigned)(channels * (size_t)(j1 - j0)) : 0u;
      fa.total_waves = (unsigned long long)channels * (unsigned long long)(j1 - j0) * (unsigned long long)ntiles;
      if (segments > 1) SDFT_TRY(hipStreamWaitEvent(stream, seg_events[sg], 0));      // (flow mode: the kernel waits chunk by chunk)
      const unsigned long long blocks = (fa.total_waves + kWavesPerBlock - 1) / kWavesPerBlock;
      // fused arithmetic only where the result is not claimed bit-identical: FD double with carries
      // from the chunk-parallel pass (use_seed == false <=> fast mode, more than one chunk)
      const bool fused = (use_rows || fuse) && opt_fused && sizeof(FD) == 8 && !use_seed;
      last_fused = fused;
      if (fuse)
      {
        // rows never leave the workgroup: synthesis in the same launch (caller checked fuse_ok())
        const bool exact_order = wants_reference_order();
        last_fused_exact = exact_order;
        const bool folded = !exact_order && !fuse->store && opt_fold && coeff_ready;
        last_fused_fold = folded;
        if (folded)
        {
          if (!launch_process(fa, *fuse, (unsigned)(channels * (size_t)(j1 - j0)), fused)) return false;
        }
        else if (!use_rows) { set_error("sdft_hip_process_n", "rows of this length are fused in the folded form only"); return false; }
        else if (!launch_syn(fa, *fuse, (unsigned)(channels * (size_t)(j1 - j0)), (unsigned)(row_waves() * kWave), fused && !exact_order, exact_order))
          return false;
      }
      else if (use_rows) launch_forward_rows(fa, (unsigned)(channels * (size_t)(j1 - j0)), (unsigned)(row_waves() * kWave), fused);
      else launch_forward(fa, (unsigned)blocks);
      SDFT_TRY(hipGetLastError());
    }
    SDFT_TRY(hipGetLastError());
    if (flow) SDFT_TRY(hipStreamWaitEvent(stream, seg_events[0], 0));          // the call ends when both launches have
    last_flow = flow;
    if (!prof_end(ST_FORWARD)) return false;

    // fid stays on the canonical rotation sequence unless this call seeded chunks from the closed-form
    // table; a call that crosses the roll-over with serial fid arithmetic puts it back
    if (!use_seed) fid_canonical = false;
    else if (cursor + n >= span) fid_canonical = true;
    cursor = (cursor + n) % span;
    st_cur ^= 1;
    return true;
  }

  static RadixList smooth_radices(size_t span)
  {
    const logic::Radices r = logic::smooth_radices(span);
    RadixList rl; rl.count = r.count;
    for (int i = 0; i < 15; ++i) rl.r[i] = r.r[i];
    return rl;
  }
  size_t self_cells() const { return logic::self_cells(nbins, opt_self >= 1, sizeof(fdx)); }
  // what the self-carried form needs of the plan and the call (the kernel that has it is chosen by the caller)
  // (any_length: pipelined calls take the form whatever the length -- one stream runs long calls faster with the pre-pass,
  // n = 1e6: 77.3 against 75.5 % of peak, but two matrices in turn, pipelined: 82.4 %)
  bool self_eligible(size_t n, bool fused_call, bool any_length = false) const
  {
    const size_t self_max = fused_call ? std::min<size_t>(kSelfMax, (size_t)1 << 16) : kSelfMax;
    return sizeof(FD) == 8 && carry_mode != CARRY_EXACT && opt_self && self_cells() != 0 && (n <= self_max || any_length);
  }

  // ---- pipelined calls ------------------------------------------------------------------------------------------
  // Asynchronous analysis calls on the plan's own stream: consecutive calls' row kernels do not wait for each other.  What
  // ties call k + 1 to call k is the plan's state; self_state_kernel computes it from the call's samples ahead of the rows
  // (main stream: a chain of small kernels), the rows go to two streams in turn, each waiting for the state it reads only.
  // The next call's workgroups take the CUs as the previous call's leave them: launch gap, prologue (fold + FFT, no HBM
  // traffic) and the ragged end of a call are covered by the neighbour's stores.  State slots form a ring of four: the
  // state kernel of call k + 3 overwrites what the rows of call k read, and waits for them.
  // Anything else that touches the plan joins first (the main stream waits for the outstanding rows).  Off once the host has
  // asked for the stream (sdft_hip_get_stream: it may queue work of its own behind a call), with profiling, on a caller's stream.
  long opt_pipeline = 1;
  long pipe_stream_kind = 0, pipe_stream_attempts = 0;        // 1 = ordinary streams, 2 = by priority (0: none found, one stream)
  std::vector<hipStream_t> spare_streams;
  bool stream_exposed = false;
  hipStream_t row_streams[2] = {nullptr, nullptr};
  hipEvent_t ev_pre = nullptr, ev_rows[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long pipe_calls = 0, pipe_ordered = 0;
  // what the outstanding row launches write and which row stream each went to: a call whose matrix or samples overlap one of
  // them is ordered behind it, as one stream would have it (logic::RowRing decides, this class waits and records)
  logic::RowRing ring;
  logic::Range prev_out;                                     // the matrix of the previous analysis call (dense, row-group kernel)
  bool pipe_this = false;                                    // forward_launch: this call is pipelined
  // Only calls whose every pointer is the caller's DEVICE memory may leave the plan's stream: the host-pointer routes reuse
  // the plan's staging buffers (d_stage_*, d_io, d_pin) on the main stream right behind a launch and promise the outputs
  // complete on return.  Set by the device/device branches of sdft_n / isdft_n for the duration of the call.
  bool pipe_allowed = false;
  bool ensure_pipe()
  {
    if (ev_pre) return true;
    // The two row streams must not share a hardware queue with each other or with the main stream: kernels of one queue
    // run one after the other, and the waits between the streams then only cost (the bench process measured 180 us per
    // pipelined call at n = 48 000 where a fresh process measured 128-140; the runtime deals its few hardware queues out
    // by its own rules).  So the streams are tried: a 200 us wave on each of the three at once (queue_probe_kernel), timed
    // by events -- they must have run at the same time.  Up to three pairs of ordinary streams, then one stream above and
    // one below the plan's priority (separate queue pools; 3-4 % slower than a good ordinary pair, the lower-priority
    // launch falls behind), else the plan stays on one stream.  Once per plan, a quarter of a millisecond per pair tried.
    {
      // (nothing in here is an error of the call: a plan that cannot have its streams stays on one stream)
      hipEvent_t ev[6] = {};
      auto release_events = [&]() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); };
      auto give_up = [&]() { (void)hipGetLastError(); release_events(); opt_pipeline = 0; pipe_stream_kind = 0; return false; };
      if (hipStreamSynchronize(stream) != hipSuccess) return give_up();
      for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) { e = nullptr; return give_up(); }
      auto concurrent = [&](hipStream_t a, hipStream_t b) -> int {
        hipStream_t st[3] = {stream, a, b};
        for (int i = 0; i < 3; ++i)
        {
          if (hipEventRecord(ev[2 * i], st[i]) != hipSuccess) return -1;
          hipLaunchKernelGGL((queue_probe_kernel<FD>), dim3(1), dim3(kWave), 0, st[i], 20000ull);
          if (hipEventRecord(ev[2 * i + 1], st[i]) != hipSuccess) return -1;
        }
        for (int i = 0; i < 3; ++i) if (hipStreamSynchronize(st[i]) != hipSuccess) return -1;
        // three waves of 200 us: from any start to any end less than two of them
        float span_ms = 0.f, worst = 0.f;
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j)
          {
            if (hipEventElapsedTime(&span_ms, ev[2 * i], ev[2 * j + 1]) != hipSuccess) return -1;
            worst = std::max(worst, span_ms);
          }
        return worst < 0.35f ? 1 : 0;
      };
      bool found = false;
      int lo = 0, hi = 0;                                     // (numerically: greatest = lowest priority)
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
      for (int attempt = 0; attempt < 4 && !found; ++attempt)
      {
        hipStream_t cand[2] = {nullptr, nullptr};
        const bool by_priority = attempt == 3;
        if (by_priority && lo == hi) break;
        for (int i = 0; i < 2; ++i)
        {
          // (blocking streams, like the plan's own: a host that reads its results with a plain hipMemcpy relies on the null
          // stream waiting for the plan's work, and that has to include the rows)
          const hipError_t e = by_priority ? hipStreamCreateWithPriority(&cand[i], hipStreamDefault, i == 0 ? hi : lo)
                                           : hipStreamCreateWithFlags(&cand[i], hipStreamDefault);
          if (e != hipSuccess) { if (cand[0]) (void)hipStreamDestroy(cand[0]); for (hipStream_t sp : spare_streams) (void)hipStreamDestroy(sp); spare_streams.clear(); return give_up(); }
        }
        const int ok = concurrent(cand[0], cand[1]);
        if (ok == 1) { row_streams[0] = cand[0]; row_streams[1] = cand[1]; found = true; pipe_stream_kind = by_priority ? 2 : 1; pipe_stream_attempts = attempt + 1; }
        else
        {
          (void)hipGetLastError();
          // (kept alive until the search is over: a destroyed stream's queue would be handed to the next candidate)
          spare_streams.push_back(cand[0]); spare_streams.push_back(cand[1]);
        }
      }
      for (hipStream_t sp : spare_streams) (void)hipStreamDestroy(sp);
      spare_streams.clear();
      if (!found) return give_up();
      release_events();
    }
    bool ok = hipEventCreateWithFlags(&ev_pre, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreateWithFlags(&ev_rows[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&ev_inv[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { release_pipe(); opt_pipeline = 0; pipe_stream_kind = 0; return false; }
    return true;
  }
  // Synthesis calls are stateless: consecutive asynchronous ones (a host that synthesises matrix after matrix) go to the two
  // row streams in turn as well -- n = 48 000: 148 -> 122 us per call, 66 -> 80 % of peak; n = 262 144: 82 -> 86 %
  // (two plans in turn, scripts/pair_overlap_probe.py ...) -- but never beside an analysis: a synthesis waits for the
  // rows of the analyses before it, an analysis for the syntheses before it (reads mixed into the write stream cost HBM
  // more than the overlap gains: 1.62 against 1.47 ms per pair at n = 262 144).
  // Only for hosts that do synthesise call after call: a host that alternates analysis and synthesis (the reference's loop)
  // would pay an event wait between streams per call and gain nothing, so the mode is learnt from the calls themselves --
  // on when a synthesis follows a synthesis, off when an analysis follows a lone synthesis.
  logic::CallPattern calls;                                   // ... learnt from the calls (the same for analyses)
  hipEvent_t ev_inv[2] = {nullptr, nullptr};                 // the last synthesis launch on each row stream
  logic::InverseStreams inv;
  unsigned long long inv_calls = 0;
  long last_inverse_pipelined = 0;
  bool pipe_join_inverses()
  {
    for (int i = 0; i < 2; ++i)
      if (inv.used[i]) { SDFT_TRY(hipStreamWaitEvent(stream, ev_inv[i], 0)); inv.used[i] = false; }
    return true;
  }
  bool pipe_join() { return pipe_join_rows() && pipe_join_inverses(); }
  bool pipe_join_rows()
  {
    if (!ring.open) return true;
    // the last launch on each row stream (among the four the ring remembers; older ones are ordered before them)
    int slots[2];
    const int count = ring.last_per_stream(slots);
    for (int i = 0; i < count; ++i) SDFT_TRY(hipStreamWaitEvent(stream, ev_rows[slots[i]], 0));
    ring.joined();
    return true;
  }
  void release_pipe()
  {
    for (int i = 0; i < 2; ++i) if (row_streams[i]) { (void)hipStreamSynchronize(row_streams[i]); (void)hipStreamDestroy(row_streams[i]); row_streams[i] = nullptr; }
    if (ev_pre) { (void)hipEventDestroy(ev_pre); ev_pre = nullptr; }
    for (int i = 0; i < 4; ++i) if (ev_rows[i]) { (void)hipEventDestroy(ev_rows[i]); ev_rows[i] = nullptr; }
    for (int i = 0; i < 2; ++i) { if (ev_inv[i]) { (void)hipEventDestroy(ev_inv[i]); ev_inv[i] = nullptr; } inv.used[i] = false; }
    ring.joined();
    (void)hipGetLastError();
  }
  bool pipe_wanted(const void* fuse) const
  {
    return !fuse && pipe_allowed && async && own_stream && !stream_exposed && opt_pipeline && profile == 0 && sizeof(FD) == 8;
  }
  bool launch_self_state(const ForwardArgs<FD>& fa, const SelfArgs<TD, FD>& sa, unsigned threads)
  {
    auto kern = self_state_kernel<TD, FD>;
    const size_t lds = self_cells() * sizeof(fdx);
    static thread_local int raised_on = -1;
    if (raised_on != device)
    {
      SDFT_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024)));
      raised_on = device;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)channels), dim3(threads), lds, stream, fa, sa);
    SDFT_TRY(hipGetLastError());
    return true;
  }

  // ---- self-carried chunks: the whole chunk-parallel call in one launch (SelfArgs in sdft_kernels.hpp) ----
  bool forward_self(size_t n, const TD* x, size_t x_stride, fdx* out, size_t out_stride, long chunks, long len,
                    const FuseArgs<TD, FD>* fuse)
  {
    const size_t nb = nbins, span = 2 * nbins;
    SelfArgs<TD, FD> sa;
    sa.x = x; sa.x_stride = x_stride;
    sa.hist_in = d_hist[hist_cur].p; sa.hist_out = d_hist[hist_cur ^ 1].p;
    sa.acc_in = d_accs[st_cur].p;
    sa.log2m = 0; sa.rl.count = 0;
    if ((span & (span - 1)) == 0) while (((size_t)1 << sa.log2m) < span) ++sa.log2m;
    else sa.rl = smooth_radices(span);
    sa.lds_deltas = 0;
#ifdef SDFT_SELF_STAMPS
    sa.stamps = reinterpret_cast<unsigned long long*>(opt_self_stamps);
#endif
    ForwardArgs<FD> fa{};
    fa.delta = nullptr; fa.tw = d_tw.p; fa.wtab = d_wtab.p; fa.carry = nullptr; fa.seed = nullptr; fa.fseed = nullptr; fa.fseed_L = 0;
    fa.out = out; fa.out_stride = out_stride; fa.out_rows = nullptr;
    fa.acc_state = d_accs[st_cur ^ 1].p; fa.fid_state = d_fids[st_cur ^ 1].p; fa.n = n;
    fa.total_waves = 0;
    fa.nbins = (unsigned)nb; fa.chunks = (unsigned)chunks; fa.chunk_len = (unsigned)len; fa.tiles = (unsigned)tiles();
    fa.interior_lanes = (unsigned)interior_lanes(); fa.cursor0 = (unsigned)cursor; fa.chunk_shift = 0;
    fa.chunk0 = 0; fa.launch_chunks = (unsigned)chunks; fa.inv_chunks = inv32(fa.launch_chunks);
    fa.xcd_map = (opt_xcd_map && channels * (size_t)chunks >= 16) ? (unsigned)(channels * (size_t)chunks) : 0u;
    fa.vec_store = 0;
    fa.wscale = (window == WIN_HANN) ? (FD)(tab.aweight * (FD)(0.25)) : tab.aweight;   // :371
    fa.done.flag = nullptr; fa.done.count = nullptr; fa.done.seq = 0; fa.done.total = 0;
    if (!grid_fits(channels * (size_t)chunks)) return false;
    if (channels * n * nb <= (fuse ? (size_t)1 << 26 : kFlagMax)) fa.done = arm_flag((unsigned)(channels * (size_t)chunks));
    const unsigned blocks = (unsigned)(channels * (size_t)chunks);
    const bool fused = opt_fused != 0;
    last_fused = fused; last_segments = 1; last_chain = 0;
    last_pipelined = 0;
    if constexpr (sizeof(FD) == 8)
    {
      // A host that writes call after call into ONE matrix gains nothing from two streams and would pay for the order
      // between them (an event wait across streams, satisfied or not, costs the row kernels 5-15 us per call: n = 48 000,
      // 165-171 against 157 us): a call whose matrix overlaps the previous call's takes the one-stream form.  A host that
      // alternates between two matrices is pipelined, each matrix on its own stream.
      const uintptr_t olo = reinterpret_cast<uintptr_t>(out), ohi = olo + ((channels - 1) * out_stride + n * nb) * sizeof(fdx);
      if (pipe_this && !fuse && ensure_pipe())
      {
        if (!pipe_join_inverses()) return false;             // never beside a synthesis
        const uintptr_t xlo = reinterpret_cast<uintptr_t>(x), xhi = xlo + ((channels - 1) * x_stride + n) * sizeof(TD);
        // samples that an outstanding launch is still writing (a matrix reinterpreted as samples): no overlap for this call
        if (ring.samples_overlap(xlo, xhi) && !pipe_join()) return false;
        const int s1 = (st_cur + 1) & 3, h1 = (hist_cur + 1) & 3;
        // Which row stream: the other one than the previous launch's -- unless this call's matrix overlaps what an
        // outstanding launch writes (a host that reuses one matrix): then the stream of the latest such launch, whose order
        // costs nothing (an event wait across streams costs ~15 us per call: n = 48 000 into one matrix 171 against 157 us)
        const logic::RowRing::Pick pk = ring.pick(olo, ohi);
        const int rsi = pk.stream;
        hipStream_t rs = row_streams[rsi];
        // the rows read the state everything queued on the main stream so far leaves behind
        SDFT_TRY(hipEventRecord(ev_pre, stream));
        SDFT_TRY(hipStreamWaitEvent(rs, ev_pre, 0));
        if (pk.behind)
        {
          ++pipe_ordered;
          if (pk.wait_launch >= 0) SDFT_TRY(hipStreamWaitEvent(rs, ev_rows[pk.wait_launch], 0));   // (overlapping launches on the other stream as well: the latest of them)
        }
        // the slot the state kernel writes was read by the rows of three calls ago
        if (ring.state_reader() >= 0) SDFT_TRY(hipStreamWaitEvent(stream, ev_rows[ring.state_reader()], 0));
        ForwardArgs<FD> fs = fa; SelfArgs<TD, FD> ss = sa;
        fs.acc_state = d_accs[s1].p; fs.fid_state = d_fids[s1].p; ss.hist_out = d_hist[h1].p;
        fs.done.flag = nullptr; fs.done.count = nullptr; fs.done.seq = 0; fs.done.total = 0;
        if (!launch_self_state(fs, ss, (unsigned)(row_waves() * kWave))) return false;
        ForwardArgs<FD> fr = fa; SelfArgs<TD, FD> sr = sa;
        fr.acc_state = nullptr; fr.fid_state = nullptr; sr.hist_out = nullptr;
        fr.done.flag = nullptr; fr.done.count = nullptr; fr.done.seq = 0; fr.done.total = 0;
        hipStream_t main_stream = stream;
        stream = rs;
        const bool ok = launch_forward_rows_self(fr, sr, blocks, (unsigned)(row_waves() * kWave), fused);
        stream = main_stream;
        if (!ok) return false;
        SDFT_TRY(hipEventRecord(ev_rows[ring.slot()], rs));
        ring.launched(olo, ohi, rsi);
        ++pipe_calls;
        last_pipelined = 1;
        st_cur = s1; hist_cur = h1;
        fid_canonical = false;                               // the rotation comes from the closed-form table (as the one-stream self form below)
        cursor = (cursor + n) % span;
        return true;
      }
    }
    if (!pipe_join()) return false;
    if (!prof_begin(ST_FORWARD)) return false;
    if constexpr (sizeof(FD) == 8)
    {
      if (fuse)
      {
        last_fused_exact = 0; last_fused_fold = 1;
        if (!launch_process(fa, *fuse, blocks, fused, &sa)) return false;
      }
      else if (!launch_forward_rows_self(fa, sa, blocks, (unsigned)(row_waves() * kWave), fused)) return false;
    }
    SDFT_TRY(hipGetLastError());
    if (!prof_end(ST_FORWARD)) return false;
    hist_cur ^= 1; st_cur ^= 1;
    fid_canonical = false;                                   // chunks seeded fid from the closed-form table
    cursor = (cursor + n) % span;
    return true;
  }
  // ---- single-chunk calls (hop-wise streaming): one fused launch, tiles spread over the CUs ----
  bool forward_hop(size_t n, const TD* x, size_t x_stride, fdx* out, size_t out_stride, fdx* const* rows)
  {
    const size_t nb = nbins, span = 2 * nbins;
    const long ntiles = tiles(), inter = interior_lanes();
    HopArgs<TD, FD> ha;
    ha.x = x; ha.x_stride = x_stride;
    ha.hist_in = d_hist[hist_cur].p; ha.hist_out = d_hist[hist_cur ^ 1].p;
    ha.tw = d_tw.p;
    ha.acc_in = d_accs[st_cur].p; ha.fid_in = d_fids[st_cur].p;
    ha.acc_out = d_accs[st_cur ^ 1].p; ha.fid_out = d_fids[st_cur ^ 1].p;
    ha.out = out; ha.out_stride = out_stride; ha.out_rows = rows; ha.n = n;
    ha.total_waves = (unsigned long long)channels * (unsigned long long)ntiles;
    ha.nbins = (unsigned)nb; ha.tiles = (unsigned)ntiles; ha.interior_lanes = (unsigned)inter; ha.cursor0 = (unsigned)cursor;
    ha.vec_store = (bins_per_lane() == 2 && (nb % 2 == 0) && ((uintptr_t)out % 16 == 0) && (out_stride % 2 == 0) && !rows) ? 1 : 0;
    ha.wscale = (window == WIN_HANN) ? (FD)(tab.aweight * (FD)(0.25)) : tab.aweight;   // :371
    // one wave per workgroup while the launch is small (the tiles of a row land on different CUs);
    // four per workgroup once there are more waves than SIMDs anyway
    const bool wide = ha.total_waves > 2048;
    const unsigned long long blocks = wide ? (ha.total_waves + 3) / 4 : ha.total_waves;
    if (!grid_fits(blocks)) return false;
    // small launches of hop-sized calls: two waves per tile (recurrence | window + stores)
    const bool pipe = !wide && n <= (size_t)kHopMax;       // small launches: two waves per tile (forward_hop2_kernel)
    last_hop_pipe = pipe;
    // ... and the call's samples in time parts, every (tile, part) a workgroup on a CU of its own (forward_hop2_kernel): as many
    // parts as leave every workgroup a CU, at most 8, at least 12 samples each (the recurrence wave of a part runs the state
    // through the samples before it: what a part saves is the window, the demodulation and the stores of those samples)
    ha.parts = 1; ha.part_len = (unsigned)n;
    if (pipe)
    {
      const logic::HopParts hp = logic::hop_parts(n, ha.total_waves, compute_units, opt_hop_parts, flag_wanted && !async);
      ha.parts = hp.parts; ha.part_len = hp.part_len;
    }
    last_hop_parts = ha.parts;
    const unsigned hop2_blocks = (unsigned)blocks * ha.parts;
    ha.done = arm_flag(wide ? 0u : (pipe ? hop2_blocks : (unsigned)blocks));     // every workgroup reports
    ha.stamps = nullptr;
#ifdef SDFT_HOP_STAMPS
    if (d_partial.reserve(64)) { ha.stamps = reinterpret_cast<unsigned long long*>(d_partial.p); last_partial_elems = 0; hop2_stamps = true; }
#endif
    if (!prof_begin(ST_FORWARD)) return false;
    if (pipe) { if (rows) launch_hop2_t<true>(ha, hop2_blocks); else launch_hop2_t<false>(ha, hop2_blocks); }
    else if (rows) { if (wide) launch_hop_t<true, 4>(ha, (unsigned)blocks); else launch_hop_t<true, 1>(ha, (unsigned)blocks); }
    else      { if (wide) launch_hop_t<false, 4>(ha, (unsigned)blocks); else launch_hop_t<false, 1>(ha, (unsigned)blocks); }
    SDFT_TRY(hipGetLastError());
    if (!prof_end(ST_FORWARD)) return false;
    hist_cur ^= 1; st_cur ^= 1;
    last_kernel = 3;
    if (cursor + n >= span) fid_canonical = true;
    cursor = (cursor + n) % span;
    return true;
  }

  // fused call, one time chunk, folded form: one launch (process_hop2_kernel); the caller has folded the coefficients
  DevBuf<double> d_partial;
  DevBuf<unsigned> d_tickets;
  // completion word in pinned host memory: the kernels of short synchronous calls set it (signal_done), finish()
  // polls it.  flag_wanted: the public entry point has device pointers on both sides and will end in finish();
  // flag_pending: the call's LAST launch was armed (every entry point starts with both false, see bind()).
  unsigned* h_done_flag = nullptr;
  unsigned* d_done_flag = nullptr;
  DevBuf<unsigned> d_done_count;
  // fused call in the reference's order with float samples: samples whose sum was walked bin by bin (the others were
  // proven by the rounding interval of the tree sum: forward_rows_kernel<SYN = 2>); get_option "ordered_walks"
  DevBuf<unsigned> d_walked;
  unsigned* walked_counter()
  {
    if (!d_walked.p)
    {
      if (!d_walked.reserve(1)) { (void)hipGetLastError(); return nullptr; }
      if (hipMemsetAsync(d_walked.p, 0, sizeof(unsigned), stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    return d_walked.p;
  }
  long ordered_walks() const
  {
    if (!d_walked.p) return 0;
    unsigned v = 0;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(&v, d_walked.p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (long)v;
  }
  unsigned flag_seq = 0;
  bool flag_pending = false, flag_wanted = false;
  bool ensure_flag()
  {
    if (h_done_flag) return true;
    if (!d_done_count.reserve(2)) return false;
    if (hipMemsetAsync(d_done_count.p, 0, 2 * sizeof(unsigned), stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipHostMalloc((void**)&h_done_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); h_done_flag = nullptr; return false; }
    *h_done_flag = 0;
    if (hipHostGetDevicePointer((void**)&d_done_flag, h_done_flag, 0) != hipSuccess)
    {
      (void)hipGetLastError(); (void)hipHostFree(h_done_flag); h_done_flag = nullptr; return false;
    }
    return true;
  }
  DoneSignal arm_flag(unsigned total)
  {
    DoneSignal d; d.flag = nullptr; d.count = nullptr; d.seq = 0; d.total = total;
    flag_pending = false;
    if (flag_wanted && !async && !profile && opt_spin && total > 0 && ensure_flag())
    {
      d.flag = d_done_flag; d.count = d_done_count.p; d.seq = ++flag_seq;
      flag_pending = true;
    }
    return d;
  }
  bool process_hop(size_t n, const TD* x, size_t x_stride, TD* y, size_t y_stride)
  {
    const size_t nb = nbins, span = 2 * nbins;
    const size_t ptiles = (nb + kWave - 1) / kWave;
    if (!grid_fits(channels * ptiles)) return false;
    if (!d_partial.reserve(channels * ptiles * n + 8)) return false;     // + stamps of the development build
    last_partial_elems = channels * ptiles * n;
    if (d_tickets.cap < channels)
    {
      if (!d_tickets.reserve(channels)) return false;
      SDFT_TRY(hipMemsetAsync(d_tickets.p, 0, channels * sizeof(unsigned), stream));
    }
    ProcHopArgs<TD, FD> pa;
    pa.x = x; pa.x_stride = x_stride; pa.y = y; pa.y_stride = y_stride;
    pa.hist_in = d_hist[hist_cur].p; pa.hist_out = d_hist[hist_cur ^ 1].p;
    pa.tw = d_tw.p;
    pa.acc_in = d_accs[st_cur].p; pa.fid_in = d_fids[st_cur].p;
    pa.acc_out = d_accs[st_cur ^ 1].p; pa.fid_out = d_fids[st_cur ^ 1].p;
    pa.alpha = d_alpha.p; pa.beta = d_beta.p; pa.partial = d_partial.p; pa.tickets = d_tickets.p;
    pa.n = n; pa.nbins = (unsigned)nb; pa.tiles = (unsigned)ptiles; pa.cursor0 = (unsigned)cursor; pa.sweight = tab.sweight;
    pa.done = arm_flag((unsigned)channels);                  // every channel's last workgroup reports
    if (!prof_begin(ST_FORWARD)) return false;
    // two waves per tile (recurrence | coefficients + sums)
    if (!coeff_has_beta) hipLaunchKernelGGL((process_hop2_kernel<TD, FD, false>), dim3((unsigned)(channels * ptiles)), dim3(2 * kWave), 0, stream, pa);
    else                 hipLaunchKernelGGL((process_hop2_kernel<TD, FD, true>), dim3((unsigned)(channels * ptiles)), dim3(2 * kWave), 0, stream, pa);
    last_hop_pipe = 1;
    SDFT_TRY(hipGetLastError());
    if (!prof_end(ST_FORWARD)) return false;
    hist_cur ^= 1; st_cur ^= 1;
    last_kernel = 3; last_chunks = 1; last_chunk_len = (long)n; last_segments = 1; last_fused = 0;
    last_fused_exact = 0; last_fused_fold = 1;
    if (cursor + n >= span) fid_canonical = true;
    cursor = (cursor + n) % span;
    return true;
  }

  // folded form of the fused call (process_rows_kernel): per-bin coefficients from the plan's window and
  // synthesis tables and the call's operation, then one launch per overlap segment
  DevBuf<double> d_alpha, d_beta;
  bool coeff_ready = false, coeff_has_beta = false;
  unsigned coeff_rows = 1;
  int coeff_kind = -1; long coeff_shift = 0;
  bool fold_coefficients(const SpectralOp<FD>& op)
  {
    coeff_ready = false;
    if (!opt_fold || nbins < 8) return true;
    // identity and shift depend on the plan only: folded once; a gain array may change between calls
    if (!op_is_linear<FD>(op.kind)) return true;                                   // gate, power: the windowed rows are needed
    const bool has_array = op.kind == OP_GAIN || op.kind == OP_CGAIN;
    const unsigned rows = has_array && op.rows > 1 ? op.rows : 1u;
    if (rows > 65535u) return true;                                                // (grid.y) -- more gain vectors than that: two passes
    coeff_has_beta = !(latency == 1) || op.kind == OP_CGAIN;                     // im X enters through the synthesis twiddle or a complex gain
    coeff_rows = rows;
    if (!has_array && coeff_kind == op.kind && coeff_shift == op.shift && d_alpha.p) { coeff_ready = true; return true; }
    if (!d_alpha.reserve((size_t)rows * nbins) || !d_beta.reserve((size_t)rows * nbins)) return false;
    const FD w = (window == WIN_HANN) ? (FD)(tab.aweight * (FD)(0.25)) : tab.aweight;   // as ForwardArgs::wscale
    FD h0 = w, h1 = (FD)0, h2 = (FD)0;                                                   // taps of window_tap()
    if (window == WIN_HANN) { h0 = w + w; h1 = -w; }
    else if (window == WIN_HAMMING) { h0 = (FD)(0.54) * w; h1 = -((FD)(0.23) * w); }
    else if (window == WIN_BLACKMAN) { h0 = (FD)(0.42) * w; h1 = -((FD)(0.25) * w); h2 = (FD)(0.04) * w; }
    hipLaunchKernelGGL((fold_coeff_kernel<FD>), dim3((unsigned)((nbins + kBlock - 1) / kBlock), rows), dim3(kBlock), 0, stream,
                       d_alpha.p, d_beta.p, op, (const fdx*)d_syn.p, (unsigned)nbins, latency == 1 ? 1 : 0, h0, h1, h2);
    SDFT_TRY(hipGetLastError());
    coeff_kind = op.kind; coeff_shift = op.shift;
    coeff_ready = true;
    return true;
  }
  // ---- the kernel launches: run-time values -> template instantiations -------------------------------------------------
#include "sdft_plan_launch.inc"
#include "sdft_plan_resident.inc"

  // ---- inverse on device-resident buffers --------------------------------------------------
  bool inverse_device(size_t n, const fdx* in, size_t in_stride, const fdx* const* rows, TD* y, size_t y_stride,
                      const SpectralOp<FD>* op = nullptr)
  {
    if (n == 0) return true;
    SDFT_TRY(hipSetDevice(device));
    const bool ops_wanted = op && op->kind != OP_IDENTITY;
    last_inverse_pipelined = 0;
    // (whatever FD is: the synthesis has no state to carry from call to call)
    inv_after_write = calls.prev_was_analysis;
    calls.on_synthesis_begin();
    const bool inv_pipe = calls.inverse_batch && pipe_allowed && !rows && !ops_wanted && async && own_stream && !stream_exposed && opt_pipeline && profile == 0 &&
                          channels * n * nbins >= ((size_t)6 << 20) && ensure_pipe();
    if (!(inv_pipe ? pipe_join_rows() : pipe_join())) return false;
    hipStream_t main_stream = stream;
    int si = 0;
    uintptr_t ylo = 0, yhi = 0;
    if (inv_pipe)
    {
      // the other stream than the previous synthesis -- the same one if the two write overlapping samples
      ylo = reinterpret_cast<uintptr_t>(y); yhi = ylo + ((channels - 1) * y_stride + n) * sizeof(TD);
      const uintptr_t ilo = reinterpret_cast<uintptr_t>(in), ihi = ilo + ((channels - 1) * in_stride + n * nbins) * sizeof(fdx);
      const logic::InverseStreams::Pick pk = inv.pick(logic::Range{ylo, yhi}, logic::Range{ilo, ihi});
      si = pk.stream;
      SDFT_TRY(hipEventRecord(ev_pre, stream));               // behind everything the main stream has been given (the joins above too)
      SDFT_TRY(hipStreamWaitEvent(row_streams[si], ev_pre, 0));
      // what the OTHER row stream still has outstanding: samples this call overwrites (a host that rotates three sample
      // buffers) or a matrix that was reinterpreted from them -- ordered behind it
      if (pk.wait_other) SDFT_TRY(hipStreamWaitEvent(row_streams[si], ev_inv[si ^ 1], 0));
      stream = row_streams[si];
    }
    struct Restore { hipStream_t& s; hipStream_t v; ~Restore() { s = v; } } restore{stream, main_stream};
    if (!prof_begin(ST_INVERSE)) return false;
    InverseArgs<TD, FD> ia;
    ia.in = in; ia.in_stride = in_stride; ia.in_rows = rows; ia.syn = d_syn.p; ia.y = y; ia.y_stride = y_stride;
    ia.n = n; ia.nbins = (unsigned)nbins; ia.channels = (unsigned)channels; ia.sweight = tab.sweight;
    ia.op = SpectralOp<FD>{}; ia.op.kind = OP_IDENTITY; ia.op.rows = 1;
    ia.done.flag = nullptr; ia.done.count = nullptr; ia.done.seq = 0; ia.done.total = 0;
    {
      const size_t matrix_bytes = channels * n * nbins * sizeof(fdx);
      ia.nt = (int)logic::inverse_streaming_loads(matrix_bytes, opt_inverse_nt);
      ia.nt_skip = logic::inverse_ordinary_rows(matrix_bytes, nbins * sizeof(fdx), opt_inverse_nt_skip_mb);
    }
    const bool ops = op && op->kind != OP_IDENTITY;
    if (ops) ia.op = *op;
    const size_t total_rows = channels * n;
    const bool lat1 = (latency == 1);                                            // :639 exact compare
    if (ops) { if (lat1) launch_inverse<true, true>(ia, total_rows); else launch_inverse<false, true>(ia, total_rows); }
    else     { if (lat1) launch_inverse<true, false>(ia, total_rows); else launch_inverse<false, false>(ia, total_rows); }
    SDFT_TRY(hipGetLastError());
    if (rtc_failed) { rtc_failed = false; return false; }    // (the compiler's words are in the error channel already)
    if (!prof_end(ST_INVERSE)) return false;
    calls.on_synthesis_launched();
    if (inv_pipe)
    {
      SDFT_TRY(hipEventRecord(ev_inv[si], stream));
      inv.launched(si, logic::Range{ylo, yhi});
      ++inv_calls; last_inverse_pipelined = 1;
    }
    return true;
  }

  // synchronous calls: short ones poll the stream (a sleeping hipStreamSynchronize wakes up late --
  // tens of microseconds, more than a whole 100-sample hop takes on the device)
  size_t last_partial_elems = 0;
  bool hop2_stamps = false;
  bool chain_stats(unsigned long long* out32)
  {
#ifdef SDFT_HOP_STAMPS
    // development build: realtime stamps (100 MHz) of forward_hop2_kernel's workgroup 0 (recurrence wave: 0..2, window wave: 4..6)
    if (hop2_stamps && d_partial.p)
    {
      memset(out32, 0, 32 * sizeof(unsigned long long));
      SDFT_TRY(hipStreamSynchronize(stream));
      SDFT_TRY(hipMemcpy(out32, d_partial.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      hop2_stamps = false;
      return true;
    }
#endif
    if (!d_chain_stats.p) return false;
    SDFT_TRY(hipMemcpy(out32, d_chain_stats.p, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return true;
  }
  // development aid: the relay form's stamps (option chain_debug bit 7), 3 per turn of workgroup 0
  bool relay_stamps(unsigned long long* out, size_t count)
  {
    if (!d_chain_stats.p || d_chain_stats.cap < 64 + count) return false;
    SDFT_TRY(hipStreamSynchronize(stream));
    if (aux) SDFT_TRY(hipStreamSynchronize(aux));
    SDFT_TRY(hipMemcpy(out, d_chain_stats.p + 64, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return true;
  }

  // Synchronous calls do not sleep on the stream (a sleeping hipStreamSynchronize wakes up tens of microseconds late):
  // they poll -- the completion word where the call's last kernel sets one, the stream otherwise -- for a bounded
  // wall-clock time (logic::sync_wait: min(20 ms, 200 us + 4 x what the call's bytes take at HBM speed); 50 ms for the word) and only then
  // block.  flag_fallbacks counts completion words that never became visible (get_option "flag_fallbacks").
  long flag_fallbacks = 0;
  static inline void cpu_relax()
  {
#if defined(__SSE2__)
    _mm_pause();
#endif
  }
  // hbm_bytes: what the call has to move through HBM at least (0: unknown -- sleep on the stream)
  size_t matrix_bytes(size_t n) const { return channels * n * nbins * sizeof(fdx); }
  bool finish(size_t hbm_bytes = 0)
  {
    if (async) return true;
    using clock = std::chrono::steady_clock;
    if (flag_pending)
    {
      // the kernel's completion word: visible ~6 us before the stream reports the kernel done
      flag_pending = false;
      volatile unsigned* f = h_done_flag;
      const clock::time_point t0 = clock::now();
      for (unsigned spins = 1;; ++spins)
      {
        if (*f == flag_seq)
        {
          // (the word says nothing about faults; a query of the stream here costs 5 us per hop -- measured 35 -> 46 us
          // for the two reference calls -- so a fault surfaces at the next call that touches the stream, as it does
          // for asynchronous calls)
          const hipError_t e = hipPeekAtLastError();
          if (e != hipSuccess) { set_error("completion word", hipGetErrorString(e)); (void)hipGetLastError(); return false; }
          return true;
        }
        if ((spins & 1023u) == 0 && clock::now() - t0 > std::chrono::milliseconds(50)) break;
      }
      ++flag_fallbacks;
      return synchronize();                                  // never seen: fall back to the stream
    }
    // The wait is the same on every box and never sleeps on the stream while the call can still be running: a sleeping
    // hipStreamSynchronize wakes up 9 us late on one box and 45 us late on another (round 4 slept for calls beyond 60 us and
    // lost 10 % of the north star's synchronous rate on the box it was not tuned on; profiles/r05_sync_completion.txt).  A call
    // cannot end before its bytes have moved at the chip's peak rate (logic::kPeakBytesPerUs -- the spec figure, a LOWER bound of the
    // time, not a tuned one): until then the host spins on its own clock without touching the runtime (queries that cannot
    // succeed yet only compete with the completion signal's handler for the runtime's locks), then it polls the stream.
    // Option "spin": 0 = sleep on the stream, 1 = this (default), 2 = poll from the start.
    if (opt_spin && hbm_bytes)
    {
      const logic::SyncWait w = logic::sync_wait(hbm_bytes);
      const clock::time_point t0 = clock::now();
      if (opt_spin == 1 && w.quiet_us > 0)
      {
        // (counted from the entry point's start: a call that has waited for its kernels already -- the exact-carry route
        // checks its poll loops' status word behind a synchronisation -- is not made to wait again)
        const auto quiet = std::chrono::nanoseconds((long long)(w.quiet_us * 1000.0));
        while (clock::now() - call_start < quiet) cpu_relax();
      }
      const auto budget = std::chrono::microseconds((long long)w.budget_us);
      for (unsigned spins = 1;; ++spins)
      {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return collect_profile();
        if (e != hipErrorNotReady) { set_error("hipStreamQuery", hipGetErrorString(e)); return false; }
        if ((spins & 15u) == 0 && clock::now() - t0 > budget) break;
      }
      (void)hipGetLastError();
    }
    return synchronize();
  }

  // calls whose kernels read and wrote the caller's host memory in place (map_host): always complete on return, and
  // through the stream's own synchronisation -- the point at which the runtime promises the host sees what the device
  // wrote over PCIe (a stream query that reports "done" first keeps the wait short; the synchronisation then returns at once)
  bool finish_mapped(size_t hbm_bytes)
  {
    const bool saved = async; async = false;
    const bool ok = finish(hbm_bytes);
    async = saved;
    if (!ok) return false;
    SDFT_TRY(hipStreamSynchronize(stream));
    return true;
  }

  // Pointer classification: every call asks the runtime (hipPointerGetAttributes).  Measured on MI355X / ROCm 7
  // (scripts/pointer_query_probe.hip, profiles/r04_pointer_query_cost.txt): 0.06-0.10 us per query for device and pinned
  // memory, 0.16 us for pageable host memory, with 2 or 2000 live allocations -- two orders of magnitude below a launch,
  // so nothing is cached and a buffer that was freed and whose address came back as the other kind of memory (hipMalloc
  // does hand a freed address out again) is classified as what it is now.  Option "pointers" = 1 / 2 declares every
  // pointer device / host memory (no query).
  bool on_device(const void* p)
  {
    if (opt_pointers == 1) return true;
    if (opt_pointers == 2) return false;
    return is_device_pointer(p);
  }

  // ---- the caller's host memory (sdft_host_io.hpp): registered in place, or copied through pinned slots of the plan ----------
  HostIo io;
  static constexpr size_t kSmallHostBytes = HostIo::kSmallHostBytes;
  void* map_host(const void* p, size_t bytes, bool will_write = false) { return io.map_host(p, bytes, will_write); }
  // Both are complete on return as far as the caller's memory goes: to_device has read it, to_host has written it.
  bool copy_failed() { set_error("host copy", "a copy between host memory and the device through the plan's pinned slots failed"); return false; }
  bool to_device(void* dst, const void* src, size_t bytes)
  {
    if (bytes == 0) return true;
    if (!pipe_join()) return false;                          // (a no-op unless row streams hold outstanding launches)
    return io.to_device(dst, src, bytes, stream) || copy_failed();
  }
  bool to_host(void* dst, const void* src, size_t bytes)
  {
    if (bytes == 0) return true;
    if (!pipe_join()) return false;
    return io.to_host(dst, src, bytes, stream) || copy_failed();
  }

  // one strip per channel (no pitch limits, works for any size); kind says which side is the caller's host memory
  bool copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, hipMemcpyKind kind)
  {
    for (size_t c = 0; c < channels; ++c)
    {
      void* d = (char*)dst + c * dpitch;
      const void* s = (const char*)src + c * spitch;
      if (kind == hipMemcpyHostToDevice) { if (!to_device(d, s, width)) return false; }
      else if (kind == hipMemcpyDeviceToHost) { if (!to_host(d, s, width)) return false; }
      else { if (!pipe_join()) return false; SDFT_TRY(hipMemcpyAsync(d, s, width, kind, stream)); }
    }
    return true;
  }

  // Small host-side sample buffers (a hop of the host's signal, the by-value sample of sdft_sdft, the sample sdft_isdft
  // returns) do not go through the runtime's pageable copy path (5-10 us per copy): they travel through a pinned scratch
  // of the plan that the kernels read and write directly over PCIe, and the call completes on the kernel's completion
  // word.  Single-sample calls on a device row: 21.9 -> see profiles/r04_single_sample.txt.
  static constexpr size_t kIoBytes = (size_t)64 << 10;      // = kSmallHostBytes: samples in the first half... one direction per call
  TD* h_io = nullptr;
  TD* d_io = nullptr;
  bool ensure_io()
  {
    if (h_io) return true;
    if (hipHostMalloc((void**)&h_io, kIoBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); h_io = nullptr; return false; }
    if (hipHostGetDevicePointer((void**)&d_io, h_io, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h_io); h_io = nullptr; d_io = nullptr; return false; }
    return true;
  }

  // the entry points of the C-ABI (sdft_n / isdft_n / _nd / single samples / the fused call / state access) and the routes
  // host memory takes through them
#include "sdft_plan_entry.inc"
};

}  // namespace sdfthip
