// plan_logic_test.cpp -- unit tests of the engine's host-side decisions (sdft_amd/csrc/sdft_plan_logic.hpp: no HIP anywhere),
// compiled by tests/test_plan_logic_cpu.py with g++ -fsanitize=address,undefined and run in the `-m "not gpu"` suite
// (sanitizers for the host side).  Exits non-zero at the first violated property.

#include "sdft_plan_logic.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <vector>

using namespace sdfthip::logic;

static int failures = 0;
#define CHECK(cond, ...)                                                                    \
  do {                                                                                      \
    if (!(cond)) { ++failures; fprintf(stderr, "%s:%d: %s -- ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); \
      if (failures > 20) exit(1); }                                                         \
  } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static size_t rnd_in(size_t lo, size_t hi) { return lo + (size_t)(rnd() % (unsigned long long)(hi - lo + 1)); }

static void test_geometry()
{
  CHECK(bins_per_lane(16) == 1 && bins_per_lane(8) == 2, "lanes store 16 bytes");
  for (int window = 0; window < 4; ++window)
    for (size_t fdx : {(size_t)8, (size_t)16})
    {
      const long inter = interior_lanes(window, fdx, 0);
      CHECK(inter >= 8 && inter % 8 == 0 && inter + 2 * halo_lanes(window, fdx) <= kLanes, "window %d fdx %zu: %ld interior lanes", window, fdx, inter);
      CHECK(interior_lanes(window, fdx, 1000) == kLanes - 2 * halo_lanes(window, fdx), "forced interior is capped");
      for (size_t nbins : {(size_t)1, (size_t)7, (size_t)56, (size_t)57, (size_t)1000, (size_t)1024, (size_t)4096, (size_t)100000})
      {
        const long t = tiles(nbins, window, fdx, 0);
        const long per = inter * bins_per_lane(fdx);
        CHECK(t * per >= (long)nbins && (t - 1) * per < (long)nbins, "tiles cover the row exactly once: nbins %zu tiles %ld", nbins, t);
      }
    }
  // rows the row-group kernel takes: 8 ... 2048 bins of 16 bytes, ... 4096 bins of 8 bytes, never a row-pointer table
  CHECK(rows_kernel_ok(1024, 16, false, true, 2) && rows_kernel_ok(2048, 16, false, true, 2) && !rows_kernel_ok(2049, 16, false, true, 2), "FD double rows");
  CHECK(rows_kernel_ok(4096, 8, false, true, 2) && !rows_kernel_ok(4097, 8, false, true, 2) && !rows_kernel_ok(4096, 8, false, true, 1), "FD float rows");
  CHECK(!rows_kernel_ok(1024, 16, true, true, 2) && !rows_kernel_ok(1024, 16, false, false, 2) && !rows_kernel_ok(7, 16, false, true, 2), "exclusions");
  for (size_t fdx : {(size_t)8, (size_t)16})
    for (size_t nbins = 8; nbins <= (size_t)(kLanes * kRowWaves * bins_per_lane(fdx) * 2); nbins += 37)
    {
      const long s = row_slots(nbins, fdx), w = row_waves(nbins, fdx);
      CHECK((s == 1 || s == 2) && w >= 1 && w <= kRowWaves, "nbins %zu: %ld slots %ld waves", nbins, s, w);
      CHECK((size_t)(w * s * kLanes * bins_per_lane(fdx)) >= nbins, "the row group holds the row: nbins %zu", nbins);
    }
}

static void test_chunks()
{
  for (int it = 0; it < 200000; ++it)
  {
    ChunkQuery q;
    q.n = (it % 5 == 0) ? rnd_in(1, 600) : rnd_in(1, 3000000);
    q.channels = (it % 3 == 0) ? rnd_in(1, 600) : 1;
    q.nbins = rnd_in(1, 4096);
    q.rows_kernel = rnd() & 1; q.exact = rnd() & 1; q.pipelined = (rnd() & 3) == 0;
    q.forced_chunk = (rnd() & 7) == 0 ? (long)rnd_in(1, 100000) : 0;
    q.target_waves = (rnd() & 7) == 0 ? (long)rnd_in(1, 100000) : 0;
    q.row_waves = (long)rnd_in(1, 16); q.tiles = (long)rnd_in(1, 80); q.compute_units = (int)rnd_in(1, 304);
    const Chunking c = choose_chunks(q);
    CHECK(c.chunks >= 1 && c.len >= 1, "n %zu: %ld chunks of %ld", q.n, c.chunks, c.len);
    CHECK((size_t)c.chunks * (size_t)c.len >= q.n && (size_t)(c.chunks - 1) * (size_t)c.len < q.n, "chunks cover the call exactly once: n %zu %ld x %ld", q.n, c.chunks, c.len);
    if (q.forced_chunk <= 0 && q.n < (size_t)kHopSamples) CHECK(c.chunks == 1, "hop-sized calls are one chunk (bit-exact): n %zu", q.n);
    if (c.chunks > 1 && q.forced_chunk <= 0) CHECK(c.len % kTimeGroup == 0, "whole scalar-load groups: len %ld", c.len);
    if (c.chunks > 1 && q.exact && q.forced_chunk <= 0) CHECK(c.len % 32 == 0, "whole trips of the exact pass: len %ld", c.len);
    if (c.chunks > 1 && !q.exact && q.forced_chunk > 0) CHECK(c.len % kSumBlockLen == 0, "whole sum blocks: len %ld", c.len);
  }
  // the shapes the documents quote
  ChunkQuery q; q.rows_kernel = true; q.row_waves = 16; q.compute_units = 256; q.nbins = 1024;
  q.n = 48000; Chunking c = choose_chunks(q);
  CHECK(c.chunks == 250 && c.len == 192, "north star: %ld x %ld", c.chunks, c.len);
  q.n = 52000; c = choose_chunks(q);
  CHECK(c.chunks <= 256, "between one and two rounds of the chip a call takes ONE round: %ld chunks", c.chunks);
  q.n = 1000000; c = choose_chunks(q);
  CHECK(c.chunks > 256 && c.chunks <= 512, "two rounds at n = 1e6: %ld", c.chunks);
  q.pipelined = true; q.n = 48000; c = choose_chunks(q);
  CHECK(c.chunks == 300 && c.len == 160, "pipelined calls: about 300 chunks of >= 160 rows: %ld x %ld", c.chunks, c.len);
  // where pipelined calls pay: calls below 2^29 bins by default, any length on request, never when off
  q.pipelined = true;
  q.n = 48000; CHECK(pipeline_pays(q, 1) && pipeline_pays(q, 2) && !pipeline_pays(q, 0), "north star: pipelined by default");
  q.n = 131072; CHECK(pipeline_pays(q, 1), "n = 131072 gains 11 %%: pipelined");
  q.n = 1000000; CHECK(!pipeline_pays(q, 1) && pipeline_pays(q, 2), "n = 1e6 is a tie: one stream by default");
  q.n = 48000; q.channels = 64; CHECK(!pipeline_pays(q, 1), "one GPU's share of configs[4]: 3e9 bins");
  q.channels = 1;
  q.pipelined = false; q.exact = true; q.n = 262144; q.nbins = 4096; c = choose_chunks(q);
  CHECK(c.chunks == 2048 && c.len == 128, "configs[2]: %ld x %ld", c.chunks, c.len);
}

static void test_relay_and_radices()
{
  CHECK(relay_block(4096, 128, 4, 8, 0) == 128 && relay_block(1024, 192, 8, 16, 0) == 64 && relay_block(1000, 32, 4, 8, 0) == 16, "block lengths");
  CHECK(relay_block(7, 64, 4, 8, 0) == 0 && relay_block(1024, 100, 4, 8, 0) == 0 && relay_block(1024, 0, 4, 8, 0) == 0, "no block length fits");
  CHECK(relay_block(4096, 128, 4, 8, 32) == 32 && relay_block(4096, 128, 8, 16, 128) == 0, "forced");
  for (size_t nbins = 1; nbins < 5000; nbins += 3)
    for (long len : {8L, 32L, 96L, 128L, 640L, 1000L})
    {
      const unsigned L = relay_block(nbins, len, 4, 8, 0);
      if (L) CHECK((2 * nbins) % L == 0 && (size_t)len % L == 0 && L <= 128, "nbins %zu len %ld: L %u", nbins, len, L);
    }
  for (size_t span = 1; span <= 8192; ++span)
  {
    const Radices r = smooth_radices(span);
    size_t prod = 1;
    for (int i = 0; i < r.count; ++i) { CHECK(r.r[i] >= 2 && r.r[i] <= 5, "radix"); prod *= r.r[i]; }
    size_t rem = span;
    for (size_t f : {(size_t)2, (size_t)3, (size_t)5}) while (rem % f == 0) rem /= f;
    if (rem == 1 && span > 1) CHECK(r.count > 0 && prod == span, "a 2/3/5-smooth span factors completely: %zu", span);
    else CHECK(r.count == 0, "other prime factors: %zu", span);
  }
  CHECK(self_cells(1024, true, 16) == 2048 && self_cells(1000, true, 16) == 4000 && self_cells(1000, false, 16) == 0 && self_cells(1022, true, 16) == 0, "self cells");
  CHECK(self_cells(4096, true, 16) == 0 && self_cells(2048, true, 16) == 4096 && self_cells(4, true, 16) == 0, "self cells, limits");
}

static void test_call_pattern()
{
  CallPattern p;
  p.on_analysis(false); CHECK(!p.analysis_batch, "one analysis is not a run");
  p.on_analysis(false); CHECK(p.analysis_batch, "two analyses in a row");
  p.on_synthesis_begin(); p.on_synthesis_launched(); CHECK(p.analysis_batch && !p.inverse_batch, "a synthesis after a run of analyses");
  p.on_synthesis_begin(); p.on_synthesis_launched(); CHECK(p.inverse_batch, "two syntheses in a row");
  p.on_analysis(false); CHECK(p.inverse_batch, "an analysis after a run of syntheses keeps the mode");
  // the reference's loop: analysis, synthesis, analysis, synthesis ... : both modes go off and stay off
  CallPattern r;
  for (int i = 0; i < 6; ++i)
  {
    r.on_analysis(false);
    r.on_synthesis_begin(); r.on_synthesis_launched();
    if (i >= 1) CHECK(!r.analysis_batch && !r.inverse_batch, "alternating calls never leave the plan's stream (hop %d)", i);
  }
  // the fused call is neither kind
  CallPattern f;
  f.on_analysis(false); f.on_analysis(true); f.on_analysis(false);
  CHECK(!f.analysis_batch, "a fused call between two analyses breaks the run");
}

static void test_row_ring()
{
  RowRing ring;
  const uintptr_t A = 0x100000, B = 0x900000, S = 0x400000;
  RowRing::Pick p = ring.pick(A, A + S);
  CHECK(p.stream == 0 && !p.behind && ring.state_reader() < 0, "first launch");
  ring.launched(A, A + S, p.stream);
  p = ring.pick(B, B + S);
  CHECK(p.stream == 1 && !p.behind, "another matrix: the other stream");
  ring.launched(B, B + S, p.stream);
  p = ring.pick(A, A + S);
  CHECK(p.stream == 0 && p.behind && p.wait_launch < 0, "the first matrix again: behind its own launch, on its stream");
  ring.launched(A, A + S, p.stream);
  CHECK(ring.state_reader() == 0, "the fourth launch's state kernel waits for the rows of the first (slot 0)");
  p = ring.pick(A + S / 2, B + S / 2);
  CHECK(p.behind && p.wait_launch >= 0 && ring.stream_of[p.wait_launch] != p.stream, "a matrix across both: the other stream's launch is waited for");
  CHECK(ring.samples_overlap(B + 8, B + 16) && !ring.samples_overlap(0x10, 0x20), "samples inside an outstanding matrix");
  int slots[2];
  CHECK(ring.last_per_stream(slots) == 2 && ring.stream_of[slots[0]] != ring.stream_of[slots[1]], "join waits for the last launch on each stream");
  ring.joined();
  CHECK(!ring.open && ring.seq == 0 && ring.last_per_stream(slots) == 0, "joined");
  // random sequences: the pick is always one of the two streams, `behind` exactly when an outstanding launch overlaps
  for (int it = 0; it < 20000; ++it)
  {
    if ((rnd() & 31) == 0) ring.joined();
    const uintptr_t lo = (uintptr_t)rnd_in(0, 64) * 0x1000, hi = lo + (uintptr_t)rnd_in(1, 16) * 0x1000;
    bool any = false;
    for (unsigned long long back = 1; back <= 3 && back <= ring.seq; ++back) any = any || overlap(lo, hi, ring.out[(ring.seq - back) & 3]);
    const RowRing::Pick q = ring.pick(lo, hi);
    CHECK((q.stream == 0 || q.stream == 1) && q.behind == any, "pick: stream %d behind %d any %d", q.stream, (int)q.behind, (int)any);
    if (q.wait_launch >= 0) CHECK(ring.stream_of[q.wait_launch] != q.stream && overlap(lo, hi, ring.out[q.wait_launch]), "waited launch overlaps, other stream");
    ring.launched(lo, hi, q.stream);
  }
}

static void test_inverse_streams()
{
  InverseStreams inv;
  const Range y1{0x1000, 0x2000}, y2{0x3000, 0x4000}, y3{0x5000, 0x6000}, none{0, 0};
  InverseStreams::Pick p = inv.pick(y1, none); inv.launched(p.stream, y1);
  const int first = p.stream;
  p = inv.pick(y2, none); CHECK(p.stream != first && !p.wait_other, "two sample buffers: two streams"); inv.launched(p.stream, y2);
  p = inv.pick(y1, none); CHECK(p.stream == first && !p.wait_other, "the first buffer again: behind its writer, same stream"); inv.launched(p.stream, y1);
  // three buffers in rotation (the advisor's case): y3 -> y1's stream ... then y1 must wait for its earlier write on the other stream
  InverseStreams r;
  p = r.pick(y1, none); r.launched(p.stream, y1); const int sa = p.stream;
  p = r.pick(y2, none); r.launched(p.stream, y2);
  p = r.pick(y3, none); CHECK(p.stream == sa, "third buffer takes the first stream again"); r.launched(p.stream, y3);
  p = r.pick(y1, none); r.launched(p.stream, y1);
  p = r.pick(y2, none);
  CHECK(p.stream != r.last || p.wait_other || overlap(y2, r.y[p.stream]), "a write to y2 is ordered behind the outstanding write of y2");
  // a matrix that is an outstanding synthesis's samples reinterpreted
  InverseStreams m;
  p = m.pick(y1, none); m.launched(p.stream, y1);
  p = m.pick(y2, y1); CHECK(p.wait_other, "reads what the other stream is writing: waits");
}

static void test_small_decisions()
{
  HopParts h = hop_parts(100, 18, 256, 0, true);
  CHECK(h.parts == 4 && h.part_len == 25, "synchronous hop of the reference's test: %u x %u", h.parts, h.part_len);
  h = hop_parts(100, 18, 256, 0, false);
  CHECK(h.parts == 8 && h.part_len == 13, "asynchronous: %u x %u", h.parts, h.part_len);
  h = hop_parts(100, 200, 256, 0, false); CHECK(h.parts == 1, "no CU to spare: one part");
  h = hop_parts(23, 18, 256, 0, false); CHECK(h.parts == 1 && h.part_len == 23, "too short");
  h = hop_parts(100, 18, 256, 1, false); CHECK(h.parts == 1, "option: never");
  for (size_t n = 1; n < 600; ++n)
    for (long forced : {0L, 2L, 5L, 16L, 1000L})
    {
      h = hop_parts(n, 18, 256, forced, (n & 1) != 0);
      CHECK(h.parts >= 1 && (size_t)h.parts * h.part_len >= n && (size_t)(h.parts - 1) * h.part_len < n, "parts cover the call once: n %zu forced %ld: %u x %u", n, forced, h.parts, h.part_len);
    }
  CHECK(inverse_rows_per_wave(100, 8, 0, 0, 0, false) == 1 && inverse_rows_per_wave(48000, 8, 0, 0, 0, false) == 4 && inverse_rows_per_wave(1000000, 8, 0, 0, 0, false) == 32, "rows per wave");
  CHECK(inverse_rows_per_wave(1000000, 4, 0, 0, 0, false) == 16 && inverse_rows_per_wave(48000, 8, 0, 8192, 8192, false) == 8, "rows per wave: float bins; a second round avoided");
  CHECK(inverse_rows_per_wave(48000, 8, 0, 16384, 8192, false) == 4 && inverse_rows_per_wave(48000, 8, 0, 8192, 8192, true) == 16 && inverse_rows_per_wave(48000, 8, 32, 8192, 8192, false) == 32, "rows per wave: fits / operation / forced");
  ProcessGeometry g = process_geometry(1024, 1, 1000000, true, 8, 0);
  CHECK(g.slots == 4 && g.waves == 4, "fused call, long: %ld waves x %ld bins per lane", g.waves, g.slots);
  g = process_geometry(1024, 1, 48000, true, 8, 0); CHECK(g.slots == 2 && g.waves == 8, "fused call, north star: %ld x %ld", g.waves, g.slots);
  g = process_geometry(1024, 1, 48000, false, 4, 0); CHECK(g.slots == 1 && g.waves == 16, "FD float: %ld x %ld", g.waves, g.slots);
  for (size_t nbins = 8; nbins <= 4096; nbins += 11)
  {
    g = process_geometry(nbins, 1, 500000, true, 8, 0);
    CHECK((size_t)(g.waves * g.slots * kLanes) >= nbins && g.waves >= 1 && g.waves <= kRowWaves, "fused geometry holds the row: nbins %zu", nbins);
  }
  const SyncWait w = sync_wait((size_t)48000 * 1024 * 16);
  CHECK(w.quiet_us > 90 && w.quiet_us < 110 && w.budget_us > w.quiet_us * 2, "north star: quiet %.1f us budget %.1f us", w.quiet_us, w.budget_us);
  CHECK(sync_wait(1000).quiet_us == 0 && sync_wait((size_t)1 << 40).budget_us <= 20000.0, "short and huge calls");
  CHECK(stage_rows(1000, 16384, (size_t)1 << 30) == 1000 && stage_rows(1000000, 16384, (size_t)1 << 30) == 65536 && stage_rows(5, 0, 100) == 5 && stage_rows(10, 1000, 10) == 1, "staging segments");
}

static void test_form_tuner()
{
  FormTuner t;
  t.reset(1000, 3);
  bool timed = false;
  // a host that waits for every call: the candidates in turn, two samples each, the smaller one counts
  const float ms[3][2] = {{2.0f, 1.9f}, {1.5f, 1.6f}, {1.8f, 3.0f}};
  int seen[3] = {0, 0, 0};
  for (int i = 0; i < 6; ++i)
  {
    const int f = t.next(true, timed);
    CHECK(timed && f >= 0 && f < 3 && seen[f] < 2, "trial %d: form %d", i, f);
    t.launched(f); t.report(f, ms[f][seen[f]]); ++seen[f];
  }
  int f = t.next(true, timed);
  CHECK(!timed && f == 1 && t.chosen == 1, "the fastest form (by its smaller sample) is chosen: %d", f);
  CHECK(t.next(false, timed) == 1 && !timed, "and stays");
  // a host that queues calls faster than they run: one trial per form in flight, then the static form untimed
  t.reset(5000, 3);
  for (int i = 0; i < 3; ++i) { f = t.next(true, timed); CHECK(timed && f == i, "flood: form %d in flight", f); t.launched(f); }
  f = t.next(true, timed); CHECK(!timed && f == 0 && t.chosen < 0, "every open form in flight: the static form, untimed");
  t.report(1, 1.0f);
  f = t.next(true, timed); CHECK(timed && f == 1, "a form that has reported takes its second sample");
  t.reset(2000, 1); CHECK(t.next(true, timed) == 0 && !timed && t.chosen == 0, "one candidate: nothing to measure");
  t.reset(3000, 9); CHECK(t.count == FormTuner::kMax, "capped");
  t.reset(3000, 2);
  CHECK(t.next(false, timed) == 0 && !timed, "no timing possible: the static form");
  t.launched(1); t.report(7, 1.0f); t.report(1, -1.0f); CHECK(t.samples[1] == 0 && !t.inflight[1], "a failed sample frees the form and counts nothing");
  // a host that alternates between call lengths: each shape keeps its tuner (and what it decided); a fourth shape replaces the least recently used
  {
    TunerTable tab;
    const int a = tab.find(1000, 3), b = tab.find(2000, 3);
    CHECK(a != b && tab.slot[a].key == 1000 && tab.slot[b].key == 2000, "two shapes, two tuners");
    for (int i = 0; i < 6; ++i) { FormTuner& q = tab.slot[tab.find(1000, 3)]; const int g = q.next(true, timed); q.launched(g); q.report(g, 1.0f + g); (void)tab.find(2000, 3); }
    FormTuner& qa = tab.slot[tab.find(1000, 3)];
    CHECK(qa.next(true, timed) == 0 && qa.chosen == 0 && !timed, "the interleaved shape did not make the first one start over");
    CHECK(tab.slot[tab.find(2000, 3)].chosen < 0, "the second shape is still open");
    const int c = tab.find(3000, 3);
    CHECK(c != a && c != b, "a third shape takes the free slot");
    (void)tab.find(1000, 3); (void)tab.find(3000, 3);
    const int d = tab.find(4000, 2);
    CHECK(d == b && tab.slot[d].key == 4000 && tab.slot[d].count == 2 && tab.slot[d].chosen < 0, "a fourth shape replaces the least recently used (2000)");
    CHECK(tab.slot[tab.find(1000, 3)].chosen == 0, "... and the decided one is still there");
    CHECK(tab.find(1000, 4) >= 0 && tab.slot[tab.find(1000, 4)].count == 4, "another candidate count is another tuner");
    tab.reset_all();
    CHECK(tab.slot[tab.find(1000, 3)].chosen < 0, "reset_all forgets");
  }
  // the kind of load of the synthesis by the size of the matrix: the window of round 4, the very large matrices of round 5, the option
  const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
  CHECK(!inverse_streaming_loads(200 * MiB, -1) && inverse_streaming_loads(300 * MiB, -1) && inverse_streaming_loads(4 * GiB, -1), "the window");
  CHECK(inverse_streaming_loads(4 * GiB + 1, -1) && inverse_streaming_loads(16 * GiB, -1) && inverse_streaming_loads(50 * GiB, -1), "beyond it");
  CHECK(inverse_streaming_loads(1, 1) && !inverse_streaming_loads(GiB, 0), "forced");
  // ... but for the rows read first: 1.5 GB of matrices from 6 GiB on, whole rows, none below; the option
  CHECK(inverse_ordinary_rows(16 * GiB, 16384, -1) == 98304 && inverse_ordinary_rows(5 * GiB, 16384, -1) == 0 && inverse_ordinary_rows(16 * GiB, 16000, -1) == 100664, "rows read first");
  CHECK(inverse_ordinary_rows(GiB, 16384, 256) == 16384 && inverse_ordinary_rows(16 * GiB, 16384, 0) == 0 && inverse_ordinary_rows(GiB, 0, 100) == 0, "forced rows");
}

static void test_piece_ring()
{
  for (int it = 0; it < 20000; ++it)
  {
    const size_t bytes = rnd_in(0, 40) == 0 ? 0 : rnd_in(1, (size_t)64 << 20), piece = rnd_in(1, (size_t)4 << 20);
    const unsigned slots = (unsigned)rnd_in(1, 8);
    const PieceRing r(bytes, piece, slots);
    size_t covered = 0;
    std::vector<size_t> occupant(slots, (size_t)-1);
    for (size_t i = 0; i < r.pieces(); ++i)
    {
      CHECK(r.offset(i) == covered && r.length(i) >= 1 && r.length(i) <= piece, "pieces tile the copy: piece %zu", i);
      covered += r.length(i);
      CHECK(r.predecessor(i) == occupant[r.slot(i)], "a slot's previous occupant is the piece that must leave first");
      occupant[r.slot(i)] = i;
    }
    CHECK(covered == bytes, "all bytes: %zu of %zu", covered, bytes);
  }
}

int main()
{
  test_geometry();
  test_chunks();
  test_relay_and_radices();
  test_call_pattern();
  test_row_ring();
  test_inverse_streams();
  test_small_decisions();
  test_piece_ring();
  test_form_tuner();
  if (failures) { fprintf(stderr, "%d failure(s)\n", failures); return 1; }
  printf("plan logic: all properties hold\n");
  return 0;
}
