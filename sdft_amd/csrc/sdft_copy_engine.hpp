// sdft_copy_engine.hpp -- copies between the caller's pageable host memory and the device through pinned slots of the plan,
// with a few host threads so that the host's copy of one piece overlaps the DMA of the next (round 5).  No HIP dependency: the
// device side is a policy (`Dev`: dma(), record(), wait()), so the whole engine -- slot ring, worker pool, hand-offs -- is
// compiled and run on the CPU under ThreadSanitizer and AddressSanitizer (tests/cpp/copy_engine_test.cpp) with a mock device.
//
// Why the library copies at all ("Host buffers"): the runtime's own path for pageable memory pins the
// caller's pages and remembers the pin by address; a host that frees such a buffer and gets the address back later makes the
// next copy fault the GPU.  So nothing of the caller's is handed to the runtime: bytes travel caller -> pinned slot -> device and
// back.  Round 4 did the host side of that with one thread: 26 GB/s on long copies against the DMA engine's 55.
//
// Pipeline per piece i (slot i % slots; piece i may enter its slot once piece i - slots has left it):
//   to the device:  FILL  (a worker copies the caller's bytes into the slot)  ->  SEND  (the calling thread queues the DMA)
//   to the host:    SEND  (the calling thread queues the DMA into the slot)   ->  DRAIN (a worker copies the slot out)
// The calling thread is the only one that talks to the stream (runtime calls stay on the caller's thread); workers wait for a
// slot's DMA through Dev::wait (an event synchronisation, which is thread-safe) and do nothing but memcpy.

#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "sdft_plan_logic.hpp"

namespace sdfthip {

// The host's copies between the caller's memory and pinned slots: streaming (non-temporal) stores for anything beyond 256 KiB
// -- the destination is not read again by this core, and without the read-for-ownership of every line the copy out of memory
// the device has just written runs at 42 instead of 30 GB/s (scripts/host_memcpy_probe.hip, profiles/r04_host_copy_paths.txt)
static inline void host_copy_bytes(void* dst_, const void* src_, size_t bytes)
{
#if defined(__SSE2__)
  if (bytes >= ((size_t)256 << 10))
  {
    char* dst = static_cast<char*>(dst_);
    const char* src = static_cast<const char*>(src_);
    size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
    memcpy(dst, src, head); dst += head; src += head; bytes -= head;
    const size_t blocks = bytes / 64;
    for (size_t i = 0; i < blocks; ++i)
    {
      const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 0), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 1);
      const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 2), d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 3);
      _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 0, a); _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 1, b);
      _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 2, c); _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 3, d);
      src += 64; dst += 64;
    }
    _mm_sfence();
    memcpy(dst, src, bytes - blocks * 64);
    return;
  }
#endif
  memcpy(dst_, src_, bytes);
}

static inline void cpu_relax()
{
#if defined(__SSE2__)
  _mm_pause();
#endif
}
// a polite spin: pauses, and gives the core away now and then (a host with fewer cores than threads must not starve the
// thread that is being waited for)
struct Spin
{
  unsigned n = 0;
  void operator()() { if ((++n & 63u) == 0) std::this_thread::yield(); else cpu_relax(); }
};

// A few worker threads that copy memory for the calling thread.  Started on first use, parked on a condition variable when
// idle; after a job a worker keeps spinning for `linger` (a host that calls hop after hop finds them awake: waking a parked
// thread costs more than a hop-sized copy takes).
class CopyPool
{
 public:
  using Job = void (*)(void* ctx, unsigned worker);
  ~CopyPool() { stop(); }
  unsigned workers() const { return (unsigned)threads_.size(); }
  bool start(unsigned count)
  {
    if (!threads_.empty() || count == 0) return !threads_.empty();
    stop_.store(false);
    unsigned long long born;                                 // (a worker serves the dispatches that come after start(), whenever it gets to run:
    { std::lock_guard<std::mutex> lock(mu_); born = generation_; }      //  not the last job of a pool that was stopped, and not none)
    try { for (unsigned w = 0; w < count; ++w) threads_.emplace_back([this, w, born]() { run(w, born); }); }
    catch (...) { stop(); return false; }
    return true;
  }
  void stop()
  {
    { std::lock_guard<std::mutex> lock(mu_); stop_.store(true); ++generation_; posted_.store(generation_, std::memory_order_release); }
    cv_.notify_all();
    for (std::thread& t : threads_) if (t.joinable()) t.join();
    threads_.clear();
    job_ = nullptr; ctx_ = nullptr;
  }
  // runs job(ctx, w) on every worker (w = 0 .. workers-1) and returns at once; wait() returns when all have finished.
  // One dispatch at a time (the calling thread owns the pool).
  void dispatch(Job job, void* ctx)
  {
    pending_.store((unsigned)threads_.size(), std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lock(mu_); job_ = job; ctx_ = ctx; ++generation_; posted_.store(generation_, std::memory_order_release); }
    cv_.notify_all();
  }
  void wait() { Spin spin; while (pending_.load(std::memory_order_acquire) != 0) spin(); }
  std::chrono::microseconds linger{300};

 private:
  void run(unsigned w, unsigned long long seen)
  {
    for (;;)
    {
      // spin for a while (posted_ is the generation of the latest dispatch), then park
      const auto t0 = std::chrono::steady_clock::now();
      bool got = false;
      Spin spin;
      for (unsigned spins = 1; !got; ++spins)
      {
        if (posted_.load(std::memory_order_acquire) != seen) { got = true; break; }
        if (stop_.load(std::memory_order_relaxed)) return;
        spin();
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > linger) break;
      }
      Job job; void* ctx;
      {
        std::unique_lock<std::mutex> lock(mu_);
        if (!got) cv_.wait(lock, [&]() { return generation_ != seen; });
        if (stop_.load()) return;
        seen = generation_; job = job_; ctx = ctx_;
      }
      job(ctx, w);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  unsigned long long generation_ = 0;                        // guarded by mu_
  std::atomic<unsigned long long> posted_{0};                // = generation_ of the latest dispatch (what a spinning worker watches)
  std::atomic<bool> stop_{false};
  std::atomic<unsigned> pending_{0};
  Job job_ = nullptr;
  void* ctx_ = nullptr;
};

// one memcpy shared by the calling thread and the pool's workers (hop-sized matrices: one piece, nothing to pipeline)
inline void parallel_copy(CopyPool* pool, void* dst, const void* src, size_t bytes)
{
  const unsigned helpers = (pool && bytes >= ((size_t)512 << 10)) ? pool->workers() : 0;
  if (helpers == 0) { host_copy_bytes(dst, src, bytes); return; }
  struct Ctx { char* dst; const char* src; size_t bytes; unsigned parts; } c{static_cast<char*>(dst), static_cast<const char*>(src), bytes, helpers + 1};
  auto share = [](const Ctx& k, unsigned part, size_t& off, size_t& len)
  {
    const size_t per = ((k.bytes / k.parts) + 63) & ~(size_t)63;          // whole cache lines per share
    off = std::min(k.bytes, per * part);
    len = (part + 1 == k.parts) ? k.bytes - off : std::min(per, k.bytes - off);
  };
  struct Call { Ctx ctx; decltype(share)* fn; } call{c, &share};
  pool->dispatch([](void* p, unsigned w) {
    Call* k = static_cast<Call*>(p);
    size_t off, len; (*k->fn)(k->ctx, w + 1, off, len);
    if (len) host_copy_bytes(k->ctx.dst + off, k->ctx.src + off, len);
  }, &call);
  size_t off, len; share(c, 0, off, len);
  if (len) host_copy_bytes(c.dst + off, c.src + off, len);
  pool->wait();
}

// The pipelined copies.  Dev: bool dma_to_device(void* dst, const void* slot, size_t len); bool dma_to_host(void* slot, const
// void* src, size_t len); bool record(unsigned slot); bool wait(unsigned slot)  (wait: callable from any thread; the others
// only from the calling thread).  `slot_mem` = slots x piece bytes of pinned memory.
template <typename Dev>
class PieceCopier
{
 public:
  PieceCopier(Dev& dev, CopyPool* pool, char* slot_mem, size_t piece, unsigned slots) : dev_(dev), pool_(pool), mem_(slot_mem), piece_(piece), slots_(slots) {}

  // returns when every byte of `src` has been read (the last DMAs may still be in flight: recorded on their slots)
  bool to_device(void* dst, const void* src, size_t bytes)
  {
    const logic::PieceRing ring(bytes, piece_, slots_);
    const size_t pieces = ring.pieces();
    if (pieces == 0) return true;
    State st(pieces);
    Ctx ctx{this, &ring, &st, static_cast<char*>(dst), static_cast<const char*>(src), true};
    const bool threaded = pool_ && pool_->workers() > 0 && pieces > 1;
    if (threaded) pool_->dispatch(&PieceCopier::worker_entry, &ctx);
    bool ok = true;
    for (size_t i = 0; i < pieces && ok; ++i)
    {
      if (!threaded) ok = fill(ctx, i);
      else { Spin spin; while (!st.done[i].load(std::memory_order_acquire)) { if (st.failed.load(std::memory_order_relaxed)) { ok = false; break; } spin(); } }
      if (!ok) break;
      ok = dev_.dma_to_device(ctx.dev_side + ring.offset(i), mem_ + (size_t)ring.slot(i) * piece_, ring.length(i)) && dev_.record(ring.slot(i));
      st.queued[i].store(true, std::memory_order_release);
    }
    if (!ok) st.failed.store(true);
    if (threaded) pool_->wait();
    return ok && !st.failed.load();
  }

  // returns when every byte of `dst` has been written
  bool to_host(void* dst, const void* src, size_t bytes)
  {
    const logic::PieceRing ring(bytes, piece_, slots_);
    const size_t pieces = ring.pieces();
    if (pieces == 0) return true;
    State st(pieces);
    Ctx ctx{this, &ring, &st, const_cast<char*>(static_cast<const char*>(src)), static_cast<const char*>(dst), false};
    const bool threaded = pool_ && pool_->workers() > 0 && pieces > 1;
    if (threaded) pool_->dispatch(&PieceCopier::worker_entry, &ctx);
    bool ok = true;
    for (size_t i = 0; i < pieces && ok; ++i)
    {
      // the slot's previous occupant has to be drained first
      const size_t prev = ring.predecessor(i);
      if (prev != (size_t)-1)
      {
        if (!threaded) ok = drain(ctx, prev);
        else { Spin spin; while (!st.done[prev].load(std::memory_order_acquire)) { if (st.failed.load(std::memory_order_relaxed)) { ok = false; break; } spin(); } }
      }
      if (!ok) break;
      ok = dev_.dma_to_host(mem_ + (size_t)ring.slot(i) * piece_, ctx.dev_side + ring.offset(i), ring.length(i)) && dev_.record(ring.slot(i));
      st.queued[i].store(true, std::memory_order_release);
    }
    if (!ok) st.failed.store(true);
    if (threaded) pool_->wait();
    else for (size_t i = (pieces > slots_ ? pieces - slots_ : 0); i < pieces && ok; ++i) ok = drain(ctx, i);    // the pieces still in their slots
    return ok && !st.failed.load();
  }

 private:
  struct State
  {
    explicit State(size_t pieces) : done(pieces), queued(pieces) { for (auto& d : done) d.store(false); for (auto& q : queued) q.store(false); }
    std::vector<std::atomic<bool>> done;       // to the device: filled; to the host: drained
    std::vector<std::atomic<bool>> queued;     // the piece's DMA has been queued and recorded on its slot
    std::atomic<size_t> next{0};               // next piece a worker takes
    std::atomic<bool> failed{false};
  };
  struct Ctx { PieceCopier* self; const logic::PieceRing* ring; State* st; char* dev_side; const char* host_side; bool to_device; };

  // FILL piece i: its slot is free once the DMA of piece i - slots out of it has completed
  bool fill(Ctx& c, size_t i)
  {
    const size_t prev = c.ring->predecessor(i);
    if (prev != (size_t)-1)
    {
      Spin spin;
      while (!c.st->queued[prev].load(std::memory_order_acquire)) { if (c.st->failed.load(std::memory_order_relaxed)) return false; spin(); }
      if (!dev_.wait(c.ring->slot(i))) return false;
    }
    else if (!dev_.wait(c.ring->slot(i))) return false;        // (an earlier copy's DMA may still be reading the slot)
    host_copy_bytes(mem_ + (size_t)c.ring->slot(i) * piece_, c.host_side + c.ring->offset(i), c.ring->length(i));
    c.st->done[i].store(true, std::memory_order_release);
    return true;
  }
  // DRAIN piece i: once its DMA into the slot has completed
  bool drain(Ctx& c, size_t i)
  {
    Spin spin;
    while (!c.st->queued[i].load(std::memory_order_acquire)) { if (c.st->failed.load(std::memory_order_relaxed)) return false; spin(); }
    if (!dev_.wait(c.ring->slot(i))) return false;
    host_copy_bytes(const_cast<char*>(c.host_side) + c.ring->offset(i), mem_ + (size_t)c.ring->slot(i) * piece_, c.ring->length(i));
    c.st->done[i].store(true, std::memory_order_release);
    return true;
  }
  static void worker_entry(void* p, unsigned)
  {
    Ctx& c = *static_cast<Ctx*>(p);
    const size_t pieces = c.ring->pieces();
    for (;;)
    {
      const size_t i = c.st->next.fetch_add(1, std::memory_order_relaxed);
      if (i >= pieces || c.st->failed.load(std::memory_order_relaxed)) return;
      const bool ok = c.to_device ? c.self->fill(c, i) : c.self->drain(c, i);
      if (!ok) { c.st->failed.store(true); return; }
    }
  }

  Dev& dev_;
  CopyPool* pool_;
  char* mem_;
  size_t piece_;
  unsigned slots_;
};

}  // namespace sdfthip
