// copy_engine_test.cpp -- the host-copy engine (sdft_amd/csrc/sdft_copy_engine.hpp: slot ring, worker pool, hand-offs between
// the calling thread, the workers and the DMA engine) on the CPU with a MOCK device: a thread that executes the queued DMAs in
// order, like a stream, and completes the recorded events when it reaches them.  Compiled and run by
// tests/test_plan_logic_cpu.py under -fsanitize=thread (a slot touched by a worker while a DMA still reads or writes it is a
// data race the sanitizer sees) and under -fsanitize=address,undefined.

#include "sdft_copy_engine.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <deque>
#include <functional>

using namespace sdfthip;

// the mock device: an in-order command queue executed by its own thread
class MockDevice
{
 public:
  explicit MockDevice(unsigned slots) : issued_(slots, 0), completed_(slots, 0), thread_([this]() { run(); }) {}
  ~MockDevice()
  {
    { std::lock_guard<std::mutex> lock(mu_); stop_ = true; }
    cv_.notify_all();
    thread_.join();
  }
  bool dma_to_device(void* dst, const void* slot, size_t len) { push([=]() { memcpy(dst, slot, len); }); return true; }
  bool dma_to_host(void* slot, const void* src, size_t len) { push([=]() { memcpy(slot, src, len); }); return true; }
  bool record(unsigned slot)
  {
    unsigned long long ticket;
    { std::lock_guard<std::mutex> lock(mu_); ticket = ++issued_[slot]; }
    push([this, slot, ticket]() { { std::lock_guard<std::mutex> lock(mu_); completed_[slot] = ticket; } done_.notify_all(); });
    return true;
  }
  bool wait(unsigned slot)                                   // any thread: the slot's latest recorded event
  {
    std::unique_lock<std::mutex> lock(mu_);
    const unsigned long long want = issued_[slot];
    done_.wait(lock, [&]() { return completed_[slot] >= want; });
    return true;
  }
  void drain_all()
  {
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [&]() { return queue_.empty() && !busy_; });
  }

 private:
  void push(std::function<void()> f)
  {
    { std::lock_guard<std::mutex> lock(mu_); queue_.push_back(std::move(f)); }
    cv_.notify_all();
  }
  void run()
  {
    for (;;)
    {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&]() { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;
        f = std::move(queue_.front()); queue_.pop_front(); busy_ = true;
      }
      f();
      { std::lock_guard<std::mutex> lock(mu_); busy_ = false; }
      done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::deque<std::function<void()>> queue_;
  std::vector<unsigned long long> issued_, completed_;
  bool stop_ = false, busy_ = false;
  std::thread thread_;
};

static unsigned long long rng_state = 88172645463325252ull;
static unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv)
{
  const int rounds = argc > 1 ? atoi(argv[1]) : 60;
  int failures = 0;
  for (unsigned workers = 0; workers <= 3; ++workers)
  {
    CopyPool pool;
    pool.linger = std::chrono::microseconds(50);
    if (workers && !pool.start(workers)) { fprintf(stderr, "no threads\n"); return 1; }
    for (int it = 0; it < rounds; ++it)
    {
      const unsigned slots = 1 + (unsigned)(rnd() % 4);
      const size_t piece = (size_t)64 << (rnd() % 9);                              // 64 B ... 16 KiB
      const size_t bytes = (it % 7 == 0) ? (rnd() % 3) * piece : (size_t)(rnd() % (40 * piece + 1));
      std::vector<char> host(bytes + 1), device(bytes + 1, 0), back(bytes + 1, 0), slot_mem(slots * piece);
      for (size_t i = 0; i < bytes; ++i) host[i] = (char)(rnd() >> 11);
      MockDevice dev(slots);
      PieceCopier<MockDevice> copier(dev, workers ? &pool : nullptr, slot_mem.data(), piece, slots);
      if (!copier.to_device(device.data(), host.data(), bytes)) { ++failures; continue; }
      // (to_device returns when the caller's bytes have been read: they may be overwritten at once)
      std::vector<char> want(host.begin(), host.begin() + bytes);
      for (size_t i = 0; i < bytes; ++i) host[i] = 0x55;
      if (!copier.to_host(back.data(), device.data(), bytes)) { ++failures; continue; }
      dev.drain_all();
      if (bytes && (memcmp(back.data(), want.data(), bytes) != 0 || memcmp(device.data(), want.data(), bytes) != 0))
      {
        ++failures;
        fprintf(stderr, "mismatch: workers %u slots %u piece %zu bytes %zu\n", workers, slots, piece, bytes);
      }
      // one shared memcpy (hop-sized matrices)
      std::vector<char> big((size_t)(600 << 10) + (size_t)(rnd() % 4096)), copy(big.size());
      for (size_t i = 0; i < big.size(); i += 61) big[i] = (char)(rnd() >> 7);
      parallel_copy(workers ? &pool : nullptr, copy.data(), big.data(), big.size());
      if (memcmp(copy.data(), big.data(), big.size()) != 0) { ++failures; fprintf(stderr, "parallel_copy mismatch: workers %u\n", workers); }
    }
    pool.stop();
    // a pool can be started again after it was stopped (a job dispatched at once is served, the last job is not replayed),
    // and stopped while idle workers are parked
    if (workers)
    {
      if (!pool.start(workers)) ++failures;
      std::vector<char> a((size_t)700 << 10, 3), b(a.size(), 0);
      parallel_copy(&pool, b.data(), a.data(), a.size());
      if (memcmp(a.data(), b.data(), a.size()) != 0) ++failures;
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  if (failures) { fprintf(stderr, "%d failure(s)\n", failures); return 1; }
  printf("copy engine: all copies arrived\n");
  return 0;
}
