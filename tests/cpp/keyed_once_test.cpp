// keyed_once_test.cpp -- the cache behind the run-time compilation of a host's spectral operation (sdft_keyed_once.hpp) under
// ThreadSanitizer: many threads ask for a few keys at once; every key is made exactly once (while it succeeds), nobody gets
// another key's value, a failed make is retried by the next caller, and no make runs under the cache's lock (makes overlap).

#include "sdft_keyed_once.hpp"

#include <stdio.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

int main()
{
  sdfthip::KeyedOnce<int> cache;
  std::atomic<int> makes[8], overlap{0}, max_overlap{0}, wrong{0}, failures_seen{0};
  for (auto& m : makes) m.store(0);
  std::atomic<bool> fail_first{true};
  std::vector<std::thread> threads;
  for (int t = 0; t < 16; ++t)
    threads.emplace_back([&, t]() {
      for (int round = 0; round < 40; ++round)
      {
        const int k = (t + round) % 8;
        int v = -1;
        const bool ok = cache.get("key" + std::to_string(k), v, [&](int& out) -> bool {
          const int now = overlap.fetch_add(1) + 1;
          int seen = max_overlap.load();
          while (now > seen && !max_overlap.compare_exchange_weak(seen, now)) {}
          std::this_thread::sleep_for(std::chrono::milliseconds(2));        // a "compilation"
          overlap.fetch_sub(1);
          if (k == 3 && fail_first.exchange(false)) return false;           // the first attempt at key 3 fails
          makes[k].fetch_add(1);
          out = 100 + k;
          return true;
        });
        if (!ok) failures_seen.fetch_add(1);
        else if (v != 100 + k) wrong.fetch_add(1);
      }
    });
  for (auto& th : threads) th.join();
  int bad = wrong.load();
  for (int k = 0; k < 8; ++k) if (makes[k].load() != 1) { fprintf(stderr, "key %d made %d times\n", k, makes[k].load()); ++bad; }
  if (failures_seen.load() < 1) { fprintf(stderr, "the failed make was never reported\n"); ++bad; }
  if (max_overlap.load() < 2) { fprintf(stderr, "makes never overlapped: they ran under the lock?\n"); ++bad; }
  if (cache.size() != 8) { fprintf(stderr, "%zu keys cached\n", cache.size()); ++bad; }
  if (bad) return 1;
  printf("keyed once: every key made once, %d makes overlapped\n", max_overlap.load());
  return 0;
}
