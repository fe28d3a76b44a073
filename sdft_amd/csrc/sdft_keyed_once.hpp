// sdft_keyed_once.hpp -- "make each key's value once, outside the lock": the cache behind the run-time compilation of a host's
// own spectral operation (sdft_common.hip: one compiled kernel per (device, statements, instantiation); a compilation takes a
// second, so it must not run under the cache's lock, and two threads that want the same key must not both compile).
// No HIP dependency: tests/cpp/keyed_once_test.cpp runs it under ThreadSanitizer in the `-m "not gpu"` suite.

#pragma once

#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <string>

namespace sdfthip {

template <typename Value>
class KeyedOnce
{
 public:
  // the value of `key`: from the cache, or made by make(value&) -> bool on THIS thread with the lock released; a thread that
  // asks for a key another thread is making waits for that result instead of starting its own.  A failed make is not cached
  // (the next caller tries again); returns false then.
  template <typename Make>
  bool get(const std::string& key, Value& out, Make&& make)
  {
    {
      std::unique_lock<std::mutex> lock(mu_);
      for (;;)
      {
        auto it = done_.find(key);
        if (it != done_.end()) { out = it->second; return true; }
        if (!in_flight_.count(key)) break;
        cv_.wait(lock);
      }
      in_flight_.insert(key);
    }
    Value v{};
    bool ok = false;
    try { ok = make(v); } catch (...) { ok = false; }
    {
      std::lock_guard<std::mutex> lock(mu_);
      in_flight_.erase(key);
      if (ok) done_[key] = v;
    }
    cv_.notify_all();
    if (ok) out = v;
    return ok;
  }
  size_t size() const { std::lock_guard<std::mutex> lock(mu_); return done_.size(); }

 private:
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::map<std::string, Value> done_;
  std::set<std::string> in_flight_;
};

}  // namespace sdfthip
