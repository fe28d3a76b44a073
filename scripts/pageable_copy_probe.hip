// Own pinned staging, both directions, by piece size and number of slots (1.6 MB = one hop of the reference's driver).
// build: hipcc --offload-arch=gfx950 -O2 -o sdft_amd/lib/probe/pageable_copy_probe scripts/pageable_copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  const size_t cap = 8u << 20;
  char *d = nullptr, *pin = nullptr;
  CHECK(hipMalloc((void**)&d, 64u << 20));
  CHECK(hipMemset(d, 0x5a, 64u << 20));
  CHECK(hipHostMalloc((void**)&pin, cap, hipHostMallocDefault));
  hipEvent_t ev[32];
  for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (size_t bytes : {(size_t)400000, (size_t)1600000, (size_t)16000000})
  {
    char* a = (char*)aligned_alloc(4096, bytes);
    memset(a, 1, bytes);
    auto bench = [&](const char* name, auto&& fn) {
      for (int i = 0; i < 3; ++i) fn();
      double best = 1e30, sum = 0; const int reps = 30;
      for (int i = 0; i < reps; ++i) { const double t0 = now(); fn(); const double t = now() - t0; best = std::min(best, t); sum += t; }
      printf("%9zu B  %-52s best %8.1f us  mean %8.1f us\n", bytes, name, best, sum / reps); fflush(stdout);
    };
    bench("D2H runtime (remembered pin)", [&] { (void)hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
    bench("H2D runtime (remembered pin)", [&] { (void)hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); });
    for (size_t piece : {(size_t)128 << 10, (size_t)256 << 10, (size_t)512 << 10, (size_t)1 << 20})
      for (int slots : {2, 4, 8})
      {
        if (piece * slots > cap) continue;
        char name[96];
        snprintf(name, sizeof name, "D2H own, %4zu KiB x %d slots, submit ahead", piece >> 10, slots);
        bench(name, [&] {
          const size_t np = (bytes + piece - 1) / piece;
          size_t sub = 0;
          for (size_t i = 0; i < np; ++i)
          {
            while (sub < np && sub < i + slots) { const size_t o = sub * piece, len = std::min(piece, bytes - o); (void)hipMemcpyAsync(pin + (sub % slots) * piece, d + o, len, hipMemcpyDeviceToHost, s); (void)hipEventRecord(ev[sub % slots], s); ++sub; }
            (void)hipEventSynchronize(ev[i % slots]);
            const size_t o = i * piece; memcpy(a + o, pin + (i % slots) * piece, std::min(piece, bytes - o));
          }
        });
        snprintf(name, sizeof name, "H2D own, %4zu KiB x %d slots", piece >> 10, slots);
        bench(name, [&] {
          const size_t np = (bytes + piece - 1) / piece;
          for (size_t i = 0; i < np; ++i)
          {
            if (i >= (size_t)slots) (void)hipEventSynchronize(ev[i % slots]);
            const size_t o = i * piece, len = std::min(piece, bytes - o);
            memcpy(pin + (i % slots) * piece, a + o, len);
            (void)hipMemcpyAsync(d + o, pin + (i % slots) * piece, len, hipMemcpyHostToDevice, s); (void)hipEventRecord(ev[i % slots], s);
          }
          (void)hipStreamSynchronize(s);
        });
      }
    bench("memcpy pageable -> pinned, whole", [&] { memcpy(pin, a, std::min(bytes, cap)); });
    bench("D2H into pinned, whole, then memcpy", [&] { const size_t b = std::min(bytes, cap); (void)hipMemcpyAsync(pin, d, b, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); memcpy(a, pin, b); });
    bench("D2H into pinned, whole, no memcpy", [&] { const size_t b = std::min(bytes, cap); (void)hipMemcpyAsync(pin, d, b, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
    free(a);
  }
  return 0;
}
