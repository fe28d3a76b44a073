// The host's copy out of pinned memory the DMA engine has just written: glibc memcpy against streaming (non-temporal)
// stores, one thread and two, 2 MiB pieces out of a 64 MiB transfer (so that nothing is in the CPU's caches).
// build: hipcc --offload-arch=gfx950 -O2 -o sdft_amd/lib/probe/host_memcpy_probe scripts/host_memcpy_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void stream_copy(char* dst, const char* src, size_t bytes)
{
  // head up to 16-byte alignment of dst
  size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
  if (head > bytes) head = bytes;
  memcpy(dst, src, head); dst += head; src += head; bytes -= head;
  const size_t blocks = bytes / 64;
  for (size_t i = 0; i < blocks; ++i)
  {
    const __m128i a = _mm_loadu_si128((const __m128i*)(src) + 0), b = _mm_loadu_si128((const __m128i*)(src) + 1);
    const __m128i c = _mm_loadu_si128((const __m128i*)(src) + 2), d = _mm_loadu_si128((const __m128i*)(src) + 3);
    _mm_stream_si128((__m128i*)(dst) + 0, a); _mm_stream_si128((__m128i*)(dst) + 1, b);
    _mm_stream_si128((__m128i*)(dst) + 2, c); _mm_stream_si128((__m128i*)(dst) + 3, d);
    src += 64; dst += 64;
  }
  _mm_sfence();
  memcpy(dst, src, bytes - blocks * 64);
}

int main()
{
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  const size_t total = 64u << 20, piece = 2u << 20;
  char *d = nullptr, *pin = nullptr;
  CHECK(hipMalloc((void**)&d, total));
  CHECK(hipMemset(d, 0x5a, total));
  CHECK(hipHostMalloc((void**)&pin, total, hipHostMallocDefault));
  char* a = (char*)aligned_alloc(4096, total + 64);
  memset(a, 1, total + 64);
  for (int variant = 0; variant < 6; ++variant)
  {
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep)
    {
      CHECK(hipMemcpyAsync(pin, d, total, hipMemcpyDeviceToHost, s));       // the DMA engine writes the pinned memory: cold for the CPU
      CHECK(hipStreamSynchronize(s));
      const double t0 = now();
      for (size_t o = 0; o < total; o += piece)
      {
        char* dst = a + 8 + o; const char* src = pin + o;
        switch (variant)
        {
          case 0: memcpy(dst, src, piece); break;
          case 1: stream_copy(dst, src, piece); break;
          case 2: { std::thread t([&] { memcpy(dst + piece / 2, src + piece / 2, piece / 2); }); memcpy(dst, src, piece / 2); t.join(); } break;
          case 3: { std::thread t([&] { stream_copy(dst + piece / 2, src + piece / 2, piece / 2); }); stream_copy(dst, src, piece / 2); t.join(); } break;
          case 4: memcpy(a + o, src, piece); break;                          // aligned destination
          case 5: stream_copy(a + o, src, piece); break;
        }
      }
      best = std::min(best, now() - t0);
    }
    const char* names[] = {"memcpy", "streaming stores", "memcpy, two threads (spawned per piece)", "streaming stores, two threads (spawned per piece)",
                           "memcpy, aligned destination", "streaming stores, aligned destination"};
    printf("%-52s %8.1f us per 64 MiB = %5.1f GB/s  (%5.1f us per 2 MiB piece)\n", names[variant], best, total / best / 1e3, best / (total / piece));
  }
  return 0;
}
