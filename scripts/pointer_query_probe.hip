// pointer_query_probe.hip -- development probe: what a call pays for classifying one pointer with
// hipPointerGetAttributes (device, pinned host, pageable host memory), per query.
// hipcc --offload-arch=gfx950 -O2 scripts/pointer_query_probe.hip -o scripts/bin/pointer_query_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  void* dev = nullptr; hipMalloc(&dev, 64 << 20);
  void* pinned = nullptr; hipHostMalloc(&pinned, 1 << 20, 0);
  void* pageable = malloc(1 << 20);
  // a process with many live allocations (the runtime's lookup walks a map)
  std::vector<void*> many;
  for (int round = 0; round < 2; ++round)
  {
    const char* names[] = {"device", "device + 1 MiB offset", "pinned host", "pageable host"};
    void* ptrs[] = {dev, (char*)dev + (1 << 20), pinned, pageable};
    for (int k = 0; k < 4; ++k)
    {
      hipPointerAttribute_t at;
      const int reps = 20000;
      for (int w = 0; w < 100; ++w) { (void)hipPointerGetAttributes(&at, ptrs[k]); (void)hipGetLastError(); }
      const double t0 = now();
      int devs = 0;
      for (int r = 0; r < reps; ++r)
      {
        if (hipPointerGetAttributes(&at, ptrs[k]) == hipSuccess) devs += at.type == hipMemoryTypeDevice; else (void)hipGetLastError();
      }
      const double t1 = now();
      printf("%-24s %6zu live allocations: %7.3f us per hipPointerGetAttributes (device verdicts %d)\n", names[k], many.size() + 2, (t1 - t0) / reps, devs);
    }
    for (int i = 0; i < 2000; ++i) { void* p = nullptr; if (hipMalloc(&p, 4096) == hipSuccess) many.push_back(p); }
  }
  // does the allocator hand a freed device address back?  (the hazard a cached verdict has)
  void* a = nullptr; hipMalloc(&a, 8 << 20); hipFree(a);
  void* b = nullptr; hipMalloc(&b, 8 << 20);
  printf("hipMalloc after hipFree of the same size returns the same address: %s\n", a == b ? "yes" : "no");
  return 0;
}
