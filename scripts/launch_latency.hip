// launch_latency.hip -- development probe: host round-trip cost of one tiny kernel, by the way the
// host learns that it is done.  hipcc --offload-arch=gfx950 -O2 scripts/launch_latency.hip -o /tmp/ll
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tiny_flag(int* p, volatile unsigned* flag, unsigned v)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
  {
    p[0] += 1;
    __atomic_store_n((unsigned*)flag, v, __ATOMIC_RELEASE);   // system-scope store to pinned host memory
  }
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  unsigned* flag; hipHostMalloc((void**)&flag, 4, hipHostMallocCoherent); *flag = 0;
  hipStream_t s; hipStreamCreate(&s);
  const int reps = 2000;
  for (int mode = 0; mode < 4; ++mode)
  {
    for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(tiny, dim3(18), dim3(64), 0, s, d); hipStreamSynchronize(s); }
    const double t0 = now();
    for (int r = 0; r < reps; ++r)
    {
      if (mode == 0) { hipLaunchKernelGGL(tiny, dim3(18), dim3(64), 0, s, d); hipStreamSynchronize(s); }
      if (mode == 1) { hipLaunchKernelGGL(tiny, dim3(18), dim3(64), 0, s, d); while (hipStreamQuery(s) == hipErrorNotReady) {} }
      if (mode == 2)
      {
        const unsigned v = (unsigned)r + 1;
        hipLaunchKernelGGL(tiny_flag, dim3(18), dim3(64), 0, s, d, flag, v);
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != v) {}
      }
      if (mode == 3) { hipLaunchKernelGGL(tiny, dim3(18), dim3(64), 0, s, d); }
    }
    if (mode == 3) hipStreamSynchronize(s);
    const double t1 = now();
    const char* names[] = {"launch + hipStreamSynchronize", "launch + hipStreamQuery spin", "launch + pinned-flag spin", "launch only (async, amortised)"};
    printf("%-34s %7.2f us per round trip\n", names[mode], (t1 - t0) / reps);
  }
  return 0;
}
