// wait_value_probe.hip -- development probe: can a running kernel release work queued on ANOTHER stream
// through hipStreamWaitValue32 on signal memory (no kernel boundary, no host round trip)?
// hipcc --offload-arch=gfx950 -O2 -w scripts/wait_value_probe.hip -o scripts/bin/wv
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void producer(unsigned* sig, unsigned long long* stamps, int stages, long spin)
{
  for (int s = 1; s <= stages; ++s)
  {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) {}          // ~spin * 10 ns of "work"
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
      __atomic_store_n(sig, (unsigned)s, __ATOMIC_RELEASE);                    // system-scope store
      stamps[s] = __builtin_amdgcn_s_memrealtime();
    }
  }
}
__global__ void consumer(unsigned long long* stamps, int s) { if (threadIdx.x == 0) stamps[16 + s] = __builtin_amdgcn_s_memrealtime(); }

int main()
{
  int can = 0;
  hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  unsigned* sig = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 0;
  hipMemset(sig, 0, 8);
  unsigned long long* stamps; hipMalloc(&stamps, 64 * 8); hipMemset(stamps, 0, 64 * 8);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  const int stages = 4;
  // consumers are queued first, each behind a wait on the producer's progress word
  for (int s = 1; s <= stages; ++s)
  {
    e = hipStreamWaitValue32(b, sig, (unsigned)s, hipStreamWaitValueGte, 0xFFFFFFFF);
    if (e != hipSuccess) { printf("hipStreamWaitValue32 -> %s\n", hipGetErrorString(e)); return 0; }
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps, s);
  }
  hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, sig, stamps, stages, 20000L);   // 4 stages x 200 us
  const auto t0 = std::chrono::steady_clock::now();
  while (hipStreamQuery(b) == hipErrorNotReady)
  {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) { printf("TIMEOUT: consumers never released\n"); return 1; }
  }
  hipDeviceSynchronize();
  unsigned long long h[64]; hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
  for (int s = 1; s <= stages; ++s)
    printf("stage %d: producer signalled at %8.1f us, consumer ran at %8.1f us  (release latency %.1f us)\n", s,
           (h[s] - h[1]) / 100.0, (h[16 + s] - h[1]) / 100.0, ((double)h[16 + s] - (double)h[s]) / 100.0);
  return 0;
}
