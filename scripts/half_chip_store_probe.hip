// half_chip_store_probe.hip -- development probe: what does a row-lockstep store stream reach when half of the CUs are
// held by another kernel (the situation of the exact-carry analysis: the relay kernel holds 128 CUs while the forward
// kernel streams the matrix)?  A "hog" kernel occupies H CUs (one 512-thread workgroup each, 140 KiB of LDS so that
// nothing shares its CU) and spins; beside it store-only kernels of the forward kernel's shape run on a second stream:
//   rows of `row_bytes`, one workgroup of W waves per time chunk, every wave 16 bytes per lane and row, a barrier every
//   `sync` rows, `work` dependent FMAs per row and lane in front of the store (emulating the recurrence), and B such
//   workgroups per CU (LDS sized so that exactly B fit).
// hipcc --offload-arch=gfx950 -O2 scripts/half_chip_store_probe.hip -o scripts/bin/half_chip_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2f64 __attribute__((ext_vector_type(2)));

// holds its CU for `ticks` of the 100 MHz constant clock without touching memory (a poll of a host flag by every hog
// wave saturates the fabric and throttles everything else -- the first version of this probe measured that, not the CUs)
__global__ void hog_kernel(unsigned long long ticks, unsigned long long* cycles)
{
  extern __shared__ char lds[];
  lds[threadIdx.x] = 1;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) cycles[blockIdx.x] = wall_clock64() - t0 + lds[0];
}

__global__ __launch_bounds__(1024) void store_rows_kernel(v2f64* dst, size_t rows, unsigned row_slots, unsigned chunk_len, unsigned sync,
                                                          unsigned work, unsigned slots_per_thread)
{
  extern __shared__ char lds[];
  if (threadIdx.x == 0) lds[0] = 0;
  const size_t t0 = (size_t)blockIdx.x * chunk_len;
  const size_t t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  v2f64 v; v.x = (double)threadIdx.x; v.y = 2.0;
  unsigned since = 0;
  for (size_t t = t0; t < t1; ++t)
  {
    float a = (float)v.x;
    for (unsigned i = 0; i < work; ++i) a = a * 1.0001f + 0.5f;
    v.x = (double)a;
    v2f64* p = dst + t * row_slots + threadIdx.x;
    for (unsigned s = 0; s < slots_per_thread; ++s) p[(size_t)s * blockDim.x] = v;
    if (sync && ++since == sync) { __syncthreads(); since = 0; }
  }
}

static float run_store(hipStream_t s, v2f64* dst, size_t bytes, unsigned row_bytes, unsigned waves, unsigned chunk_len, unsigned sync, unsigned work,
                       unsigned per_cu, int reps)
{
  const unsigned row_slots = row_bytes / 16;
  const size_t rows = bytes / row_bytes;
  const unsigned threads = waves * 64;
  const unsigned spt = row_slots / threads;
  const unsigned chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
  const size_t lds = per_cu == 1 ? 96 * 1024 : (per_cu == 2 ? 56 * 1024 : 36 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(store_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(store_rows_kernel, dim3(chunks), dim3(threads), lds, s, dst, rows, row_slots, chunk_len, sync, work, spt);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(store_rows_kernel, dim3(chunks), dim3(threads), lds, s, dst, rows, row_slots, chunk_len, sync, work, spt);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main()
{
  const size_t bytes = (size_t)8 << 30;
  v2f64* dst; hipMalloc(&dst, bytes);
  int* stop; hipHostMalloc((void**)&stop, 4, hipHostMallocCoherent);
  unsigned long long* cyc; hipMalloc(&cyc, 4096 * 8);
  hipStream_t sh, ss;
  hipStreamCreateWithFlags(&sh, hipStreamNonBlocking); hipStreamCreateWithFlags(&ss, hipStreamNonBlocking);
  hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  struct Cfg { unsigned row_bytes, waves, chunk, sync, work, per_cu; };
  const Cfg cfgs[] = {
      {32768, 16, 128, 4, 0, 1},   {32768, 16, 128, 4, 200, 1}, {32768, 16, 128, 4, 400, 1},
      {32768, 8, 128, 4, 0, 2},    {32768, 8, 128, 4, 200, 2},  {32768, 16, 128, 0, 0, 1},
      {32768, 16, 128, 1, 0, 1},   {32768, 4, 128, 4, 0, 3},    {16384, 16, 192, 8, 0, 1},  {16384, 8, 192, 8, 0, 2},
      {8192, 8, 256, 4, 0, 2},     {8192, 8, 256, 4, 0, 1},
  };
  for (unsigned hogs : {0u, 64u, 128u, 192u})
  {
    if (hogs) hipLaunchKernelGGL(hog_kernel, dim3(hogs), dim3(512), 140 * 1024, sh, (unsigned long long)(2.5 * 100e6), cyc);   // 2.5 s
    for (const Cfg& c : cfgs)
    {
      const float ms = run_store(ss, dst, bytes, c.row_bytes, c.waves, c.chunk, c.sync, c.work, c.per_cu, 3);
      printf("hog CUs %3u | rows of %5u B, %2u waves/WG, %u WG/CU, chunk %3u, barrier every %u rows, %3u FMAs/row: %7.0f GB/s\n", hogs, c.row_bytes, c.waves,
             c.per_cu, c.chunk, c.sync, c.work, bytes / (ms * 1e-3) / 1e9);
    }
    hipStreamSynchronize(sh);
  }
  return 0;
}
