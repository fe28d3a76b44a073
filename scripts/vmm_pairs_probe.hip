// vmm_pairs_probe.hip -- development probe (round 5): can an allocator MAKE a matrix fast?  Physical chunks of half a matrix each (hipMemCreate), pairs of them mapped
// next to each other into one virtual range (hipMemMap), the row-lockstep store-only kernel of the analysis' shape on the mapped matrix.  If the chunks of a pair
// lie in different stretches of device memory the matrix should take 7.1 TB/s, else 5.8 (profiles/r05_split_matrix.txt).
// hipcc --offload-arch=gfx950 -O2 -w scripts/vmm_pairs_probe.hip -o scripts/bin/vmm_pairs_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2f64 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void store_rows_kernel(v2f64* dst, size_t rows, unsigned chunk_len)
{
  const unsigned R = 8, q = gridDim.x / R, r = gridDim.x % R, x = blockIdx.x % R;
  const unsigned chunk = x * q + (x < r ? x : r) + blockIdx.x / R;
  const size_t t0 = (size_t)chunk * chunk_len, t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  v2f64 v; v.x = (double)threadIdx.x; v.y = 2.0;
  unsigned since = 0;
  for (size_t t = t0; t < t1; ++t)
  {
    dst[t * 1024 + threadIdx.x] = v;
    v.x += 1.0;
    if (++since == 8) { __syncthreads(); since = 0; }
  }
}
static double rate(void* p, size_t rows)
{
  const unsigned chunk_len = 1960, chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(store_rows_kernel, dim3(chunks), dim3(1024), 0, 0, (v2f64*)p, rows, chunk_len);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 0; }
  hipEventRecord(e0, 0);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_rows_kernel, dim3(chunks), dim3(1024), 0, 0, (v2f64*)p, rows, chunk_len);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return (double)rows * 16384.0 / (ms / 2 * 1e-3) / 1e9;
}
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
  int dev = 0; CHECK(hipSetDevice(dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t rows = 1000000, bytes = rows * 16384;
  const size_t big = (size_t)1 << 30;                        // (whole GiB: the driver can map large fragments)
  const size_t half = ((bytes / 2 + big - 1) / big) * big;
  printf("granularity %zu B; matrix %zu B = two chunks of %zu B\n", gran, bytes, half);
  const int K = 12;
  std::vector<hipMemGenericAllocationHandle_t> h(K);
  int made = 0;
  for (int i = 0; i < K; ++i) { if (hipMemCreate(&h[i], half, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; } ++made; }
  printf("%d physical chunks\n", made);
  void* va = nullptr;
  CHECK(hipMemAddressReserve(&va, 2 * half, big, nullptr, 0));
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int i = 0; i < made; ++i)
    for (int j = 0; j < made; ++j)
    {
      if (i == j || (i != 0 && !(i == j - 1 && i % 3 == 1))) continue;          // pairs (0, j) and a few neighbours
      CHECK(hipMemMap(va, half, 0, h[i], 0));
      CHECK(hipMemMap((char*)va + half, half, 0, h[j], 0));
      CHECK(hipMemSetAccess(va, 2 * half, &acc, 1));
      printf("chunks %2d + %2d: %.0f GB/s\n", i, j, rate(va, rows)); fflush(stdout);
      CHECK(hipMemUnmap(va, half));
      CHECK(hipMemUnmap((char*)va + half, half));
    }
  for (int i = 0; i < made; ++i) hipMemRelease(h[i]);
  hipMemAddressFree(va, 2 * half);
  return 0;
}
