// load_forms_probe.hip -- development probe (round 5): how fast can a kernel READ a matrix of the synthesis' size, by the
// form of the load?  plain global_load_dwordx4, the same nontemporal, LDS-DMA (global_load_lds_dwordx4: memory -> LDS
// without a register in between), each as a grid-stride sweep over 16.4 GB with `depth` 16-byte loads in flight per lane
// and as whole rows read in step by one workgroup per chunk of rows (the shape the analysis writes in).
// hipcc --offload-arch=gfx950 -O2 scripts/load_forms_probe.hip -o scripts/bin/load_forms_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int FORM, int DEPTH> __global__ __launch_bounds__(1024) void sweep_kernel(const v4f* __restrict__ src, size_t slots, float* sink)
{
  extern __shared__ v4f lds[];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  for (; i + (DEPTH - 1) * stride < slots; i += DEPTH * stride)
  {
    if constexpr (FORM == 2)
    {
      // one LDS-DMA per wave-instruction: 64 lanes x 16 B land at M0-base + lane * 16 (the LDS address is wave-uniform)
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        __builtin_amdgcn_global_load_lds(src + i + d * stride, (__attribute__((address_space(3))) void*)(lds + (threadIdx.x & ~63u) + d * blockDim.x), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else
    {
      v4f v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = FORM == 1 ? __builtin_nontemporal_load(src + i + d * stride) : src[i + d * stride];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  if constexpr (FORM == 2) { __syncthreads(); acc = lds[threadIdx.x]; }
  if (acc.x == 1.2345e-30f) sink[0] = acc.y + acc.z + acc.w;
}

// whole rows in step: workgroup b reads rows [b * chunk_len, (b + 1) * chunk_len), `row_slots` 16-byte slots each
template <int FORM, int DEPTH> __global__ __launch_bounds__(1024) void rows_kernel(const v4f* __restrict__ src, size_t rows, unsigned row_slots, unsigned chunk_len, float* sink)
{
  extern __shared__ v4f lds[];
  const size_t t0 = (size_t)blockIdx.x * chunk_len, t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  v4f acc = {0, 0, 0, 0};
  for (size_t t = t0; t + DEPTH <= t1; t += DEPTH)
  {
    if constexpr (FORM == 2)
    {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        for (unsigned s = threadIdx.x; s < row_slots; s += blockDim.x)
          __builtin_amdgcn_global_load_lds(src + (t + d) * row_slots + s, (__attribute__((address_space(3))) void*)(lds + (s & ~63u) + d * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else
    {
      v4f v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        for (unsigned s = threadIdx.x; s < row_slots; s += blockDim.x)
          v[d] = FORM == 1 ? __builtin_nontemporal_load(src + (t + d) * row_slots + s) : src[(t + d) * row_slots + s];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  if constexpr (FORM == 2) { __syncthreads(); acc = lds[threadIdx.x]; }
  if (acc.x == 1.2345e-30f) sink[0] = acc.y + acc.z + acc.w;
}

template <typename L> static double timed(L launch, int reps)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1e9; }
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}

int main(int argc, char** argv)
{
  const size_t bytes = (size_t)16384 * 1000000;             // the n = 1e6, m = 1024, FD double matrix
  v4f* buf; float* sink;
  if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { printf("no memory\n"); return 1; }
  hipMemset(buf, 0, bytes);
  const size_t slots = bytes / 16;
  const char* names[3] = {"plain", "nontemporal", "LDS-DMA"};
  for (int rep = 0; rep < 2; ++rep)
  {
    for (unsigned wg_per_cu : {2u, 4u, 8u})
      for (unsigned threads : {256u, 512u, 1024u})
      {
        if (wg_per_cu * threads > 2048) continue;
        const unsigned blocks = 256 * wg_per_cu;
        const size_t lds4 = (size_t)threads * 16 * 4, lds8 = (size_t)threads * 16 * 8;          // depth x threads slots of 16 bytes
        hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        double ms[3][2];
        ms[0][0] = timed([&] { hipLaunchKernelGGL((sweep_kernel<0, 4>), dim3(blocks), dim3(threads), 0, 0, buf, slots, sink); }, 3);
        ms[0][1] = timed([&] { hipLaunchKernelGGL((sweep_kernel<0, 8>), dim3(blocks), dim3(threads), 0, 0, buf, slots, sink); }, 3);
        ms[1][0] = timed([&] { hipLaunchKernelGGL((sweep_kernel<1, 4>), dim3(blocks), dim3(threads), 0, 0, buf, slots, sink); }, 3);
        ms[1][1] = timed([&] { hipLaunchKernelGGL((sweep_kernel<1, 8>), dim3(blocks), dim3(threads), 0, 0, buf, slots, sink); }, 3);
        ms[2][0] = timed([&] { hipLaunchKernelGGL((sweep_kernel<2, 4>), dim3(blocks), dim3(threads), lds4, 0, buf, slots, sink); }, 3);
        ms[2][1] = timed([&] { hipLaunchKernelGGL((sweep_kernel<2, 8>), dim3(blocks), dim3(threads), lds8, 0, buf, slots, sink); }, 3);
        printf("sweep  %u workgroups of %4u threads per CU:", wg_per_cu, threads);
        for (int f = 0; f < 3; ++f) printf("   %s depth 4 / 8: %5.0f / %5.0f GB/s", names[f], bytes / ms[f][0] / 1e6, bytes / ms[f][1] / 1e6);
        printf("\n"); fflush(stdout);
      }
    // rows of 16 KiB read in step by 1024-thread workgroups (one slot per thread and row), chunks of 1960 rows (511 workgroups) and 980
    for (unsigned chunk_len : {1960u, 980u, 490u})
    {
      const size_t rows = 1000000;
      const unsigned blocks = (unsigned)((rows + chunk_len - 1) / chunk_len);
      double ms[3];
      ms[0] = timed([&] { hipLaunchKernelGGL((rows_kernel<0, 4>), dim3(blocks), dim3(1024), 0, 0, buf, rows, 1024u, chunk_len, sink); }, 3);
      ms[1] = timed([&] { hipLaunchKernelGGL((rows_kernel<1, 4>), dim3(blocks), dim3(1024), 0, 0, buf, rows, 1024u, chunk_len, sink); }, 3);
      ms[2] = timed([&] { hipLaunchKernelGGL((rows_kernel<2, 4>), dim3(blocks), dim3(1024), 1024 * 16 * 4, 0, buf, rows, 1024u, chunk_len, sink); }, 3);
      printf("rows in step, chunks of %4u rows (%u workgroups), 4 rows in flight:", chunk_len, blocks);
      for (int f = 0; f < 3; ++f) printf("   %s %5.0f GB/s", names[f], bytes / ms[f] / 1e6);
      printf("\n"); fflush(stdout);
    }
  }
  return 0;
}
