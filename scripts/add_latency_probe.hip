// Dependent-add latency of one wave (the ordered walk of fused_exact = 1 is such a chain): cycles per v_add_f64 / v_add_f32
// with 8 or 64 active lanes, operands from registers or from LDS (ds_read_b128, with and without prefetch).
//   hipcc --offload-arch=gfx950 -O3 scripts/add_latency_probe.hip -o /tmp/add_latency_probe && /tmp/add_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ void chain_regs(T* out, unsigned long long* cyc, int lanes, T seed)
{
  T s = seed; T a = seed * (T)1.5, b = seed * (T)0.25;
  if ((int)threadIdx.x >= lanes) return;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 256; ++i)
  {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s += a; s += b; }
    asm volatile("" : "+v"(s));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <typename T, bool PREFETCH> __global__ void chain_lds(T* out, unsigned long long* cyc, int lanes, int stride)
{
  extern __shared__ __align__(16) unsigned char raw[];
  T* img = reinterpret_cast<T*>(raw);
  typedef T tvec __attribute__((ext_vector_type(16 / sizeof(T))));
  constexpr int NV = 16 / (int)sizeof(T);
  for (int i = threadIdx.x; i < 8 * stride; i += blockDim.x) img[i] = (T)(i % 7) * (T)0.125;
  __syncthreads();
  if ((int)threadIdx.x >= lanes) return;
  const T* tr = img + (threadIdx.x % 8) * stride;
  T sum = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (!PREFETCH)
  {
    for (int k0 = 0; k0 < 1024; k0 += 8 * NV)
    {
      tvec tv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) tv[i] = *reinterpret_cast<const tvec*>(tr + k0 + i * NV);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < NV; ++e) sum += tv[i][e];
    }
  }
  else
  {
    tvec ta[8], tb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ta[i] = *reinterpret_cast<const tvec*>(tr + i * NV);
    for (int k0 = 0; k0 < 1024; k0 += 16 * NV)
    {
#pragma unroll
      for (int i = 0; i < 8; ++i) tb[i] = *reinterpret_cast<const tvec*>(tr + k0 + (8 + i) * NV);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < NV; ++e) sum += ta[i][e];
      const int kn = (k0 + 16 * NV < 1024) ? k0 + 16 * NV : k0;
#pragma unroll
      for (int i = 0; i < 8; ++i) ta[i] = *reinterpret_cast<const tvec*>(tr + kn + i * NV);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < NV; ++e) sum += tb[i][e];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = sum;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void chain_pk(f2* out, unsigned long long* cyc, f2 seed)
{
  f2 s = seed; f2 a = seed * 1.5f, b = seed * 0.25f;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 256; ++i)
  {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s += a; s += b; }
    asm volatile("" : "+v"(s));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
  {
    f2* o2; unsigned long long* c2;
    hipMalloc(&o2, 64 * 8); hipHostMalloc(&c2, 8);
    f2 sd; sd.x = 1.0f; sd.y = 2.0f;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chain_pk, dim3(1), dim3(64), 0, 0, o2, c2, sd); hipDeviceSynchronize(); }
    printf("registers, packed float2 (v_pk_add_f32), 64 lanes: raw %llu per 4096 adds\n", *c2);
  }
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 8); hipHostMalloc(&cyc, 8);
  for (int lanes : {8, 64})
  {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chain_regs<double>, dim3(1), dim3(64), 0, 0, out, cyc, lanes, 1.0); hipDeviceSynchronize(); }
    printf("registers, double, %2d lanes: %.2f cycles (s_memtime counts, 100 MHz -> x clock/100 MHz) per add over 4096 adds: raw %llu\n", lanes, (double)*cyc / 4096, *cyc);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chain_regs<float>, dim3(1), dim3(64), 0, 0, (float*)out, cyc, lanes, 1.0f); hipDeviceSynchronize(); }
    printf("registers, float,  %2d lanes: raw %llu per 4096 adds\n", lanes, *cyc);
  }
  for (int stride : {1024, 1026, 1040})
  {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((chain_lds<double, false>), dim3(1), dim3(1024), 8 * stride * 8, 0, out, cyc, 8, stride); hipDeviceSynchronize(); }
    printf("LDS image stride %d doubles, 8 lanes, batch of 8 reads then 16 adds: raw %llu per 1024 adds\n", stride, *cyc);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((chain_lds<double, true>), dim3(1), dim3(1024), 8 * stride * 8, 0, out, cyc, 8, stride); hipDeviceSynchronize(); }
    printf("LDS image stride %d doubles, 8 lanes, prefetched: raw %llu per 1024 adds\n", stride, *cyc);
  }
  return 0;
}
