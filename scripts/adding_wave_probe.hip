// adding_wave_probe.hip -- development probe (round 6): what does ONE wave pay per term when lane j adds the terms of LDS row slot j in order
// (the adding wave of inverse_rows_ordered_kernel)?  One workgroup, one wave, the slots filled once; cycles per term (s_memtime, 100 MHz -> x 24)
// by the width of the LDS read, the number of lanes that add, the stride between slots and the length of a stage.
// hipcc --offload-arch=gfx950 -O3 -w scripts/adding_wave_probe.hip -o scripts/bin/adding_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

template <typename FD, int SV, int WIDE>
__global__ __launch_bounds__(64) void probe(unsigned lanes, unsigned stride, unsigned terms, unsigned rounds, unsigned long long* out, FD* sink)
{
  extern __shared__ __align__(16) unsigned char lds[];
  using TV = typename std::conditional<sizeof(FD) == 8, v2f64, v4f32>::type;
  constexpr int PER = 16 / (int)sizeof(FD);
  const int lane = threadIdx.x;
  for (unsigned i = lane; i < (lanes * stride + 512) / 4; i += 64) reinterpret_cast<float*>(lds)[i] = 1.0f + (float)(i & 7) * 0.125f;
  __syncthreads();
  const unsigned char* my = lds + (size_t)(lane < (int)lanes ? lane : 0) * stride;
  FD sum = (FD)0;
  const bool active = lane < (int)lanes;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (unsigned r = 0; r < rounds; ++r)
  {
    if (active)
    {
      if constexpr (WIDE == 1)
      {
        TV b0[SV], b1[SV];
        const TV* t = reinterpret_cast<const TV*>(my);
#pragma unroll
        for (int c = 0; c < SV; ++c) b0[c] = t[c];
        const unsigned stages = terms / (SV * PER);
        for (unsigned st = 0; st < stages; st += 2)
        {
          const TV* u = reinterpret_cast<const TV*>(my + (size_t)(st + 1) * SV * 16);
#pragma unroll
          for (int c = 0; c < SV; ++c) b1[c] = u[c];
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SV == 8) asm volatile("" : : "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b0[4]), "v"(b0[5]), "v"(b0[6]), "v"(b0[7]) : "memory");
          else asm volatile("" : : "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]) : "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < SV; ++c)
#pragma unroll
            for (int e = 0; e < PER; ++e) sum += b0[c][e];
          __builtin_amdgcn_sched_barrier(0);
          const TV* w = reinterpret_cast<const TV*>(my + (size_t)(st + 2 < stages ? st + 2 : 0) * SV * 16);
#pragma unroll
          for (int c = 0; c < SV; ++c) b0[c] = w[c];
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SV == 8) asm volatile("" : : "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]), "v"(b1[4]), "v"(b1[5]), "v"(b1[6]), "v"(b1[7]) : "memory");
          else asm volatile("" : : "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]) : "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < SV; ++c)
#pragma unroll
            for (int e = 0; e < PER; ++e) sum += b1[c][e];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      else
      {
        // no LDS at all: the chain of dependent additions alone
        for (unsigned k = 0; k < terms; k += 16)
        {
#pragma unroll
          for (int e = 0; e < 16; ++e) { sum += (FD)1.25; asm volatile("" : "+v"(sum)); }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
  if (sum == (FD)12345) sink[lane] = sum;
}

template <typename FD, int SV, int WIDE>
static void run(const char* name, unsigned lanes, unsigned stride, unsigned terms)
{
  unsigned long long* d_out; FD* d_sink;
  hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * sizeof(FD));
  auto kern = probe<FD, SV, WIDE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const unsigned rounds = 64;
  const size_t lds = (size_t)lanes * stride + 1024;
  unsigned long long best = ~0ull;
  for (int it = 0; it < 5; ++it)
  {
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), lds, 0, lanes, stride, terms, rounds, d_out, d_sink);
    unsigned long long h = 0; hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    if (h < best) best = h;
  }
  // s_memrealtime-like counter of __builtin_readcyclecounter: shader clock cycles on gfx9
  printf("%-64s %7.2f cycles per term (%llu for %u x %u terms)\n", name, (double)best / ((double)rounds * terms), best, rounds, terms);
  hipFree(d_out); hipFree(d_sink);
}

int main()
{
  run<double, 8, 0>("double, no LDS: the dependent additions alone", 16, 8208, 1024);
  run<float, 8, 0>("float, no LDS: the dependent additions alone", 16, 4112, 1024);
  run<double, 8, 1>("double, 16 lanes, slots 8208 B apart, stages of 16 terms", 16, 8208, 1024);
  run<double, 4, 1>("double, 16 lanes, slots 8208 B apart, stages of 8 terms", 16, 8208, 1024);
  run<double, 8, 1>("double, 8 lanes, slots 16400 B apart, stages of 16 terms", 8, 16400, 2048);
  run<double, 8, 1>("double, 1 lane", 1, 8208, 1024);
  run<double, 8, 1>("double, 64 lanes, slots 2064 B apart (256 terms)", 64, 2064, 256);
  run<double, 8, 1>("double, 16 lanes, slots 8192 B apart (no padding)", 16, 8192, 1024);
  run<double, 8, 1>("double, 16 lanes, slots 8256 B apart", 16, 8256, 1024);
  run<float, 8, 1>("float, 32 lanes, slots 4112 B apart, stages of 32 terms", 32, 4112, 1024);
  run<float, 8, 1>("float, 16 lanes, slots 8208 B apart, stages of 32 terms", 16, 8208, 2048);
  run<float, 8, 1>("float, 8 lanes, slots 16400 B apart, stages of 32 terms", 8, 16400, 4096);
  run<float, 4, 1>("float, 8 lanes, slots 16400 B apart, stages of 16 terms", 8, 16400, 4096);
  return 0;
}
