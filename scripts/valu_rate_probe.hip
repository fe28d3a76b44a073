// valu_rate_probe.hip -- development probe: cycles a SIMD needs per vector instruction with 1, 2 and 4 waves per SIMD
// issuing independent instructions: v_mul_f32 / v_add_f32 against v_pk_mul_f32 / v_pk_add_f32 (two results per lane) and
// the fp64 forms.  Decides whether the FD float kernels should pack bin pairs or stay scalar.
// hipcc --offload-arch=gfx950 -O2 scripts/valu_rate_probe.hip -o scripts/bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int KIND> __global__ void rate_kernel(unsigned long long* out, float seed, int iters)
{
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3, b4 = seed + 4, b5 = seed + 5, b6 = seed + 6, b7 = seed + 7;
  const float m = 1.0000001f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i)
  {
    if constexpr (KIND == 0)       // 16 independent v_mul_f32
      asm volatile(REP8("v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n"
                        "v_mul_f32 %8, %8, %16\n v_mul_f32 %9, %9, %16\n v_mul_f32 %10, %10, %16\n v_mul_f32 %11, %11, %16\n v_mul_f32 %12, %12, %16\n v_mul_f32 %13, %13, %16\n v_mul_f32 %14, %14, %16\n v_mul_f32 %15, %15, %16\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(m));
    else if constexpr (KIND == 1)  // 8 independent v_pk_mul_f32 (register pairs a0a1 ... b6b7) = the same 16 results
    {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {b4, b5}, p7 = {b6, b7}, mm = {m, m};
      asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mm));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y; b0 = p4.x; b1 = p4.y; b2 = p5.x; b3 = p5.y; b4 = p6.x; b5 = p6.y; b6 = p7.x; b7 = p7.y;
    }
    else if constexpr (KIND == 2)  // 16 independent v_add_f32
      asm volatile(REP8("v_add_f32 %0, %0, %16\n v_add_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_add_f32 %3, %3, %16\n v_add_f32 %4, %4, %16\n v_add_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_add_f32 %7, %7, %16\n"
                        "v_add_f32 %8, %8, %16\n v_add_f32 %9, %9, %16\n v_add_f32 %10, %10, %16\n v_add_f32 %11, %11, %16\n v_add_f32 %12, %12, %16\n v_add_f32 %13, %13, %16\n v_add_f32 %14, %14, %16\n v_add_f32 %15, %15, %16\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(m));
    else if constexpr (KIND == 3)  // 8 independent v_pk_add_f32
    {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {b4, b5}, p7 = {b6, b7}, mm = {m, m};
      asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mm));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y; b0 = p4.x; b1 = p4.y; b2 = p5.x; b3 = p5.y; b4 = p6.x; b5 = p6.y; b6 = p7.x; b7 = p7.y;
    }
    else if constexpr (KIND == 4)  // 8 independent v_mul_f64
    {
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7; const double dm = 1.0000001;
      asm volatile(REP8("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n")
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm));
      a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3; a4 = (float)d4; a5 = (float)d5; a6 = (float)d6; a7 = (float)d7;
    }
    else if constexpr (KIND == 5)  // 16 v_mov_b32_dpp wave_shr:1
      asm volatile(REP8("v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %9 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %10 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %11 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %13 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %14 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %15 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)s; }
}

template <int KIND> static void run(const char* name, int per_iter)
{
  unsigned long long* out; hipMalloc(&out, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {4, 8, 16, 32})
  {
    const int iters = 4000;
    // `waves` waves per CU = waves / 4 per SIMD: 8 rounds of (256 CUs x waves) so that placement evens out; the aggregate
    // rate comes from the events around the launch, the per-wave cost from the cycle counter of workgroup 0
    const int threads = waves * 64 > 1024 ? 1024 : waves * 64;
    const int blocks = 256 * (waves * 64 / threads);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double per_wave_instr = (double)h[0] / ((double)iters * per_iter);
    const double wave_instr = (double)blocks * (threads / 64) * iters * per_iter;
    printf("%-16s %2d waves/SIMD: %6.2f cycles per instruction per wave (workgroup 0) | chip: %7.1f G wave-instructions/s = %5.2f per CU and cycle at 2.4 GHz (%.3f ms)\n",
           name, waves / 4, per_wave_instr, wave_instr / (ms * 1e-3) / 1e9, wave_instr / (ms * 1e-3) / 256 / 2.4e9, ms);
  }
  hipFree(out);
}

int main()
{
  run<0>("v_mul_f32", 128); run<1>("v_pk_mul_f32", 64); run<2>("v_add_f32", 128); run<3>("v_pk_add_f32", 64); run<4>("v_mul_f64", 64); run<5>("v_mov_b32_dpp", 64);
  return 0;
}
