// dep_latency.hip -- development probe: what one wave alone on a SIMD pays per DEPENDENT vector
// instruction (the floor of any strictly ordered floating point chain), next to independent issue.
// hipcc --offload-arch=gfx950 -O2 scripts/dep_latency.hip -o scripts/bin/dep
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void probe(float* out, unsigned long long* clk, int iters, float b)
{
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
  double d0 = threadIdx.x, db = b;
  const float sb = __builtin_amdgcn_readfirstlane(b);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i)
  {
    if (MODE == 0) asm volatile(REP64("v_add_f32_e32 %0, %0, %1\n\t") : "+v"(a0) : "v"(b));                       // dependent f32 add
    if (MODE == 1) asm volatile(REP64("v_add_f32_e32 %0, %0, %4\n\tv_add_f32_e32 %1, %1, %4\n\tv_add_f32_e32 %2, %2, %4\n\tv_add_f32_e32 %3, %3, %4\n\t")
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));                                  // 4 independent chains
    if (MODE == 2) asm volatile(REP64("v_add_f64 %0, %0, %1\n\t") : "+v"(d0) : "v"(db));                            // dependent f64 add
    if (MODE == 3) asm volatile(REP64("v_mul_f32_e32 %1, %0, %2\n\tv_add_f32_e32 %0, %1, %2\n\t") : "+v"(a0), "+v"(a1) : "v"(b));   // mul -> add chain
    if (MODE == 4) asm volatile(REP64("v_add_f32_e32 %0, %0, %1\n\tv_mul_f32_e32 %2, %3, %1\n\t") : "+v"(a0) : "v"(b), "v"(a1), "v"(a2));  // dep add + indep mul
    if (MODE == 6) asm volatile(REP64("v_mul_f32_e32 %1, %4, %0\n\tv_mul_f32_e32 %2, %0, %5\n\tv_mul_f32_dpp %3, %0, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_e32 %0, %2, %3\n\ts_nop 0\n\t")
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sb), "v"(b));                           // the chain kernel's producer step
    if (MODE == 7) asm volatile(REP64("v_mul_f32_e32 %1, %4, %0\n\tv_mul_f32_e32 %2, %0, %5\n\tv_mul_f32_dpp %3, %0, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_e32 %0, %2, %3\n\t")
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sb), "v"(b));                           // ... without the s_nop
    if (MODE == 8) asm volatile(REP64("v_mul_f32_e32 %1, %4, %0\n\tv_mul_f32_e32 %2, %0, %5\n\tv_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_fma_f32 %0, %3, %5, %2\n\t")
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sb), "v"(b));                           // mov_dpp variant (timing only)
    if (MODE == 5) asm volatile(REP64("v_mul_f32_dpp %1, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_e32 %0, %1, %2\n\ts_nop 1\n\t")
                                : "+v"(a0), "+v"(a1) : "v"(b));                                                     // dpp mul -> add chain
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)d0;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// the chain kernel's consumer loop shape: 8 products per group by two ds_read_b128, issued AHEAD groups
// before their eight dependent additions
template <int AHEAD, bool WAITS>
__global__ void consumer_probe(float* out, unsigned long long* clk, int iters)
{
  __shared__ float4 buf[64 * 66];
  for (int i = threadIdx.x; i < 64 * 66; i += blockDim.x) buf[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const float4* p = buf + threadIdx.x * 65;
  float acc = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
  {
    float4 v[4][2];
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) { v[g][0] = p[2 * g]; v[g][1] = p[2 * g + 1]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 32; ++g)
    {
      const int slot = (g + AHEAD) & 3;
      v[slot][0] = p[(2 * (g + AHEAD)) & 63]; v[slot][1] = p[(2 * (g + AHEAD) + 1) & 63];
      __builtin_amdgcn_sched_barrier(0);
      const float4 a = v[g & 3][0], b = v[g & 3][1];
      acc += a.x; acc += a.y; acc += a.z; acc += a.w; acc += b.x; acc += b.y; acc += b.z; acc += b.w;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int AHEAD> void run_consumer(int threads)
{
  float* out; unsigned long long* clk;
  hipMalloc(&out, 32 * threads * 4); hipMalloc(&clk, 32 * 16);
  const int iters = 2000;
  hipLaunchKernelGGL((consumer_probe<AHEAD, true>), dim3(32), dim3(threads), 0, 0, out, clk, 10);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((consumer_probe<AHEAD, true>), dim3(32), dim3(threads), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("consumer loop, reads %d groups ahead, %d threads: %6.2f cycles per step (1 dependent add + 1/4 ds_read_b128)\n", AHEAD, threads,
         (double)h[0] / ((double)iters * 256));
  hipFree(out); hipFree(clk);
}

template <int MODE> void run(const char* name, int per_iter, int blocks, int threads)
{
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, blocks * 16);
  const int iters = 20000;
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, clk, 100, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double n = (double)iters * 64 * per_iter;
  printf("%-44s blocks=%4d thr=%4d: %6.2f ns/instr  %6.2f memtime-ticks/instr  core clock ~%.0f MHz (memtime/realtime*100)\n", name, blocks, threads,
         ms * 1e6 / n, (double)h[0] / n, (double)h[0] / (double)h[1] * 100.0);
  hipFree(out); hipFree(clk);
}

int main()
{
  for (int blocks : {32, 256})
  {
    run<0>("dependent v_add_f32", 1, blocks, 64);
    run<1>("4 independent v_add_f32 chains", 4, blocks, 64);
    run<2>("dependent v_add_f64", 1, blocks, 64);
    run<3>("v_mul_f32 -> v_add_f32 chain", 2, blocks, 64);
    run<4>("dependent add + independent mul", 2, blocks, 64);
    run<5>("v_mul_f32_dpp -> v_add_f32 (+s_nop 1) chain", 2, blocks, 64);
  }
  run<0>("dependent v_add_f32, 4 waves/SIMD", 1, 256, 1024);
  run_consumer<1>(64); run_consumer<2>(64); run_consumer<3>(64); run_consumer<3>(448);
  run<6>("producer step (mul,mul,mul_dpp,add,nop)", 5, 32, 64);
  run<7>("producer step without s_nop", 4, 32, 64);
  run<8>("producer step shape with mov_dpp+fma", 4, 32, 64);
  run<6>("producer step, 2 waves/SIMD", 5, 32, 512);
  run<6>("producer step, 7 waves/CU", 5, 32, 448);
  return 0;
}
