// relay_probe.hip -- development probe for carry_relay_kernel: what a token hand-off between two waves of a workgroup
// costs through LDS, by polling strategy, with and without a chain of dependent additions in each turn.
// hipcc --offload-arch=gfx950 -O2 scripts/relay_probe.hip -o scripts/bin/relay_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef volatile __attribute__((address_space(3))) u2* lds_u2_p;
typedef volatile __attribute__((address_space(3))) unsigned* lds_u_p;

#define REP8(x) x x x x x x x x
#define REP128(x) REP8(REP8(x)) REP8(REP8(x))

// MODE 0: one ds_read_b64 per poll, per-lane token {acc, seq}
// MODE 1: two polls in flight
// MODE 2: separate flag word (wave-uniform address, one ds_read_b32 per poll), acc read after the flag was seen
// MODE 3: as 0, with s_sleep 1 between polls
// MODE 4: as 2 but the waiting wave sleeps (s_sleep) and the writer wakes the workgroup up (s_wakeup) after its write
template <int MODE, int ADDS>
__global__ __launch_bounds__(512) void relay(float* out, unsigned long long* clk, int turns, float b)
{
  __shared__ __attribute__((aligned(16))) u2 mail[64];
  __shared__ unsigned flag;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = blockDim.x >> 6;
  if (threadIdx.x < 64) { u2 z; z.x = 0; z.y = 0; mail[threadIdx.x] = z; }
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  lds_u2_p my = (lds_u2_p)&mail[lane];
  lds_u_p fl = (lds_u_p)&flag;
  float acc = (float)lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int g = wave; g < turns; g += C)
  {
    if (g > 0)
    {
      if (MODE == 0 || MODE == 3)
      {
        for (;;)
        {
          const u2 tk = *my;
          if (__all((int)(tk.y == (unsigned)g))) { acc = __int_as_float((int)tk.x); break; }
          if (MODE == 3) __builtin_amdgcn_s_sleep(1);
        }
      }
      else if (MODE == 1)
      {
        u2 t0v = *my, t1v;
        for (;;)
        {
          t1v = *my;
          if (__all((int)(t0v.y == (unsigned)g))) { acc = __int_as_float((int)t0v.x); break; }
          t0v = *my;
          if (__all((int)(t1v.y == (unsigned)g))) { acc = __int_as_float((int)t1v.x); break; }
        }
      }
      else
      {
        for (;;)
        {
          const unsigned f = *fl;
          if (f == (unsigned)g) break;
          if (MODE == 4) __builtin_amdgcn_s_sleep(8);
        }
        acc = __int_as_float((int)(*my).x);
      }
    }
    if (ADDS) asm volatile(REP128("v_add_f32_e32 %0, %0, %1\n\t") : "+v"(acc) : "v"(b));
    u2 tk; tk.x = (unsigned)__float_as_int(acc); tk.y = (unsigned)(g + 1);
    *my = tk;
    if (MODE == 2 || MODE == 4)
    {
      if (lane == 0) *fl = (unsigned)(g + 1);
      if (MODE == 4) asm volatile("s_wakeup");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE, int ADDS> void run(const char* what, int waves, int blocks)
{
  float* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)blocks * 512 * sizeof(float));
  hipMalloc(&clk, blocks * sizeof(unsigned long long));
  const int turns = 4000;
  relay<MODE, ADDS><<<blocks, 64 * waves>>>(out, clk, turns, 1.0f);
  hipDeviceSynchronize();
  relay<MODE, ADDS><<<blocks, 64 * waves>>>(out, clk, turns, 1.0f);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); return; }
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += (double)v;
  // wave 0 of each block times its own turns: the last turn it takes is about turns - C
  printf("%-62s waves=%d blocks=%3d: %7.1f cycles per turn%s\n", what, waves, blocks, sum / blocks / turns, ADDS ? " (128 dependent adds = 576 of them)" : "");
  hipFree(out); hipFree(clk);
}

int main()
{
  for (int waves : {2, 4, 6, 8})
  {
    run<0, 0>("token only, ds_read_b64 poll", waves, 128);
    run<1, 0>("token only, two polls in flight", waves, 128);
    run<2, 0>("token only, flag word + acc read", waves, 128);
    run<3, 0>("token only, ds_read_b64 poll + s_sleep 1", waves, 128);
    run<4, 0>("token only, s_sleep + s_wakeup", waves, 128);
  }
  for (int waves : {2, 4, 8})
  {
    run<0, 1>("token + 128 adds, ds_read_b64 poll", waves, 128);
    run<1, 1>("token + 128 adds, two polls in flight", waves, 128);
    run<2, 1>("token + 128 adds, flag word + acc read", waves, 128);
    run<4, 1>("token + 128 adds, s_sleep + s_wakeup", waves, 128);
  }
  run<0, 1>("token + 128 adds, ds_read_b64 poll", 1, 128);
  return 0;
}
