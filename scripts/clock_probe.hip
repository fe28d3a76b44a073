// clock_probe.hip -- development probe: which shader clock does a short, isolated kernel see (a 100-sample hop),
// against back-to-back launches?  hipcc --offload-arch=gfx950 -O2 -w scripts/clock_probe.hip -o scripts/bin/clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>

__global__ void spin(unsigned long long* out, int iters)
{
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float v = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)v; }
}

int main()
{
  unsigned long long* d; hipMalloc(&d, 64);
  unsigned long long h[3];
  auto once = [&](const char* what, int iters)
  {
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, iters);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %8llu shader ticks in %7.2f us => %7.1f MHz\n", what, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
  };
  once("first launch (2000 adds)", 2000);
  for (int k = 0; k < 3; ++k) { usleep(20000); once("after 20 ms idle (2000 adds ~ one hop)", 2000); }
  for (int k = 0; k < 3; ++k) { usleep(200); once("after 0.2 ms idle (2000 adds)", 2000); }
  for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, 2000);
  once("after 200 back-to-back launches", 2000);
  once("long kernel (2M adds)", 2000000);
  once("right after the long kernel (2000 adds)", 2000);
  hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, 0, d, 200000);
  once("after a chip-filling kernel (2000 adds)", 2000);
  return 0;
}
