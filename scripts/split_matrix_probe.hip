// split_matrix_probe.hip -- development probe (round 5): is a 16.4 GB matrix written fast exactly when its parts lie in different 32-GiB stretches of an
// allocation?  One arena of 200 GB; the row-lockstep store-only kernel of the analysis' shape (16 KiB rows, one workgroup per chunk of 1960 rows, every XCD a
// contiguous eighth of the chunks) writes a "matrix" whose PARTS (halves, quarters) start at chosen offsets of the arena.
// hipcc --offload-arch=gfx950 -O2 -w scripts/split_matrix_probe.hip -o scripts/bin/split_matrix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2f64 __attribute__((ext_vector_type(2)));
struct Parts { unsigned long long base[4]; unsigned count; };          // part p holds rows [p * rows / count, (p + 1) * rows / count)

__global__ __launch_bounds__(1024) void store_parts_kernel(Parts parts, size_t rows, unsigned chunk_len)
{
  const unsigned R = 8, q = gridDim.x / R, r = gridDim.x % R, x = blockIdx.x % R;
  const unsigned chunk = x * q + (x < r ? x : r) + blockIdx.x / R;
  const size_t t0 = (size_t)chunk * chunk_len, t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  const size_t per = (rows + parts.count - 1) / parts.count;
  v2f64 v; v.x = (double)threadIdx.x; v.y = 2.0;
  unsigned since = 0;
  for (size_t t = t0; t < t1; ++t)
  {
    const size_t p = t / per;
    v2f64* row = reinterpret_cast<v2f64*>(parts.base[p]) + (t - p * per) * 1024;
    row[threadIdx.x] = v;
    v.x += 1.0;
    if (++since == 8) { __syncthreads(); since = 0; }
  }
}

static double rate(const Parts& parts, size_t rows)
{
  const unsigned chunk_len = 1960, chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(store_parts_kernel, dim3(chunks), dim3(1024), 0, 0, parts, rows, chunk_len);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_parts_kernel, dim3(chunks), dim3(1024), 0, 0, parts, rows, chunk_len);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return (double)rows * 16384.0 / (ms / 2 * 1e-3) / 1e9;
}

int main()
{
  const size_t GiB = (size_t)1 << 30, arena_bytes = (size_t)186 * GiB, rows = 1000000;
  char* arena = nullptr;
  if (hipMalloc((void**)&arena, arena_bytes) != hipSuccess) { printf("no arena\n"); return 1; }
  printf("arena of 186 GiB at %p; matrix of %zu rows x 16 KiB = 15.26 GiB\n", (void*)arena, rows);
  auto at = [&](size_t gib) { return (unsigned long long)(arena + gib * GiB); };
  printf("contiguous, by offset (GiB):");
  std::vector<double> whole;
  for (size_t o = 0; o + 16 <= 186; o += 8) { Parts p{{at(o), 0, 0, 0}, 1}; whole.push_back(rate(p, rows)); printf("  %zu:%.0f", o, whole.back()); }
  printf("\n");
  // halves (7.63 GiB each) at offsets (a, b)
  const size_t pairs[][2] = {{0, 8}, {0, 16}, {0, 24}, {0, 32}, {0, 40}, {0, 48}, {0, 64}, {0, 96}, {16, 48}, {8, 72}, {100, 140}, {136, 168}, {136, 144}, {40, 44}, {0, 170}};
  for (auto& ab : pairs) { Parts p{{at(ab[0]), at(ab[1]), 0, 0}, 2}; printf("halves at %3zu and %3zu GiB: %.0f GB/s\n", ab[0], ab[1], rate(p, rows)); }
  const size_t quads[][4] = {{0, 4, 8, 12}, {0, 40, 80, 120}, {0, 32, 64, 96}, {0, 8, 40, 48}, {136, 140, 170, 174}};
  for (auto& qd : quads) { Parts p{{at(qd[0]), at(qd[1]), at(qd[2]), at(qd[3])}, 4}; printf("quarters at %3zu, %3zu, %3zu, %3zu GiB: %.0f GB/s\n", qd[0], qd[1], qd[2], qd[3], rate(p, rows)); }
  return 0;
}
