// inverse_forms_probe.hip -- development probe (round 6): the synthesis kernels side by side on one matrix, every type pair.  The tiles of
// inverse_exact_kernel by their shape (rows per wave x bytes of a row per load instruction x tiles in flight; the library's forms read 256- or
// 512-byte pieces of 16 ... 32 rows per wave, here also 1 KiB pieces of 4 ... 8 rows), whole rows in step with the tree sum and the proof
// (inverse_rows1_kernel, float samples from double bins), and whole rows with the ordered sum (inverse_rows_ordered_kernel) by chunks and
// loader waves.  The digest of the samples is printed: every form gives the same bits.
// hipcc --offload-arch=gfx950 -O3 -w -I sdft_amd/csrc scripts/inverse_forms_probe.hip -o scripts/bin/inverse_forms_probe
#include "sdft_inverse.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace sdfthip;

template <typename FD> __global__ void fill_kernel(cx<FD>* p, size_t count)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
  {
    unsigned h = (unsigned)(i * 2654435761ull) ^ (unsigned)(i >> 17);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    p[i] = cmake<FD>((FD)((int)(h & 0xffff) - 32768) * (FD)(1.0 / 32768), (FD)((int)(h >> 16) - 32768) * (FD)(1.0 / 32768));
  }
}
template <typename TD> __global__ void sum_kernel(const TD* y, size_t n, unsigned long long* out)
{
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    unsigned long long bits = 0;
    __builtin_memcpy(&bits, &y[i], sizeof(TD));
    acc += bits * (2 * i + 1);
  }
  atomicAdd(out, acc);
}

template <typename TD, typename FD, int RW, int DEPTH, int RPI>
static void run(const char* name, InverseArgs<TD, FD> a, size_t max_blocks, unsigned long long* d_sum, hipStream_t s)
{
  const size_t waves = (a.n + RW - 1) / RW;
  const size_t blocks = std::max<size_t>(1, std::min<size_t>((waves + kWavesPerBlock - 1) / kWavesPerBlock, max_blocks));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f, sum_ms = 0.f;
  const int reps = 6;
  for (int it = 0; it < reps + 2; ++it)
  {
    hipEventRecord(e0, s);
    hipLaunchKernelGGL((inverse_exact_kernel<TD, FD, true, RW, DEPTH, false, RPI>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) { best = std::min(best, ms); sum_ms += ms; }
  }
  hipMemsetAsync(d_sum, 0, 8, s);
  hipLaunchKernelGGL((sum_kernel<TD>), dim3(1024), dim3(256), 0, s, a.y, a.n, d_sum);
  unsigned long long h = 0; hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
  const double bytes = (double)a.n * ((double)a.nbins * sizeof(cx<FD>) + sizeof(TD));
  printf("%-44s blocks %6zu: mean %7.3f ms %6.0f GB/s  best %7.3f ms %6.0f GB/s  digest %016llx%s\n", name, blocks, sum_ms / reps, bytes / (sum_ms / reps) * 1e-6,
         best, bytes / best * 1e-6, h, hipGetLastError() == hipSuccess ? "" : "  (error)");
  fflush(stdout);
}

template <typename TD, typename FD, int NLOAD = 11>
static void run_ordered(const char* name, InverseArgs<TD, FD> a, size_t want_chunks, size_t lds_budget, unsigned long long* d_sum, hipStream_t s)
{
  ordered_rows_geometry<FD, NLOAD> geo;
  if (!geo.make(a.nbins, lds_budget)) { printf("%-44s does not apply\n", name); return; }
  auto kern = inverse_rows_ordered_kernel<TD, FD, true, NLOAD>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.lds_bytes) != hipSuccess) { printf("attribute refused\n"); return; }
  size_t chunk = (a.n + want_chunks - 1) / want_chunks;
  chunk = std::max<size_t>(geo.G, (chunk + geo.G - 1) / geo.G * geo.G);
  const size_t nchunks = (a.n + chunk - 1) / chunk;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f, sum_ms = 0.f;
  const int reps = 6;
  hipMemsetAsync(a.y, 0xff, a.n * sizeof(TD), s);
  for (int it = 0; it < reps + 2; ++it)
  {
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(kern, dim3((unsigned)nchunks), dim3(kWave * (NLOAD + 1)), geo.lds_bytes, s, a, (unsigned)chunk, geo);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) { best = std::min(best, ms); sum_ms += ms; }
  }
  hipMemsetAsync(d_sum, 0, 8, s);
  hipLaunchKernelGGL((sum_kernel<TD>), dim3(1024), dim3(256), 0, s, a.y, a.n, d_sum);
  unsigned long long h = 0; hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
  const double bytes = (double)a.n * ((double)a.nbins * sizeof(cx<FD>) + sizeof(TD));
  char full[160]; snprintf(full, sizeof(full), "%s (%d ld, %u x %u, %zu KB)", name, NLOAD, geo.NG, geo.G, geo.lds_bytes / 1024);
  printf("%-44s chunks %6zu: mean %7.3f ms %6.0f GB/s  best %7.3f ms %6.0f GB/s  digest %016llx%s\n", full, nchunks, sum_ms / reps, bytes / (sum_ms / reps) * 1e-6,
         best, bytes / best * 1e-6, h, hipGetLastError() == hipSuccess ? "" : "  (error)");
  fflush(stdout);
}

template <typename TD, typename FD>
static void run_step(const char* name, InverseArgs<TD, FD> a, size_t want_chunks, unsigned long long* d_sum, hipStream_t s)
{
  if constexpr (sizeof(TD) == 4 && sizeof(FD) == 8)
  {
    size_t chunk = (a.n + want_chunks - 1) / want_chunks;
    chunk = std::max<size_t>(16, (chunk + 3) / 4 * 4);
    const size_t nchunks = (a.n + chunk - 1) / chunk;
    const unsigned waves = (a.nbins + 63) / 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum_ms = 0.f;
    const int reps = 6;
    for (int it = 0; it < reps + 2; ++it)
    {
      hipEventRecord(e0, s);
      hipLaunchKernelGGL((inverse_rows1_kernel<TD, FD, true>), dim3((unsigned)nchunks), dim3(waves * kWave), 0, s, a, (unsigned)chunk);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
      if (it >= 2) { best = std::min(best, ms); sum_ms += ms; }
    }
    hipMemsetAsync(d_sum, 0, 8, s);
    hipLaunchKernelGGL((sum_kernel<TD>), dim3(1024), dim3(256), 0, s, a.y, a.n, d_sum);
    unsigned long long h = 0; hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
    const double bytes = (double)a.n * ((double)a.nbins * sizeof(cx<FD>) + sizeof(TD));
    printf("%-44s chunks %6zu: mean %7.3f ms %6.0f GB/s  best %7.3f ms %6.0f GB/s  digest %016llx%s\n", name, nchunks, sum_ms / reps, bytes / (sum_ms / reps) * 1e-6,
           best, bytes / best * 1e-6, h, hipGetLastError() == hipSuccess ? "" : "  (error)");
    fflush(stdout);
  }
}

template <typename TD, typename FD> static void sweep(size_t n, unsigned nbins, int nt)
{
  hipStream_t s; hipStreamCreate(&s);
  cx<FD>* mat = nullptr; TD* y = nullptr; cx<FD>* syn = nullptr; unsigned long long* d_sum = nullptr;
  if (hipMalloc(&mat, n * (size_t)nbins * sizeof(cx<FD>)) != hipSuccess) { printf("no memory\n"); return; }
  hipMalloc(&y, n * sizeof(TD)); hipMalloc(&syn, nbins * sizeof(cx<FD>)); hipMalloc(&d_sum, 8);
  hipMemset(syn, 0, nbins * sizeof(cx<FD>));
  hipLaunchKernelGGL((fill_kernel<FD>), dim3(4096), dim3(256), 0, s, mat, n * (size_t)nbins);
  hipStreamSynchronize(s);
  InverseArgs<TD, FD> a{};
  a.in = mat; a.in_stride = n * (size_t)nbins; a.in_rows = nullptr; a.syn = syn; a.y = y; a.y_stride = n; a.n = n; a.nbins = nbins; a.channels = 1;
  a.sweight = (FD)(1.0 / nbins); a.op.kind = 0; a.nt = nt; a.nt_skip = 0;
  printf("---- TD %zu bytes, FD %zu bytes, n = %zu, N = %u, %s loads ----\n", sizeof(TD), sizeof(FD), n, nbins, nt ? "non-temporal" : "ordinary");
  for (int rnd = 0; rnd < 2; ++rnd)
  {
    const size_t big = (size_t)256 * 32;
    if constexpr (sizeof(FD) == 8) run<TD, FD, 32, 1, 4>("32 rows x 256 B, 1 tile ahead (library)", a, big, d_sum, s);
    run<TD, FD, 16, 1, 4>("16 rows x 256 B, 1 ahead (library)", a, big, d_sum, s);
    run<TD, FD, 16, 1, 2>("16 rows x 512 B, 1 ahead (library)", a, big, d_sum, s);
    run<TD, FD, 8, 4, 4>("8 rows x 256 B, 4 ahead (library)", a, big, d_sum, s);
    run_step<TD, FD>("rows in step, tree sum + proof (library)", a, 2048, d_sum, s);
    run_step<TD, FD>("rows in step, tree sum + proof (library)", a, 1024, d_sum, s);
    run_ordered<TD, FD>("whole rows, ordered", a, 512, 140 * 1024, d_sum, s);
    run_ordered<TD, FD>("whole rows, ordered", a, 1024, 140 * 1024, d_sum, s);
    run_ordered<TD, FD>("whole rows, ordered", a, 2048, 140 * 1024, d_sum, s);
    run_ordered<TD, FD>("whole rows, ordered", a, 4096, 140 * 1024, d_sum, s);
    run_ordered<TD, FD, 11>("whole rows, ordered", a, 256, 140 * 1024, d_sum, s);
    run_ordered<TD, FD, 11>("whole rows, ordered", a, 768, 140 * 1024, d_sum, s);
    run_ordered<TD, FD, 11>("whole rows, ordered", a, 1500, 140 * 1024, d_sum, s);
    run_ordered<TD, FD, 11>("whole rows, ordered, 158 KB", a, 512, 158 * 1024, d_sum, s);
    run_ordered<TD, FD, 11>("whole rows, ordered, 158 KB", a, 1024, 158 * 1024, d_sum, s);
    run_ordered<TD, FD, 11>("whole rows, ordered, 158 KB", a, 2048, 158 * 1024, d_sum, s);
    run_ordered<TD, FD, 7>("whole rows, ordered", a, 2048, 140 * 1024, d_sum, s);
    run_ordered<TD, FD, 8>("whole rows, ordered", a, 2048, 140 * 1024, d_sum, s);
    run<TD, FD, 8, 2, 2>("8 rows x 512 B, 2 ahead", a, big, d_sum, s);
    run<TD, FD, 8, 1, 1>("8 rows x 1 KiB, 1 ahead", a, big, d_sum, s);
    run<TD, FD, 4, 2, 1>("4 rows x 1 KiB, 2 ahead", a, big, d_sum, s);
  }
  hipFree(mat); hipFree(y); hipFree(syn); hipFree(d_sum); hipStreamDestroy(s);
}

int main(int argc, char** argv)
{
  const int which = argc > 1 ? atoi(argv[1]) : 0;
  if (which == 0 || which == 1) sweep<double, double>(1000000, 1024, 1);
  if (which == 0 || which == 2) sweep<float, float>(262144, 4096, 1);
  if (which == 0 || which == 4) sweep<float, float>(1000000, 1024, 1);
  if (which == 0 || which == 3) sweep<double, double>(1000000, 1000, 1);
  if (which == 0 || which == 5) sweep<float, double>(1000000, 1024, 1);
  if (which == 0 || which == 6) sweep<float, double>(500000, 2048, 1);
  if (which == 0 || which == 7) sweep<float, float>(1000000, 2048, 1);
  if (which == 8) sweep<float, double>(48000, 1024, 1);
  if (which == 9) sweep<float, double>(48000, 1024, 0);
  if (which == 10) sweep<float, double>(200000, 1024, 1);
  return 0;
}
