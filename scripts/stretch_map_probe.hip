// stretch_map_probe.hip -- development probe (round 6): WHAT is a "stretch" of device memory?  Round 5 found that a 16.4 GB matrix takes the analysis' store
// stream at 7.1 instead of 5.8 TB/s exactly when its halves lie in different stretches of an allocation (profiles/r05_split_matrix.txt).  This probe maps them:
//   0. what the system says about the memory (runtime properties, allocation granularity, KFD topology memory banks, partition modes, address range)
//   1. how small the two parts may be for the contrast to show (so that the map can be fine)
//   2. part A fixed, part B moved in steps of 256 MiB over the whole arena: where the rate switches = where a stretch ends (three positions of A)
//   3. every pair of positions on an 8 GiB grid: is "same stretch" an equivalence with FEW classes (a property of the physical memory: ranks / banks) or are
//      all stretches different from each other (a property of the allocation: physically contiguous blocks)?
//   4. the same questions asked of separate allocations against the arena's first stretch
// hipcc --offload-arch=gfx950 -O2 -w scripts/stretch_map_probe.hip -o scripts/bin/stretch_map_probe ; scripts/bin/stretch_map_probe [arena GiB = 160]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <dirent.h>

typedef double v2f64 __attribute__((ext_vector_type(2)));
struct Parts { unsigned long long base[2]; };            // part p holds rows [p * rows / 2, (p + 1) * rows / 2)

// the analysis' store stream: one workgroup per chunk of rows, 16 KiB rows written in step, every XCD a contiguous eighth of the chunks
__global__ __launch_bounds__(1024) void store_parts_kernel(Parts parts, size_t rows, unsigned chunk_len)
{
  const unsigned R = 8, q = gridDim.x / R, r = gridDim.x % R, x = blockIdx.x % R;
  const unsigned chunk = x * q + (x < r ? x : r) + blockIdx.x / R;
  const size_t t0 = (size_t)chunk * chunk_len, t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  const size_t per = (rows + 1) / 2;
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

static hipEvent_t e0, e1;
static double rate(const Parts& parts, size_t rows, int reps = 2)
{
  const unsigned chunk_len = (unsigned)((rows + 510) / 511), chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
  hipLaunchKernelGGL(store_parts_kernel, dim3(chunks), dim3(1024), 0, 0, parts, rows, chunk_len);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_parts_kernel, dim3(chunks), dim3(1024), 0, 0, parts, rows, chunk_len);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return (double)rows * 16384.0 / (ms / reps * 1e-3) / 1e9;
}

static void cat(const std::string& path, const char* indent = "    ")
{
  FILE* f = fopen(path.c_str(), "r");
  if (!f) { printf("%s%s: not readable\n", indent, path.c_str()); return; }
  char line[512];
  printf("%s%s:\n", indent, path.c_str());
  while (fgets(line, sizeof line, f)) printf("%s  %s", indent, line);
  fclose(f);
}
static std::vector<std::string> ls(const std::string& dir)
{
  std::vector<std::string> out;
  if (DIR* d = opendir(dir.c_str())) { while (dirent* e = readdir(d)) if (e->d_name[0] != '.') out.push_back(e->d_name); closedir(d); }
  return out;
}

int main(int argc, char** argv)
{
  const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
  size_t arena_gib = argc > 1 ? (size_t)atoll(argv[1]) : 160;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // ---- 0. what the system says ----
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
  printf("== 0. the system ==\ndevice %s (%s): totalGlobalMem %.3f GiB, free %.3f GiB, memoryBusWidth %d bit, memoryClockRate %d kHz, l2CacheSize %d B, CUs %d\n",
         pr.name, pr.gcnArchName, pr.totalGlobalMem / (double)GiB, fr / (double)GiB, pr.memoryBusWidth, pr.memoryClockRate, pr.l2CacheSize, pr.multiProcessorCount);
  {
    hipMemAllocationProp ap; memset(&ap, 0, sizeof ap);
    ap.type = hipMemAllocationTypePinned; ap.location.type = hipMemLocationTypeDevice; ap.location.id = 0;
    size_t gmin = 0, grec = 0;
    hipMemGetAllocationGranularity(&gmin, &ap, hipMemAllocationGranularityMinimum);
    hipMemGetAllocationGranularity(&grec, &ap, hipMemAllocationGranularityRecommended);
    printf("hipMemGetAllocationGranularity: minimum %zu B, recommended %zu B\n", gmin, grec);
  }
  for (const std::string& node : ls("/sys/class/kfd/kfd/topology/nodes"))
  {
    const std::string nd = "/sys/class/kfd/kfd/topology/nodes/" + node;
    const std::vector<std::string> banks = ls(nd + "/mem_banks");
    printf("KFD node %s: %zu memory bank(s)\n", node.c_str(), banks.size());
    for (const std::string& b : banks) cat(nd + "/mem_banks/" + b + "/properties");
  }
  for (const std::string& card : ls("/sys/class/drm"))
  {
    if (card.compare(0, 4, "card") != 0 || card.find('-') != std::string::npos) continue;
    const std::string dv = "/sys/class/drm/" + card + "/device/";
    for (const char* f : {"current_memory_partition", "available_memory_partition", "current_compute_partition", "mem_info_vram_total", "mem_info_vram_used", "mem_info_vis_vram_total", "mem_info_vram_vendor"})
      cat(dv + f, "  ");
  }
  // ---- the arena ----
  char* arena = nullptr;
  while (arena_gib >= 80 && hipMalloc((void**)&arena, arena_gib * GiB) != hipSuccess) { (void)hipGetLastError(); arena = nullptr; arena_gib -= 16; }
  if (!arena) { printf("no arena\n"); return 1; }
  {
    hipDeviceptr_t b = nullptr; size_t sz = 0;
    hipMemGetAddressRange(&b, &sz, (hipDeviceptr_t)(arena + 5 * GiB));
    printf("arena of %zu GiB at %p; hipMemGetAddressRange(arena + 5 GiB) = base %p size %zu (%.3f GiB): one range\n", arena_gib, (void*)arena, (void*)b, sz, sz / (double)GiB);
  }
  auto at = [&](size_t mib) { return (unsigned long long)(arena + mib * MiB); };
  const size_t A = arena_gib * 1024;                         // arena in MiB

  // ---- 1. how small may the parts be ----
  printf("\n== 1. contrast by part size: parts at (0, 8 GiB) [same stretch in every session so far] against (0, 72 GiB) and (0, 100 GiB) ==\n");
  for (size_t part_mib : {256, 512, 1024, 2048, 4096, 7812})
  {
    const size_t rows = part_mib * 2 * 64;                   // 64 rows of 16 KiB per MiB
    const double same = rate(Parts{{at(0), at(8192)}}, rows, 4), d72 = rate(Parts{{at(0), at(72 * 1024)}}, rows, 4), d100 = rate(Parts{{at(0), at(100 * 1024)}}, rows, 4);
    printf("parts of %5zu MiB: (0, 8) %.0f   (0, 72) %.0f   (0, 100) %.0f GB/s   contrast %.3f / %.3f\n", part_mib, same, d72, d100, d72 / same, d100 / same);
  }

  // ---- 2. the map: A fixed, B every 256 MiB ----
  const size_t part_mib = 1024, rows = part_mib * 2 * 64;
  for (size_t a_gib : {(size_t)0, (size_t)72, (size_t)100})
  {
    if ((a_gib + 2) * 1024 > A) continue;
    printf("\n== 2. parts of %zu MiB; part A at %zu GiB, part B at offset (GiB) -> GB/s, every 256 MiB ==\n", part_mib, a_gib);
    std::vector<double> v;
    for (size_t b = 0; b + part_mib <= A; b += 256)
    {
      const bool overlap = b < a_gib * 1024 + part_mib && a_gib * 1024 < b + part_mib;
      v.push_back(overlap ? 0.0 : rate(Parts{{at(a_gib * 1024), at(b)}}, rows, 2));
    }
    double lo = 1e30, hi = 0; for (double r : v) if (r > 0) { lo = r < lo ? r : lo; hi = r > hi ? r : hi; }
    const double mid = 0.5 * (lo + hi);
    for (size_t i = 0; i < v.size(); ++i) { if (i % 16 == 0) printf("\n%6.2f:", i * 0.25); printf(" %5.0f", v[i]); }
    printf("\nmin %.0f max %.0f; as classes (. = slow: same stretch as A, # = fast, - = overlaps A), one character per 256 MiB, 64 per line (16 GiB):\n", lo, hi);
    for (size_t i = 0; i < v.size(); ++i) { if (i % 64 == 0) printf("\n%4zu GiB ", i / 4); putchar(v[i] == 0.0 ? '-' : v[i] > mid ? '#' : '.'); }
    printf("\nswitches at (GiB):");
    int prev = -1;
    for (size_t i = 0; i < v.size(); ++i) { if (v[i] == 0.0) continue; const int c = v[i] > mid; if (prev >= 0 && c != prev) printf(" %.2f", i * 0.25); prev = c; }
    printf("\n");
  }

  // ---- 3. pairs on an 8 GiB grid ----
  {
    const size_t step = 8 * 1024, npos = (A - 2048) / step + 1, prow = 2048 * 2 * 64;      // parts of 2 GiB
    printf("\n== 3. every pair of positions on an 8 GiB grid, parts of 2 GiB: GB/s / 100 (rows: part A, columns: part B) ==\n      ");
    for (size_t j = 0; j < npos; ++j) printf("%4zu", j * 8);
    printf("\n");
    std::vector<std::vector<double>> mtx(npos, std::vector<double>(npos, 0.0));
    for (size_t i = 0; i < npos; ++i)
    {
      printf("%4zu: ", i * 8);
      for (size_t j = 0; j < npos; ++j)
      {
        if (j <= i) { printf("    "); continue; }
        mtx[i][j] = mtx[j][i] = rate(Parts{{at(i * step), at(j * step)}}, prow, 2);
        printf("%4.0f", mtx[i][j] / 100.0);
      }
      printf("\n");
    }
    // classes: position j joins the first class whose every member it is SLOW with
    double lo = 1e30, hi = 0;
    for (size_t i = 0; i < npos; ++i) for (size_t j = i + 1; j < npos; ++j) { lo = mtx[i][j] < lo ? mtx[i][j] : lo; hi = mtx[i][j] > hi ? mtx[i][j] : hi; }
    const double mid = 0.5 * (lo + hi);
    std::vector<std::vector<size_t>> classes;
    for (size_t j = 0; j < npos; ++j)
    {
      bool placed = false;
      for (auto& c : classes)
      {
        bool all_slow = true;
        for (size_t k : c) if (mtx[k][j] > mid) { all_slow = false; break; }
        if (all_slow) { c.push_back(j); placed = true; break; }
      }
      if (!placed) classes.push_back({j});
    }
    printf("threshold %.0f GB/s; positions that are slow with each other (greedy):\n", mid);
    for (size_t c = 0; c < classes.size(); ++c) { printf("  class %zu:", c); for (size_t k : classes[c]) printf(" %zu", k * 8); printf("\n"); }
  }

  // ---- 4. separate allocations against the arena ----
  {
    printf("\n== 4. separate allocations of 2 GiB (part B) against the arena's positions 0 and 72 GiB (part A), parts of 2 GiB ==\n");
    const size_t prow = 2048 * 2 * 64;
    std::vector<char*> bufs;
    for (int i = 0; i < 12; ++i) { char* p = nullptr; if (hipMalloc((void**)&p, 2 * GiB) != hipSuccess) { (void)hipGetLastError(); break; } bufs.push_back(p); }
    for (size_t i = 0; i < bufs.size(); ++i)
      printf("allocation %2zu at %p: with arena+0 %.0f   with arena+72 GiB %.0f GB/s\n", i, (void*)bufs[i],
             rate(Parts{{at(0), (unsigned long long)bufs[i]}}, prow, 2), rate(Parts{{at(72 * 1024), (unsigned long long)bufs[i]}}, prow, 2));
    for (char* p : bufs) hipFree(p);
  }

  // ---- 5. the whole matrix: contiguous windows of 15.26 GiB every 2 GiB (what the library's placement probes) ----
  {
    printf("\n== 5. contiguous windows of 1e6 rows (15.26 GiB), offset (GiB) -> GB/s ==\n");
    const size_t mrows = 1000000, half = (mrows + 1) / 2;
    for (size_t o = 0; o * 1024 + 15626 <= A; o += 2)
    {
      const Parts p{{at(o * 1024), (unsigned long long)(arena + o * GiB + half * 16384)}};
      printf(" %zu:%.0f", o, rate(p, mrows, 2));
    }
    printf("\n");
  }
  hipFree(arena);
  return 0;
}
