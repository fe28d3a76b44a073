// sdft_common.hip -- untyped part of the C-ABI: error channel, device selection, self test.

#include "sdft_plan.hpp"
#include "sdft_keyed_once.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>

namespace sdfthip {

static thread_local std::string g_error;
static thread_local bool g_has_error = false;

void set_error(const char* what, const char* detail)
{
  g_error = std::string(what ? what : "?") + ": " + (detail ? detail : "?");
  g_has_error = true;
}

// calls that succeeded but have something to tell (a poll loop timed out and the call was re-run): never an error
static thread_local std::string g_warning;
static thread_local bool g_has_warning = false;
void set_warning(const char* what, const char* detail)
{
  g_warning = std::string(what ? what : "?") + ": " + (detail ? detail : "?");
  g_has_warning = true;
}

// ------------------------------------------------------------------------------------------
// Run-time compilation of a host's own spectral operation (sdft_hip_process_n with sdft_hip_op_expr).  The library carries the text of
// sdft_kernels.hpp; the host's statements become the header "sdft_user_expr.inc" of a hiprtc program that instantiates
// ONE kernel (the name expression the plan asks for: about a second).  libhiprtc.so is opened on first use -- the
// library itself depends on no HIP library (build.py) -- and a compiled kernel stays loaded for the life of the process.
// ------------------------------------------------------------------------------------------
static const char kKernelSource[] =
#include "sdft_kernels_src.inc"
    ;

namespace {
struct Rtc
{
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcAddNameExpression) add_name = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetLoweredName) lowered = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  std::once_flag once;
  KeyedOnce<hipFunction_t> kernels;                        // device | expression | name expression -> function (sdft_keyed_once.hpp)

  // host threads may arrive together (sdft_hip_check_expr takes no lock): the library is opened exactly once
  bool open()
  {
    std::call_once(once, [this]() { if (!open_once()) lib = nullptr; });
    return lib != nullptr;
  }
  bool open_once()
  {
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so"})
      if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!lib) return false;
#define SDFT_RTC_SYM(field, symbol) field = reinterpret_cast<decltype(field)>(dlsym(lib, #symbol)); if (!field) return false;
    SDFT_RTC_SYM(create, hiprtcCreateProgram) SDFT_RTC_SYM(destroy, hiprtcDestroyProgram) SDFT_RTC_SYM(add_name, hiprtcAddNameExpression)
    SDFT_RTC_SYM(compile, hiprtcCompileProgram) SDFT_RTC_SYM(log_size, hiprtcGetProgramLogSize) SDFT_RTC_SYM(log, hiprtcGetProgramLog)
    SDFT_RTC_SYM(lowered, hiprtcGetLoweredName) SDFT_RTC_SYM(code_size, hiprtcGetCodeSize) SDFT_RTC_SYM(code, hiprtcGetCode)
#undef SDFT_RTC_SYM
    return true;
  }
};
Rtc g_rtc;
}  // namespace

// the code object of `name_expr` (a kernel of sdft_kernels.hpp) with the host's statements compiled in; no GPU is needed
bool rtc_compile(const char* expr, const char* name_expr, const char* arch, std::string& lowered_name, std::vector<char>& code)
{
  if (!g_rtc.open()) { set_error("sdft_hip_process_n (expression)", "libhiprtc.so could not be opened (the operation is compiled at run time)"); return false; }
  // (SDFT_FIXED_OP: the kernel serves OP_USER only -- the dispatch on the operation and the other operations' code go)
  const std::string source = std::string("#define SDFT_USER_EXPR 1\n#define SDFT_FIXED_OP 6\n") + kKernelSource;
  const std::string body = std::string(expr) + "\n";
  const char* headers[] = {body.c_str()};
  const char* include_names[] = {"sdft_user_expr.inc"};
  hiprtcProgram prog = nullptr;
  if (g_rtc.create(&prog, source.c_str(), "sdft_kernels_user.hip", 1, headers, include_names) != HIPRTC_SUCCESS)
  { set_error("sdft_hip_process_n (expression)", "hiprtcCreateProgram failed"); return false; }
  bool ok = g_rtc.add_name(prog, name_expr) == HIPRTC_SUCCESS;
  const std::string arch_flag = std::string("--offload-arch=") + arch;
  const char* opts[] = {arch_flag.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wno-pragma-once-outside-header"};
  if (ok && g_rtc.compile(prog, 7, opts) != HIPRTC_SUCCESS)
  {
    // the compiler's words about the host's statements are the useful part of the failure
    size_t n = 0; g_rtc.log_size(prog, &n);
    std::string text(n ? n : 1, '\0');
    if (n > 1) g_rtc.log(prog, &text[0]);
    const size_t at = text.find("sdft_user_expr.inc");
    set_error("sdft_hip_process_n: the expression does not compile", text.substr(at == std::string::npos ? 0 : at, 1500).c_str());
    ok = false;
  }
  const char* low = nullptr;
  if (ok) ok = g_rtc.lowered(prog, name_expr, &low) == HIPRTC_SUCCESS && low;
  size_t bytes = 0;
  if (ok) { lowered_name = low; ok = g_rtc.code_size(prog, &bytes) == HIPRTC_SUCCESS && bytes > 0; }
  if (ok) { code.resize(bytes); ok = g_rtc.code(prog, code.data()) == HIPRTC_SUCCESS; }
  g_rtc.destroy(&prog);
  if (ok)
    if (const char* dir = getenv("SDFT_HIP_RTC_DUMP"))       // development aid: the code object, for llvm-objdump / llvm-readelf
    {
      static std::atomic<int> serial{0};
      const std::string path = std::string(dir) + "/sdft_rtc_" + std::to_string(serial.fetch_add(1)) + ".co";
      if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
    }
  if (!ok && !g_has_error) set_error("sdft_hip_process_n (expression)", "run-time compilation failed");
  return ok;
}

// the kernel `name_expr` with the host's statements, loaded on `device` (compiled once per process, expression and kernel)
bool rtc_kernel(const char* expr, const char* name_expr, int device, hipFunction_t* fn)
{
  // The second-long compilation runs OUTSIDE the cache lock (other plans' expression calls go on); a key being compiled is
  // marked in flight, and a thread that wants the same key waits for that compilation instead of starting its own (KeyedOnce).
  const std::string key = std::to_string(device) + "|" + expr + "|" + name_expr;
  return g_rtc.kernels.get(key, *fn, [&](hipFunction_t& f) -> bool
  {
    hipDeviceProp_t prop;
    SDFT_TRY(hipGetDeviceProperties(&prop, device));
    std::string arch = prop.gcnArchName;                     // "gfx950:sramecc+:xnack-"
    arch = arch.substr(0, arch.find(':'));
    std::string lowered; std::vector<char> code;
    if (!rtc_compile(expr, name_expr, arch.c_str(), lowered, code)) return false;
    hipModule_t module = nullptr;
    SDFT_TRY(hipModuleLoadData(&module, code.data()));
    SDFT_TRY(hipModuleGetFunction(&f, module, lowered.c_str()));
    return true;
  });
}

__global__ void lane_selftest_kernel(int* out)
{
  const int lane = threadIdx.x;
  out[lane] = lane_from_below(lane + 100);
  out[64 + lane] = lane_from_above(lane + 100);
  // row broadcast (carry_relay_kernel): every lane of a row of 16 receives the row's lane 3
  out[128 + lane] = __builtin_amdgcn_update_dpp(0, lane + 100, 0x153 /*row_newbcast:3*/, 0xf, 0xf, true);
}

// Verifies on the device that the DPP wave shifts move data the way the kernels assume
// (lane i <- lane i-1 / lane i+1).  Run once per process.
bool lane_selftest()
{
  static std::mutex mu;
  static int state = 0;       // 0 unknown, 1 ok, -1 failed
  std::lock_guard<std::mutex> lock(mu);
  if (state != 0) { if (state < 0) set_error("lane_selftest", "DPP wave shift semantics mismatch"); return state > 0; }
  int* d = nullptr;
  SDFT_TRY(hipMalloc((void**)&d, 192 * sizeof(int)));
  hipLaunchKernelGGL(lane_selftest_kernel, dim3(1), dim3(64), 0, 0, d);
  int h[192];
  hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { set_error("lane_selftest", hipGetErrorString(e)); return false; }
  bool ok = true;
  for (int l = 1; l < 64; ++l) ok = ok && (h[l] == l - 1 + 100);
  for (int l = 0; l < 63; ++l) ok = ok && (h[64 + l] == l + 1 + 100);
  for (int l = 0; l < 64; ++l) ok = ok && (h[128 + l] == (l & ~15) + 3 + 100);
  state = ok ? 1 : -1;
  if (!ok) set_error("lane_selftest", "DPP wave shift semantics mismatch");
  return ok;
}

// ------------------------------------------------------------------------------------------
// Achievable-write ceiling (measurement aid): a kernel that does nothing but the
// forward kernel's store stream.  pattern 0: plain linear grid-stride fill, 16 B per lane.
// pattern 1: the forward kernel's own geometry -- every wave owns `lanes` consecutive 16-byte
// slots of a row of `row_slots` slots and walks `chunk_len` consecutive rows.
// ------------------------------------------------------------------------------------------
typedef double v2f64 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kBlock) void store_linear_kernel(v2f64* dst, size_t slots)
{
  const size_t stride = (size_t)gridDim.x * kBlock;
  v2f64 v; v.x = (double)threadIdx.x; v.y = 1.0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < slots; i += stride) dst[i] = v;
}

// load-only counterpart: every thread streams 16-byte slots (four in flight), the sum keeps the loads alive
__global__ __launch_bounds__(kBlock) void load_linear_kernel(const v2f64* src, size_t slots, double* sink)
{
  const size_t stride = (size_t)gridDim.x * kBlock;
  double acc = 0.0;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  for (; i + 3 * stride < slots; i += 4 * stride)
  {
    const v2f64 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc += (a.x + b.x) + (c.x + d.x) + (a.y + b.y) + (c.y + d.y);
  }
  for (; i < slots; i += stride) { const v2f64 a = src[i]; acc += a.x + a.y; }
  if (acc == 12345.678) *sink = acc;                        // never true for the buffers measured; defeats dead-code elimination
}

__global__ __launch_bounds__(kBlock) void store_tiled_kernel(v2f64* dst, size_t rows, unsigned row_slots, unsigned lanes,
                                                            unsigned chunk_len, unsigned tiles, unsigned chunks)
{
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long wave = (unsigned long long)blockIdx.x * kWavesPerBlock + wib;
  if (wave >= (unsigned long long)tiles * chunks) return;
  const unsigned tile = (unsigned)(wave % tiles), chunk = (unsigned)(wave / tiles);
  const size_t t0 = (size_t)chunk * chunk_len;
  const size_t t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  const unsigned slot = tile * lanes + lane;
  if (lane >= (int)lanes || slot >= row_slots) return;
  v2f64 v; v.x = (double)lane; v.y = (double)tile;
  v2f64* p = dst + t0 * row_slots + slot;
  for (size_t t = t0; t < t1; ++t) { *p = v; p += row_slots; v.x += 1.0; }
}

// pattern 2: one workgroup of row_slots/64 waves per time chunk; the waves write one whole row
// per step and meet at a barrier every `sync_every` rows (the shape of a row-lockstep kernel).
// Round 5, the store-ceiling study (pattern bits on top of 2): MAP -- which chunk a workgroup takes: 0 = its own number
// (consecutive workgroups -> consecutive 32 MB regions; the dispatcher deals consecutive workgroups to different XCDs), 1 = every
// XCD a contiguous eighth of the matrix (chunk = (b % 8) * chunks / 8 + b / 8); STAGGER -- a workgroup starts `(b * 37) % 64`
// rows into its chunk and wraps around (workgroups in step no longer write the same row phase -- the same address bits above
// the row size -- at the same time).
template <bool NT, int MAP, bool STAGGER>
__global__ __launch_bounds__(1024) void store_rowgroup_kernel(v2f64* dst, size_t rows, unsigned row_slots, unsigned chunk_len,
                                                              unsigned sync_every, unsigned regions)
{
  unsigned chunk = blockIdx.x;
  if constexpr (MAP == 1)
  {
    // workgroup b is the (b / R)-th chunk of region b % R (R = 8: every XCD a contiguous eighth); a bijection for any grid
    const unsigned R = regions, q = gridDim.x / R, r = gridDim.x % R, x = blockIdx.x % R;
    chunk = x * q + (x < r ? x : r) + blockIdx.x / R;
  }
  const size_t t0 = (size_t)chunk * chunk_len;
  const size_t t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  if (t0 >= t1) return;
  v2f64 v; v.x = (double)threadIdx.x; v.y = 2.0;
  const size_t len = t1 - t0;
  size_t off = STAGGER ? ((size_t)blockIdx.x * 37u) % 64u % len : 0;
  unsigned since = 0;
  for (size_t i = 0; i < len; ++i)
  {
    v2f64* p = dst + (t0 + off) * row_slots + threadIdx.x;
    if (threadIdx.x < row_slots) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }
    v.x += 1.0;
    if (++off == len) off = 0;
    if (sync_every && ++since == sync_every) { __syncthreads(); since = 0; }
  }
}

// The same store stream (16 KiB rows in step, every XCD a contiguous eighth of the chunks) into a matrix whose two halves lie at two
// places: what tells whether two places of an allocation are the same KIND of memory (round 6, profiles/r06_stretch_map.txt: device
// memory comes in kinds that alternate every 16-32 GiB of a large allocation; the halves of a matrix written at the same time take
// the stream at 6.8-7.1 TB/s when they are of different kinds and at 5.6-5.8 when of one).
struct StoreParts { unsigned long long base[2]; };
__global__ __launch_bounds__(1024) void store_parts_kernel(StoreParts parts, size_t rows, unsigned chunk_len)
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
    reinterpret_cast<v2f64*>(parts.base[p])[(t - p * per) * 1024 + threadIdx.x] = v;
    v.x += 1.0;
    if (++since == 8) { __syncthreads(); since = 0; }
  }
}

// load-only counterpart of store_rowgroup_kernel (the synthesis' ceiling by access shape): one workgroup of row_slots/64 waves per
// chunk of rows reads whole rows in step (16 bytes per lane), `depth` rows in flight per lane; regions as above (0: workgroup b
// -> chunk b)
__global__ __launch_bounds__(1024) void load_rowgroup_kernel(const v2f64* src, size_t rows, unsigned row_slots, unsigned chunk_len, unsigned regions, double* sink)
{
  unsigned chunk = blockIdx.x;
  if (regions)
  {
    const unsigned R = regions, q = gridDim.x / R, r = gridDim.x % R, x = blockIdx.x % R;
    chunk = x * q + (x < r ? x : r) + blockIdx.x / R;
  }
  const size_t t0 = (size_t)chunk * chunk_len;
  const size_t t1 = t0 + chunk_len < rows ? t0 + chunk_len : rows;
  if (t0 >= t1 || threadIdx.x >= row_slots) return;
  const v2f64* p = src + t0 * row_slots + threadIdx.x;
  double acc = 0.0;
  size_t t = t0;
  for (; t + 4 <= t1; t += 4)
  {
    const v2f64 a = p[0], b = p[row_slots], c = p[2 * (size_t)row_slots], d = p[3 * (size_t)row_slots];
    acc += (a.x + b.x) + (c.x + d.x) + (a.y + b.y) + (c.y + d.y);
    p += 4 * (size_t)row_slots;
  }
  for (; t < t1; ++t) { const v2f64 a = *p; acc += a.x + a.y; p += row_slots; }
  if (acc == 12345.678) *sink = acc;
}

// measurement aid: holds one CU per workgroup (159 of the 160 KiB of LDS: no workgroup of the library shares the CU) for `ticks` of the 100 MHz clock
// without touching memory
__global__ __launch_bounds__(512) void hold_cu_kernel(unsigned long long ticks, unsigned* sink)
{
  extern __shared__ char hold_lds[];
  hold_lds[threadIdx.x] = 1;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0 && hold_lds[0] == 77) *sink = 1;
}

}  // namespace sdfthip

extern "C" {

// measurement aid (how much of its speed does a kernel keep when another kernel holds `cus` CUs -- what the relay kernel
// of the exact-carry route does to the forward launch): occupies `cus` CUs for `milliseconds` on a stream of its own and
// returns at once; the held CUs are released when the time is up.  0, or -1.
int sdft_hip_hold_cus(unsigned cus, double milliseconds)
{
  using namespace sdfthip;
  static hipStream_t s = nullptr;
  static unsigned* sink = nullptr;
  if (!s)
  {
    // (the statics are assigned only once all three steps have succeeded: a half-initialised first call must not make
    // later calls launch with a null sink or without the LDS attribute)
    hipStream_t ns = nullptr;
    unsigned* nsink = nullptr;
    if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (hipMalloc((void**)&nsink, 4) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(hold_cu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
    {
      (void)hipGetLastError();
      if (nsink) (void)hipFree(nsink);
      (void)hipStreamDestroy(ns);
      return -1;
    }
    s = ns; sink = nsink;
  }
  if (cus == 0) return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;       // wait for the release
  hipLaunchKernelGGL(hold_cu_kernel, dim3(cus), dim3(512), 159 * 1024, s, (unsigned long long)(milliseconds * 1e5), sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// sdft.h:184
extern const size_t sdft_convolution_kernel_size;
const size_t sdft_convolution_kernel_size = 2;

// Times `reps` launches of the store-only kernel over `bytes` of device memory at `dst`;
// returns the average milliseconds per launch (negative on error).
double sdft_hip_store_ceiling(void* dst, size_t bytes, int pattern, unsigned row_slots, unsigned lanes,
                              unsigned chunk_len, int reps)
{
  using namespace sdfthip;
  const size_t slots = bytes / 16;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  auto launch = [&]() {
    if (pattern == 0)
      hipLaunchKernelGGL(store_linear_kernel, dim3(256 * 8), dim3(kBlock), 0, 0, (v2f64*)dst, slots);
    else if ((pattern >= 2 && pattern <= 7) || pattern >= 100)   // 3: non-temporal stores (7: and XCD-contiguous chunks); 4: XCD-contiguous chunks; 5: staggered row phase; 6: both; 100 + R: R regions
    {
      const size_t rows = slots / row_slots;
      const unsigned chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
      const dim3 b(((row_slots + 63) / 64) * 64), g(chunks);
      const unsigned sync_every = lanes, regions = pattern > 100 ? (unsigned)(pattern - 100) : 8u;     // (pattern 100 would be no region at all: eight)
      if (pattern == 2) hipLaunchKernelGGL((store_rowgroup_kernel<false, 0, false>), g, b, 0, 0, (v2f64*)dst, rows, row_slots, chunk_len, sync_every, regions);
      else if (pattern == 3) hipLaunchKernelGGL((store_rowgroup_kernel<true, 0, false>), g, b, 0, 0, (v2f64*)dst, rows, row_slots, chunk_len, sync_every, regions);
      else if (pattern == 5) hipLaunchKernelGGL((store_rowgroup_kernel<false, 0, true>), g, b, 0, 0, (v2f64*)dst, rows, row_slots, chunk_len, sync_every, regions);
      else if (pattern == 6) hipLaunchKernelGGL((store_rowgroup_kernel<false, 1, true>), g, b, 0, 0, (v2f64*)dst, rows, row_slots, chunk_len, sync_every, regions);
      else if (pattern == 7) hipLaunchKernelGGL((store_rowgroup_kernel<true, 1, false>), g, b, 0, 0, (v2f64*)dst, rows, row_slots, chunk_len, sync_every, regions);
      else hipLaunchKernelGGL((store_rowgroup_kernel<false, 1, false>), g, b, 0, 0, (v2f64*)dst, rows, row_slots, chunk_len, sync_every, regions);
    }
    else
    {
      const size_t rows = slots / row_slots;
      const unsigned tiles = (row_slots + lanes - 1) / lanes;
      const unsigned chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
      const unsigned long long waves = (unsigned long long)tiles * chunks;
      hipLaunchKernelGGL(store_tiled_kernel, dim3((unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, 0,
                         (v2f64*)dst, rows, row_slots, lanes, chunk_len, tiles, chunks);
    }
  };
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (double)ms / (reps > 0 ? reps : 1);
}

// average ms of a load-only kernel over `bytes` of device memory (the synthesis kernel's ceiling)
double sdft_hip_load_ceiling(const void* src, size_t bytes, int reps)
{
  using namespace sdfthip;
  const size_t slots = bytes / 16;
  hipEvent_t e0, e1;
  double* sink = nullptr;
  if (hipMalloc((void**)&sink, 8) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(sink); return -1.0; }
  auto launch = [&]() { hipLaunchKernelGGL(load_linear_kernel, dim3(256 * 16), dim3(kBlock), 0, 0, (const v2f64*)src, slots, sink); };
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1, 0);
  float ms = -1.f;
  if (hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1); else (void)hipGetLastError();
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
  return ms < 0.f ? -1.0 : (double)ms / (reps > 0 ? reps : 1);
}

// the same for whole rows read in step by one workgroup per chunk of rows (load_rowgroup_kernel); regions: see store patterns
double sdft_hip_load_rows_ceiling(const void* src, size_t bytes, unsigned row_slots, unsigned chunk_len, unsigned regions, int reps)
{
  using namespace sdfthip;
  const size_t slots = bytes / 16, rows = slots / row_slots;
  const unsigned chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
  hipEvent_t e0, e1;
  double* sink = nullptr;
  if (hipMalloc((void**)&sink, 8) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(sink); return -1.0; }
  auto launch = [&]() { hipLaunchKernelGGL(load_rowgroup_kernel, dim3(chunks), dim3(((row_slots + 63) / 64) * 64), 0, 0, (const v2f64*)src, rows, row_slots, chunk_len, regions, sink); };
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1, 0);
  float ms = -1.f;
  if (hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1); else (void)hipGetLastError();
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
  return ms < 0.f ? -1.0 : (double)ms / (reps > 0 ? reps : 1);
}

// Device memory for a DFT matrix, chosen for how fast it can be WRITTEN.  Which physical memory backs a large allocation decides
// what the row-lockstep store stream reaches in it -- 5.85, 6.4 or 7.1 TB/s for 16 GB buffers of one process, the same buffer the same
// rate on every pass, reads unaffected (profiles/r05_buffer_placement.txt) -- and neither the runtime nor the library can move a
// buffer afterwards.  So: up to `candidates` allocations of `bytes` (as many as fit beside each other), each probed with the
// store-only kernel of the analysis' shape (rows of 16 KiB, every XCD a contiguous eighth: 2 launches), the best kept, the others
// freed.  *gbs (may be NULL) receives the kept buffer's probe rate.  Free with hipFree.  NULL on failure (sdft_hip_last_error).
void* sdft_hip_malloc_matrix(size_t bytes, int candidates, double* gbs)
{
  using namespace sdfthip;
  if (bytes == 0) return nullptr;
  void* best = nullptr;
  double best_ms = 0.0;
  std::vector<void*> others;
  const int tries = candidates < 1 ? 1 : (candidates > 16 ? 16 : candidates);
  for (int i = 0; i < tries; ++i)
  {
    size_t free_b = 0, total_b = 0;
    if (i > 0 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + (bytes >> 3))) { (void)hipGetLastError(); break; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
    double ms = 0.0;
    if (tries > 1 && bytes >= ((size_t)64 << 20))
    {
      const size_t rows = bytes / 16384;
      const unsigned chunk_len = (unsigned)std::max<size_t>(8, (rows + 510) / 511);       // two rounds of the chip, as the analysis cuts time
      ms = sdft_hip_store_ceiling(p, rows * 16384, 4, 1024, 8, chunk_len, 2);
      if (ms <= 0.0) ms = 1e30;
    }
    if (!best || ms < best_ms) { if (best) others.push_back(best); best = p; best_ms = ms; }
    else others.push_back(p);
  }
  for (void* p : others) (void)hipFree(p);
  if (!best) { set_error("sdft_hip_malloc_matrix", "out of device memory"); return nullptr; }
  if (gbs) *gbs = best_ms > 0.0 && best_ms < 1e29 ? (double)((bytes / 16384) * 16384) / (best_ms * 1e-3) / 1e9 : 0.0;
  return best;
}

// ------------------------------------------------------------------------------------------
// The same choice made inside ONE allocation, by what round 6 learnt about device memory (profiles/r06_stretch_map.txt): a large
// allocation is made of stretches of two or three KINDS of memory that alternate every 16-32 GiB (the first change of kind lies
// 32 or 64 GiB into the allocation in every session kept; the same map whichever place it is taken from: an equivalence, most
// likely the high physical address bits that select a rank of the HBM stacks, i.e. another set of DRAM banks), and the analysis'
// store stream -- every XCD writing its own eighth of the matrix at the same time -- runs at 6.8-7.1 TB/s when the matrix lies
// half in one kind and half in another, at 5.6-5.85 when all of it is of one kind.  So the place of a matrix is COMPUTED:
// small two-part probes (2 GiB written, 0.35 ms each) find the first place where the kind changes -- steps of 16 GiB, then
// bisection to 1 GiB -- and the matrix is the window centred on it.  Two full-size probes (the window, and a window at the
// allocation's start: what a plain hipMalloc would have been) check the result and the better of the two is returned; only an
// arena in which no change of kind is within reach is searched the old way (a window every 4 GiB, at most 8).
// ------------------------------------------------------------------------------------------
extern "C" {
// (layout of include/sdft/sdft_hip.h; this file sees no public header)
typedef struct
{
  size_t arena_bytes, window_offset, boundary_offset;
  int    pair_probes, window_probes;
  double window_gbs, start_gbs, probe_ms;
  int    arenas_tried;
} sdft_hip_placement_t;
}
namespace sdfthip {
struct ArenaEntry { void* window; void* base; sdft_hip_placement_t info; };
static std::mutex g_arena_mutex;
static std::vector<ArenaEntry> g_arenas;

// GB/s of the two-part store stream with parts of `part_bytes` at a and b (1 warm-up + 2 timed launches); 0 on failure
static double pair_rate(char* a, char* b, size_t part_bytes, double* ms_total)
{
  const size_t rows = 2 * (part_bytes / 16384);
  const unsigned chunk_len = (unsigned)std::max<size_t>(8, (rows + 510) / 511);
  const unsigned chunks = (unsigned)((rows + chunk_len - 1) / chunk_len);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e0); return 0.0; }
  StoreParts parts{{(unsigned long long)a, (unsigned long long)b}};
  hipLaunchKernelGGL(store_parts_kernel, dim3(chunks), dim3(1024), 0, 0, parts, rows, chunk_len);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_parts_kernel, dim3(chunks), dim3(1024), 0, 0, parts, rows, chunk_len);
  (void)hipEventRecord(e1, 0);
  float ms = 0.f;
  const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
  if (!ok) (void)hipGetLastError();
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (!ok) return 0.0;
  if (ms_total) *ms_total += 1.5 * ms;
  return (double)rows * 16384.0 / (ms / 2 * 1e-3) / 1e9;
}
}  // namespace sdfthip

namespace sdfthip {
// places a matrix of `bytes` inside the allocation [base, base + arena_bytes): fills `info`, returns the window's offset
static size_t place_in_arena(char* base, size_t bytes, size_t arena_bytes, sdft_hip_placement_t& info)
{
  memset(&info, 0, sizeof info);
  info.arena_bytes = arena_bytes;
  const size_t GiB = (size_t)1 << 30, room = arena_bytes - bytes;         // the window may start anywhere in [0, room]
  size_t off = 0;
  if (bytes >= ((size_t)64 << 20))
  {
    const size_t rows = bytes / 16384;
    const unsigned chunk_len = (unsigned)std::max<size_t>(8, (rows + 510) / 511);       // two rounds of the chip, as the analysis cuts time
    auto window_rate = [&](size_t o) {
      const double ms = sdft_hip_store_ceiling(base + o, rows * 16384, 4, 1024, 8, chunk_len, 2);
      ++info.window_probes;
      if (ms > 0.0) info.probe_ms += 3.0 * ms;
      return ms > 0.0 ? (double)(rows * 16384) / (ms * 1e-3) / 1e9 : 0.0;
    };
    // ---- where does the kind of memory change?  (parts of 1 GiB; a change shows as >= 1.1 x the rate of two parts of the start's kind) ----
    // (parts of 1 GiB -- of what the arena has left beyond the last place probed when that is less, down to 256 MiB: the contrast shows from there on.
    // A small matrix' arena ends just beyond 64 GiB, where round 5's sessions had their first change of kind)
    const size_t step = 16 * GiB;
    const size_t tail = arena_bytes % step;                              // what lies beyond the last multiple of 16 GiB
    const size_t part = ((tail >= ((size_t)256 << 20) && tail < GiB ? tail : GiB) >> 14) << 14;
    size_t found = 0;                                                    // offset of the change the window is centred on (0: none)
    bool probed = false;                                                 // the arena was large enough for the two-part probes
    if (arena_bytes >= 3 * part && room >= GiB)
    {
      probed = true;
      const double same = pair_rate(base, base + part, part, &info.probe_ms);
      ++info.pair_probes;
      auto differs = [&](size_t o) { ++info.pair_probes; return same > 0.0 && pair_rate(base, base + o, part, &info.probe_ms) > 1.1 * same; };
      // the centred window [b - bytes / 2, b + bytes / 2) has to fit: b <= room + bytes / 2; a change before bytes / 2 leaves the window at the
      // start (it straddles the change, off centre) -- kept in reserve, a later change that allows centring is preferred
      size_t early = 0;
      for (size_t o = step; o + part <= arena_bytes && o <= room + bytes / 2 + step && info.pair_probes < 12; o += step)
      {
        if (!differs(o)) continue;
        size_t lo = o - step, hi = o;                                    // the kind changes in (lo, hi]: bisect to 1 GiB
        while (hi - lo > GiB && info.pair_probes < 16)
        {
          const size_t mid = lo + ((hi - lo) / 2 / GiB) * GiB;
          if (mid == lo || mid + part > arena_bytes) break;
          if (differs(mid)) hi = mid; else lo = mid;
        }
        if (hi > room + bytes / 2) break;
        if (2 * hi >= bytes) { found = hi; break; }
        if (!early) early = hi;
        break;                                                           // (beyond the first change the reference kind is no longer the start's: stop here)
      }
      if (!found && early) found = early;
    }
    info.boundary_offset = found;
    info.start_gbs = window_rate(0);
    info.window_gbs = info.start_gbs;
    if (found)
    {
      const size_t centred = 2 * found >= bytes ? ((found - bytes / 2) >> 21) << 21 : 0;      // 2 MiB aligned
      const size_t cand = std::min(centred, (room >> 21) << 21);
      if (cand > 0)
      {
        const double r = window_rate(cand);
        if (r > info.window_gbs) { info.window_gbs = r; off = cand; }
      }
    }
    if (!found && !probed)
    {
      // an arena too small for the two-part probes: the search of round 5, a window every 4 GiB, at most 8 of them
      const size_t scan = (size_t)4 << 30;
      for (size_t o = scan; o <= room && info.window_probes < 9; o += scan)
      {
        const double r = window_rate(o);
        if (r > info.window_gbs) { info.window_gbs = r; off = o; }
      }
    }
  }
  info.window_offset = off;
  return off;
}
}  // namespace sdfthip

void* sdft_hip_malloc_matrix_in_arena(size_t bytes, size_t arena_bytes, double* gbs)
{
  using namespace sdfthip;
  if (gbs) *gbs = 0.0;
  if (bytes == 0 || arena_bytes < bytes) { set_error("sdft_hip_malloc_matrix_in_arena", "the arena is smaller than the matrix"); return nullptr; }
  if (bytes < ((size_t)64 << 20)) arena_bytes = bytes;     // (nothing is probed below 64 MiB: no arena either)
  char* base = nullptr;
  if (hipMalloc((void**)&base, arena_bytes) != hipSuccess) { (void)hipGetLastError(); set_error("sdft_hip_malloc_matrix_in_arena", "out of device memory"); return nullptr; }
  sdft_hip_placement_t info;
  size_t off = place_in_arena(base, bytes, arena_bytes, info);
  info.arenas_tried = 1;
  // An allocation may happen to lie in ONE kind of memory for all of matrix + 64 GiB (seen in a process that had allocated and freed a lot before).
  // Then a second allocation is made WHILE the first is held -- other memory by construction -- and the better of the two is kept.
  if (bytes >= ((size_t)64 << 20) && info.boundary_offset == 0 && info.pair_probes >= 2)     // (an arena that was probed at 16 GiB and beyond)
  {
    char* second = nullptr;
    if (hipMalloc((void**)&second, arena_bytes) == hipSuccess)
    {
      sdft_hip_placement_t info2;
      const size_t off2 = place_in_arena(second, bytes, arena_bytes, info2);
      info2.arenas_tried = 2;
      info2.probe_ms += info.probe_ms; info2.pair_probes += info.pair_probes; info2.window_probes += info.window_probes;
      if (info2.window_gbs > 1.02 * info.window_gbs) { (void)hipFree(base); base = second; off = off2; info = info2; }
      else { (void)hipFree(second); info.arenas_tried = 2; info.probe_ms = info2.probe_ms; info.pair_probes = info2.pair_probes; info.window_probes = info2.window_probes; }
    }
    else (void)hipGetLastError();
  }
  if (gbs) *gbs = info.window_gbs;
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  g_arenas.push_back(ArenaEntry{(void*)(base + off), (void*)base, info});
  return base + off;
}
// how a window of sdft_hip_malloc_matrix_in_arena was placed: 0, or -1 for a pointer that call did not return
int sdft_hip_matrix_placement(const void* window, sdft_hip_placement_t* out)
{
  using namespace sdfthip;
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  for (const ArenaEntry& e : g_arenas)
    if (e.window == window) { if (out) *out = e.info; return 0; }
  set_error("sdft_hip_matrix_placement", "not a window of sdft_hip_malloc_matrix_in_arena");
  return -1;
}
int sdft_hip_free_matrix(void* window)
{
  using namespace sdfthip;
  void* base = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    for (size_t i = 0; i < g_arenas.size(); ++i)
      if (g_arenas[i].window == window) { base = g_arenas[i].base; g_arenas.erase(g_arenas.begin() + (long)i); break; }
  }
  if (!base) { set_error("sdft_hip_free_matrix", "not a window of sdft_hip_malloc_matrix_in_arena"); return -1; }
  if (hipFree(base) != hipSuccess) { (void)hipGetLastError(); set_error("sdft_hip_free_matrix", "hipFree failed"); return -1; }
  return 0;
}

// NULL when no error has been recorded on this thread since the last clear
const char* sdft_hip_last_error(void) { return sdfthip::g_has_error ? sdfthip::g_error.c_str() : nullptr; }
void sdft_hip_clear_error(void) { sdfthip::g_has_error = false; sdfthip::g_error.clear(); }
// NULL when no call on this thread has left a warning since the last clear (a warning never changes a return code)
const char* sdft_hip_last_warning(void) { return sdfthip::g_has_warning ? sdfthip::g_warning.c_str() : nullptr; }
void sdft_hip_clear_warning(void) { sdfthip::g_has_warning = false; sdfthip::g_warning.clear(); }

int sdft_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
int sdft_hip_set_device(int device)
{
  const hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { sdfthip::set_error("hipSetDevice", hipGetErrorString(e)); return -1; }
  return 0;
}
int sdft_hip_get_device(void)
{
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return d;
}
const char* sdft_hip_version(void) { return "sdft-hip 0.1 (gfx950)"; }
int sdft_hip_selftest(void) { return sdfthip::lane_selftest() ? 0 : -1; }
// compiles the statements of a sdft_hip_op_expr operation for `arch` (NULL: gfx950) without running them: 0, or -1 with the
// compiler's words in sdft_hip_last_error().  Needs no GPU -- a host can check its expressions where it is built.
int sdft_hip_check_expr(const char* expr, const char* arch)
{
  if (!expr || !*expr) { sdfthip::set_error("sdft_hip_check_expr", "no statements"); return -1; }
  std::string lowered; std::vector<char> code;
  const char* target = arch && *arch ? arch : "gfx950";
  // (both bin types on the two-pass route, the fused kernel of the headline shape, the row synthesis of a hop)
  return sdfthip::rtc_compile(expr, "sdfthip::user_rows_kernel<double>", target, lowered, code) &&
         sdfthip::rtc_compile(expr, "sdfthip::user_rows_kernel<float>", target, lowered, code) &&
         sdfthip::rtc_compile(expr, "sdfthip::forward_rows_kernel<double, 1, 1, true, 1, 1, true, float, false>", target, lowered, code) &&
         sdfthip::rtc_compile(expr, "sdfthip::inverse_row_kernel<float, double, true, true>", target, lowered, code) ? 0 : -1;
}

}  // extern "C"
