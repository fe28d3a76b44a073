// sdft_common.hip -- untyped part of the C-ABI: error channel, device selection, self test.

#include "sdft_plan.hpp"

#include <mutex>

namespace sdfthip {

static thread_local std::string g_error;
static thread_local bool g_has_error = false;

void set_error(const char* what, const char* detail)
{
  g_error = std::string(what ? what : "?") + ": " + (detail ? detail : "?");
  g_has_error = true;
}

__global__ void lane_selftest_kernel(int* out)
{
  const int lane = threadIdx.x;
  out[lane] = lane_from_below(lane + 100);
  out[64 + lane] = lane_from_above(lane + 100);
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
  SDFT_TRY(hipMalloc((void**)&d, 128 * sizeof(int)));
  hipLaunchKernelGGL(lane_selftest_kernel, dim3(1), dim3(64), 0, 0, d);
  int h[128];
  hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { set_error("lane_selftest", hipGetErrorString(e)); return false; }
  bool ok = true;
  for (int l = 1; l < 64; ++l) ok = ok && (h[l] == l - 1 + 100);
  for (int l = 0; l < 63; ++l) ok = ok && (h[64 + l] == l + 1 + 100);
  state = ok ? 1 : -1;
  if (!ok) set_error("lane_selftest", "DPP wave shift semantics mismatch");
  return ok;
}

}  // namespace sdfthip

extern "C" {

// NULL when no error has been recorded on this thread since the last clear
const char* sdft_hip_last_error(void) { return sdfthip::g_has_error ? sdfthip::g_error.c_str() : nullptr; }
void sdft_hip_clear_error(void) { sdfthip::g_has_error = false; sdfthip::g_error.clear(); }

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

}  // extern "C"
