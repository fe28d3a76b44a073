// doorbell_probe.hip -- development probe (round 6): how long does a word take from the host to a resident kernel and back?
//   (a) the doorbell in pinned HOST memory, polled by the kernel over PCIe (what resident_hop_kernel does), one lane / sixteen lanes per poll;
//   (b) the doorbell in DEVICE memory that the host can write (fine-grained allocation, written by the CPU through the BAR), polled by the kernel locally.
// The kernel answers by writing the value to an ack word in pinned host memory; the host measures the round trip.  (b) may not be possible at all:
// each variant runs in a child process so that a fault in one does not hide the others.
// hipcc --offload-arch=gfx950 -O2 -w scripts/doorbell_probe.hip -o scripts/bin/doorbell_probe ; scripts/bin/doorbell_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

__global__ void echo_kernel(unsigned* bell, unsigned* ack, unsigned rounds, int lanes)
{
  const int lane = threadIdx.x;
  for (unsigned want = 1; want <= rounds; ++want)
  {
    unsigned long long spins = 0;
    for (;;)
    {
      unsigned v = lane < lanes ? __hip_atomic_load(bell + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
      v = (unsigned)__builtin_amdgcn_readlane((int)v, lanes - 1);
      if (v == want) break;
      if (++spins > (1ull << 26)) return;                   // bounded
    }
    if (lane == 0) __hip_atomic_store(ack, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static int run(int variant)
{
  const unsigned rounds = 2000;
  unsigned* bell_h = nullptr; unsigned* bell_d = nullptr; unsigned* ack_h = nullptr; unsigned* ack_d = nullptr;
  if (hipHostMalloc((void**)&ack_h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return 2;
  hipHostGetDevicePointer((void**)&ack_d, ack_h, 0);
  memset(ack_h, 0, 64);
  int lanes = 1;
  const char* what = "";
  if (variant == 0 || variant == 1)
  {
    hipHostMalloc((void**)&bell_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void**)&bell_d, bell_h, 0);
    lanes = variant == 0 ? 1 : 16;
    what = variant == 0 ? "pinned host memory, one lane polls" : "pinned host memory, sixteen lanes read the line";
  }
  else if (variant == 2)
  {
    if (hipExtMallocWithFlags((void**)&bell_d, 4096, hipDeviceMallocFinegrained) != hipSuccess) { printf("fine-grained device memory: allocation failed\n"); return 3; }
    bell_h = bell_d; lanes = 16; what = "fine-grained DEVICE memory written by the CPU, sixteen lanes";
  }
  else if (variant == 3)
  {
    if (hipMalloc((void**)&bell_d, 4096) != hipSuccess) return 3;
    hipMemset(bell_d, 0, 4096); hipDeviceSynchronize();
    bell_h = bell_d; lanes = 16; what = "plain hipMalloc DEVICE memory written by the CPU, sixteen lanes";
  }
  else if (variant == 4)
  {
    if (hipMallocManaged((void**)&bell_d, 4096) != hipSuccess) return 3;
    hipMemAdvise(bell_d, 4096, hipMemAdviseSetPreferredLocation, 0);
    hipMemAdvise(bell_d, 4096, hipMemAdviseSetAccessedBy, hipCpuDeviceId);
    bell_h = bell_d; lanes = 16; what = "managed memory preferred on the device, sixteen lanes";
  }
  volatile unsigned* bell = bell_h;
  for (int i = 0; i < 16; ++i) bell[i] = 0;                  // (a fault here ends the child)
  hipStream_t s; hipStreamCreate(&s);
  hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, s, bell_d, ack_d, rounds, lanes);
  std::vector<double> us;
  volatile unsigned* ack = ack_h;
  usleep(2000);
  for (unsigned r = 1; r <= rounds; ++r)
  {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 15; ++i) bell[i] = r;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    bell[lanes - 1] = r;
    unsigned long long spins = 0;
    while (*ack != r) { if (++spins > (1ull << 30)) { printf("%s: no answer at round %u\n", what, r); return 4; } }
    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  hipStreamSynchronize(s);
  std::sort(us.begin(), us.end());
  printf("%-72s round trip median %.2f us  p10 %.2f  p90 %.2f\n", what, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
  return 0;
}

int main(int argc, char** argv)
{
  if (argc > 1) return run(atoi(argv[1]));
  for (int v = 0; v < 5; ++v)
  {
    fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) { execl(argv[0], argv[0], std::to_string(v).c_str(), (char*)nullptr); _exit(9); }
    int st = 0; waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("variant %d: the child died with signal %d (the CPU cannot write that memory)\n", v, WTERMSIG(st));
    else if (WEXITSTATUS(st) != 0) printf("variant %d: exit code %d\n", v, WEXITSTATUS(st));
  }
  return 0;
}
