// sdft_host_io.hpp -- the caller's HOST memory: pointer classification, device buffers of the plan, host buffers registered in
// place (option "host_register"), and the copies between pageable host memory and the device through pinned slots of the plan
// (sdft_copy_engine.hpp: a few host threads, so that the host's copy of one piece overlaps the DMA of the next).
// Host side of libsdft_hip.so; used by Plan<TD, FD> (sdft_plan.hpp).  Citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include <hip/hip_runtime.h>

#include <chrono>

#include "sdft_copy_engine.hpp"

namespace sdfthip {

void set_error(const char* what, const char* detail);   // sdft_common.hip

#ifndef SDFT_TRY
#define SDFT_TRY(expr)                                                        \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { set_error(#expr, hipGetErrorString(e_)); return false; } \
  } while (0)
#endif

static inline bool is_device_pointer(const void* p)
{
  if (!p) return false;
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  const hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}


template <typename T> struct DevBuf
{
  T* p = nullptr;
  size_t cap = 0;
  bool reserve(size_t count)
  {
    if (count <= cap) return true;
    // allocate first, free afterwards: a failed growth leaves the old buffer usable
    T* q = nullptr;
    SDFT_TRY(hipMalloc((void**)&q, count * sizeof(T)));
    if (p) (void)hipFree(p);
    p = q; cap = count;
    return true;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// everything about the caller's host memory that does not depend on the plan's types
class HostIo
{
 public:
  // ---- host buffers, mapped in place ------------------------------------------------------------------------------
  // A host of the reference hands malloc'ed buffers to every call and reuses them hop after hop
  // (/root/reference/test/test.c:62-83).  Copying through the runtime's pageable path costs such a hop-sized call more
  // than its kernels (round 1: 191 us for a 1.6 MB hop); registering the caller's buffer once (hipHostRegister: the
  // pages are pinned and mapped, the driver follows the mapping with MMU notifiers) lets the kernels read and write it
  // over PCIe directly -- no staging copy, one synchronisation.  Buffers of 1 MiB and more only (smaller ones share
  // pages with their heap neighbours), at most 8 ranges that never share a page; anything the runtime refuses
  // falls back to the staged path.  Option "host_register" = 1 turns it on, a buffer of more than 256 MiB is never registered: the
  // bytes of one buffer (default 256 MiB: longer calls run at PCIe speed through the staged path anyway).
  static constexpr size_t kSmallHostBytes = (size_t)64 << 10;
  static constexpr size_t kHostRegisterMin = (size_t)1 << 20;       // smaller buffers share pages with their heap neighbours: staged
  // OFF by default: a registration dies with the mapping it was made on.  A host that frees a buffer and gets the same
  // address back from its allocator (numpy does, every call) would make a remembered registration fault the GPU
  // (measured: "Memory access fault" on the second call) -- the driver does not re-attach it.  A C host that allocates
  // its buffers once, like the reference's driver, sets option "host_register" = 1.
  long opt_host_register = 0;
  static constexpr size_t kHostRegisterMax = (size_t)256 << 20;
  // A registration covers exactly the caller's bytes [a, b) -- NOT the whole pages around them: a long-lived process gets
  // its megabyte buffers from the heap (glibc raises its mmap threshold as buffers are freed), where the neighbours share
  // the first and last page; with whole pages registered, a later copy from or to such a neighbour -- inside the
  // registered range at one end, outside at the other -- fails in the runtime ("invalid argument": seen in the test
  // suite, never in a fresh process).  Our own registrations still never share a page with each other ([plo, phi) are the
  // page ranges: two registrations sharing a page would lose it when the first of them is dropped).
  struct HostReg { uintptr_t a, b, plo, phi; char* dev; unsigned long long used; bool owned, writable; };
  HostReg host_regs[8] = {};
  unsigned long long host_reg_clock = 0;
  long host_reg_hits = 0, host_reg_misses = 0;
  void drop_host(HostReg& e)
  {
    if (e.b && e.owned) { (void)hipHostUnregister(reinterpret_cast<void*>(e.a)); (void)hipGetLastError(); }
    e = HostReg{};
  }
  void forget_host_buffers() { for (HostReg& e : host_regs) drop_host(e); }
  // device-side address of a host buffer of `bytes` bytes, or nullptr (not used / not possible: take the staged path)
  void* map_host(const void* p, size_t bytes, bool will_write = false)
  {
    if (!opt_host_register || !p || bytes < kHostRegisterMin || bytes > kHostRegisterMax) return nullptr;
    const uintptr_t page = 4096, a = reinterpret_cast<uintptr_t>(p), b = a + bytes;
    const uintptr_t plo = a & ~(page - 1), phi = (b + page - 1) & ~(page - 1);
    for (HostReg& e : host_regs)
      if (e.b && a >= e.a && b <= e.b)
      {
        if (will_write && !e.writable && e.owned) { drop_host(e); break; }   // registered for reading: again, with its pages made writable first
        e.used = ++host_reg_clock; ++host_reg_hits; return e.dev + (a - e.a);
      }
    // anything of ours that shares a page with the new range goes first
    for (HostReg& e : host_regs)
      if (e.b && plo < e.phi && e.plo < phi) drop_host(e);
    HostReg* slot = &host_regs[0];
    for (HostReg& e : host_regs) { if (!e.b) { slot = &e; break; } if (e.used < slot->used) slot = &e; }
    drop_host(*slot);
    ++host_reg_misses;
    void* dev = nullptr;
    // memory the host pinned itself (hipHostMalloc, its own hipHostRegister) is mapped already
    if (hipHostGetDevicePointer(&dev, const_cast<void*>(p), 0) == hipSuccess && dev)
    {
      *slot = HostReg{a, b, plo, phi, static_cast<char*>(dev), ++host_reg_clock, false, true};
      return dev;
    }
    (void)hipGetLastError();
    if (will_write)
    {
      // an output buffer the host has never written (calloc, numpy.zeros) may still be mapped to the kernel's shared zero
      // page, copy on write: every page gets its own writable frame BEFORE it is pinned (writing a byte back to itself --
      // the buffer is ours to overwrite for the duration of the call), so that what the device writes is what the host reads
      volatile char* q = reinterpret_cast<volatile char*>(a);
      for (uintptr_t off = 0; off < bytes; off += page) q[off] = q[off];
      q[bytes - 1] = q[bytes - 1];
    }
    if (hipHostRegister(reinterpret_cast<void*>(a), bytes, hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipHostGetDevicePointer(&dev, reinterpret_cast<void*>(a), 0) != hipSuccess || !dev)
    {
      (void)hipGetLastError(); (void)hipHostUnregister(reinterpret_cast<void*>(a)); (void)hipGetLastError(); return nullptr;
    }
    *slot = HostReg{a, b, plo, phi, static_cast<char*>(dev), ++host_reg_clock, true, will_write};
    return dev;
  }

  // ---- copies between the caller's host memory and the device ---------------------------------------------------
  // The runtime's own path for pageable memory PINS the caller's pages once a copy exceeds its threshold (about 1 MiB) and
  // remembers the pin per stream, keyed by address and size.  A host that frees such a buffer, lets the heap shrink and
  // later gets the address back (numpy does; so does any long-lived process) makes the next copy find a pin whose pages
  // left the process in between -- the driver does not re-attach it and the copy kernel faults the GPU ("Write access to a
  // read-only page", the process is gone; seen in this library's own test suite about one run in three, round 4).  So
  // nothing of the caller's is ever handed to the runtime: every copy goes through pinned 2 MiB pieces of the plan (DMA of
  // one piece while the host copies the other) -- until round 6 copies of up to 64 KiB went through the runtime's staging
  // buffers; see to_device below.  scripts/pageable_copy_probe.hip, profiles/r04_host_copy_paths.txt: every piece costs ~15 us
  // of its own (1.6 MB in pieces of 128 KiB: 200 us), the host's copy out of pinned memory the device has just written
  // runs at 42 GB/s with streaming stores (host_copy_bytes; glibc memcpy 30); 1.6 MB as one piece: ~70 us out, ~50 us in,
  // against 37 us each way on a pin the runtime remembered;
  // long copies 26 against 55 GB/s.  A hop-sized matrix (up to both pieces, 4 MiB) skips the DMA: the kernels write or
  // read the pinned pieces themselves over PCIe (sdft_n / isdft_n below; option "host_direct" = 0 turns that off).
  // Option "host_copy" = 1 hands everything to the runtime (a host that allocates its buffers once and keeps them, like
  // the reference's driver, loses nothing by it).
  // Both are complete on return as far as the caller's memory goes: to_device has read it, to_host has written it.
  // Round 5: the pieces are a ring of four slots and the host's copies run on two worker threads of the plan (option
  // "copy_threads"; sdft_copy_engine.hpp): the copy of one piece overlaps the DMA of the next, long copies move at the DMA
  // engine's pace instead of one core's (26 -> ~50 GB/s), and a hop-sized matrix is copied by the workers and the caller together.
  static constexpr size_t kPinPiece = (size_t)2 << 20;
  static constexpr unsigned kPinSlots = 4;
  // long copies: every DMA costs ~15 us of its own beside its bytes (2 MiB pieces: 53 us each = 39.5 GB/s, measured with
  // two and with three copy threads alike), so copies beyond 32 MiB travel in pieces of 8 MiB through a second, larger ring
  static constexpr size_t kBigPiece = (size_t)8 << 20;
  static constexpr size_t kBigCopy = (size_t)32 << 20;
  char* h_big = nullptr;
  static constexpr size_t kDirectBytes = 2 * kPinPiece;      // a hop-sized matrix the kernels write / read in the pinned memory itself
  long opt_host_copy = 0;
  long opt_host_direct = 1;                                  // hop-sized matrices: the kernels work on the pinned pieces themselves
  long opt_copy_threads = 2;                                 // worker threads of the host copies (0: the calling thread alone)
  char* h_pin = nullptr;
  char* d_pin = nullptr;                                     // the same memory as the kernels see it
  long pin_copies = 0;
  double pin_us_memcpy = 0, pin_us_device = 0;               // where a staged call's time went: the host's memcpy, waiting for the device
  static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  // the device side of the copy engine: DMAs on the plan's stream, an event per slot
  // Copies of 16 MiB and more alternate between the plan's stream and a stream of their own: every DMA costs ~15 us beside
  // its bytes (completion of one, start of the next), and two queues fill each other's gaps: 1.6 GB 50 -> 54.5 GB/s (the
  // runtime's own path: 55), 31 MiB 925 -> 810 us, 19 MiB 600 -> 555; no gain below (profiles/r05_host_copy_streams.txt).
  // The second stream starts behind everything the plan's stream holds (fork) and the plan's stream continues behind the
  // second one's copies (join): to the work before and after it, the copy is still one operation on the plan's stream.
  struct HipDev
  {
    hipStream_t stream = nullptr, aux = nullptr, last = nullptr;
    hipEvent_t ev[kPinSlots] = {};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool recorded[kPinSlots] = {};
    bool two = false;                                        // this copy uses both streams
    unsigned turn = 0;
    hipStream_t next_stream() { last = (two && (turn++ & 1u)) ? aux : stream; return last; }
    bool dma_to_device(void* dst, const void* slot, size_t len) { SDFT_TRY(hipMemcpyAsync(dst, slot, len, hipMemcpyHostToDevice, next_stream())); return true; }
    bool dma_to_host(void* slot, const void* src, size_t len) { SDFT_TRY(hipMemcpyAsync(slot, src, len, hipMemcpyDeviceToHost, next_stream())); return true; }
    bool record(unsigned slot) { SDFT_TRY(hipEventRecord(ev[slot], last ? last : stream)); recorded[slot] = true; return true; }   // (right after the slot's DMA, same thread)
    bool wait(unsigned slot) { if (recorded[slot]) SDFT_TRY(hipEventSynchronize(ev[slot])); return true; }      // (any thread)
    bool fork(hipStream_t s, bool both)
    {
      stream = s; last = s; turn = 0; two = false;
      if (!both) return true;
      if (!aux)
      {
        hipStream_t ns = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess)
        {
          (void)hipGetLastError();
          if (e0) (void)hipEventDestroy(e0);
          if (ns) (void)hipStreamDestroy(ns);
          return true;                                       // (one stream then)
        }
        aux = ns; ev_fork = e0; ev_join = e1;
      }
      SDFT_TRY(hipEventRecord(ev_fork, stream));
      SDFT_TRY(hipStreamWaitEvent(aux, ev_fork, 0));
      two = true;
      return true;
    }
    bool join()
    {
      if (!two) return true;
      two = false;
      SDFT_TRY(hipEventRecord(ev_join, aux));
      SDFT_TRY(hipStreamWaitEvent(stream, ev_join, 0));
      return true;
    }
    void destroy_aux()
    {
      if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); aux = nullptr; }
      if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
      if (ev_join) { (void)hipEventDestroy(ev_join); ev_join = nullptr; }
    }
  };
  long opt_copy_streams = 2;                                 // copies from 16 MiB on: 2 = alternate between the plan's stream and a second one, 1 = the plan's stream only
  static constexpr size_t kTwoStreamBytes = (size_t)16 << 20;
  HipDev dev;
  CopyPool pool;

  void free_pin_pages()
  {
    if (h_pin) (void)hipHostFree(h_pin);
    if (h_big) (void)hipHostFree(h_big);
    h_pin = nullptr; d_pin = nullptr; h_big = nullptr;
  }
  // the slots a copy of `bytes` bytes travels through: the 2 MiB ring, or (long copies) the 8 MiB ring, allocated on first use
  char* slots_for(size_t bytes, size_t& piece)
  {
    piece = kPinPiece;
    if (bytes < kBigCopy) return h_pin;
    if (!h_big)
    {
      if (hipHostMalloc((void**)&h_big, kPinSlots * kBigPiece, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h_big = nullptr; return h_pin; }
      if (!pin_idle()) return h_pin;                         // (the two rings share the slots' events)
    }
    piece = kBigPiece;
    return h_big;
  }
  bool ensure_pin()
  {
    if (h_pin) return true;
    if (hipHostMalloc((void**)&h_pin, kPinSlots * kPinPiece, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h_pin = nullptr; return false; }
    if (hipHostGetDevicePointer((void**)&d_pin, h_pin, 0) != hipSuccess || !d_pin) { (void)hipGetLastError(); free_pin_pages(); return false; }
    for (unsigned k = 0; k < kPinSlots; ++k)
      if (hipEventCreateWithFlags(&dev.ev[k], hipEventDisableTiming) != hipSuccess)
      {
        (void)hipGetLastError();
        for (unsigned j = 0; j < k; ++j) (void)hipEventDestroy(dev.ev[j]);
        for (unsigned j = 0; j < kPinSlots; ++j) dev.ev[j] = nullptr;
        free_pin_pages(); return false;
      }
    return true;
  }
  void release_pin()
  {
    pool.stop();
    for (unsigned k = 0; k < kPinSlots; ++k)
    {
      if (dev.ev[k]) { if (dev.recorded[k]) (void)hipEventSynchronize(dev.ev[k]); (void)hipEventDestroy(dev.ev[k]); dev.ev[k] = nullptr; }
      dev.recorded[k] = false;
    }
    dev.destroy_aux();
    free_pin_pages();
    (void)hipGetLastError();
  }
  // no DMA of an earlier copy still reads or writes the pinned memory (before the kernels use it directly)
  bool pin_idle()
  {
    for (unsigned k = 0; k < kPinSlots; ++k) { if (!dev.wait(k)) return false; dev.recorded[k] = false; }
    return true;
  }
  CopyPool* workers()
  {
    if (opt_copy_threads <= 0) return nullptr;
    if (pool.workers() == 0 && !pool.start((unsigned)std::min<long>(opt_copy_threads, 8))) return nullptr;
    return &pool;
  }
  // the host's own copy into / out of pinned memory the kernels work on directly (one piece: shared by the workers and the caller)
  void copy_bytes(void* dst, const void* src, size_t bytes) { parallel_copy(bytes >= ((size_t)512 << 10) ? workers() : nullptr, dst, src, bytes); }

  // Both are complete on return as far as the caller's memory goes: to_device has read it, to_host has written it.
  bool to_device(void* dst, const void* src, size_t bytes, hipStream_t stream)
  {
    if (bytes == 0) return true;
    // (small copies too, since round 6: the runtime looks a pageable address up among the pins it remembers BEFORE it decides how to copy, so a few
    // hundred bytes of samples at an address some earlier large copy of the process -- not this library's -- once had pinned went by that stale
    // pin, and the copy faulted the GPU: "Write access to a read-only page", one run of the test suite in four on some hosts)
    if (opt_host_copy == 1 || !ensure_pin())
    {
      SDFT_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
      return true;
    }
    ++pin_copies;
    size_t piece; char* mem = slots_for(bytes, piece);
    if (!dev.fork(stream, opt_copy_streams >= 2 && bytes >= kTwoStreamBytes)) return false;
    PieceCopier<HipDev> copier(dev, bytes > piece ? workers() : nullptr, mem, piece, kPinSlots);
    const bool ok = copier.to_device(dst, src, bytes);       // the pieces still in flight are waited for before their slots' next use
    return dev.join() && ok;                                 // (what follows on the plan's stream follows the second stream's copies too)
  }
  bool to_host(void* dst, const void* src, size_t bytes, hipStream_t stream)
  {
    if (bytes == 0) return true;
    if (opt_host_copy == 1 || !ensure_pin())
    {
      SDFT_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
      return true;
    }
    ++pin_copies;
    size_t piece; char* mem = slots_for(bytes, piece);
    if (!dev.fork(stream, opt_copy_streams >= 2 && bytes >= kTwoStreamBytes)) return false;
    PieceCopier<HipDev> copier(dev, bytes > piece ? workers() : nullptr, mem, piece, kPinSlots);
    const bool ok = copier.to_host(dst, src, bytes);         // (every piece drained: nothing left on either stream)
    return dev.join() && ok;
  }
};

}  // namespace sdfthip
