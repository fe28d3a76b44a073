/*
 * sdft/sdft.hpp -- C++ facade `sdft::SDFT<T, F>` over the C-ABI of libsdft_hip.so.
 *
 * Mirrors the public interface of the reference's C++ header (cpp/src/sdft/sdft.h:29-255:
 * enum class Window :34-40, constructor :61, reset :97, size/window/latency :109-128,
 * sdft :135/:179/:193, isdft :205/:235/:249) so that code written against
 * `sdft::SDFT<float, double>` compiles unchanged; the work is done by the HIP kernels behind
 * the drop-in C functions of sdft/sdft.h.  `-I include/cpp` makes `#include <sdft/sdft.h>`
 * resolve to this facade for C++ hosts (the reference uses the same file name in its cpp tree).
 *
 * T (time domain) and F (frequency domain) may be float or double; long double has no GPU
 * representation and fails to compile.  std::complex<F> is layout-compatible with the C-ABI's
 * interleaved complex.  Pointers may be host or device pointers, as in the C API.
 */

#pragma once

#include <complex>
#include <cstddef>
#include <stdexcept>
#include <string>

extern "C" {
const char* sdft_hip_last_error(void);
#define SDFT_HPP_DECLARE(SUF, TD)                                                                   \
  void* sdft_hip_alloc_custom_##SUF(std::size_t dftsize, int window, double latency);               \
  void sdft_hip_free_##SUF(void* plan);                                                             \
  void sdft_hip_reset_##SUF(void* plan);                                                            \
  void sdft_hip_sdft_##SUF(void* plan, TD sample, void* dft);                                       \
  void sdft_hip_sdft_n_##SUF(void* plan, std::size_t n, const TD* samples, void* dfts);             \
  void sdft_hip_sdft_nd_##SUF(void* plan, std::size_t n, const TD* samples, void** dfts);           \
  TD sdft_hip_isdft_##SUF(void* plan, const void* dft);                                             \
  void sdft_hip_isdft_n_##SUF(void* plan, std::size_t n, const void* dfts, TD* samples);            \
  void sdft_hip_isdft_nd_##SUF(void* plan, std::size_t n, const void** dfts, TD* samples);
SDFT_HPP_DECLARE(f32f64, float)
SDFT_HPP_DECLARE(f32f32, float)
SDFT_HPP_DECLARE(f64f64, double)
SDFT_HPP_DECLARE(f64f32, double)
#undef SDFT_HPP_DECLARE
}

namespace sdft
{
  /** Supported SDFT analysis window types (reference cpp/src/sdft/sdft.h:34-40). */
  enum class Window
  {
    Boxcar,
    Hann,
    Hamming,
    Blackman
  };

  namespace detail
  {
    template <typename T, typename F> struct abi;   // undefined for unsupported type pairs
#define SDFT_HPP_ABI(SUF, TD, FD)                                                                   \
    template <> struct abi<TD, FD>                                                                  \
    {                                                                                               \
      static void* alloc(std::size_t n, int w, double l) { return sdft_hip_alloc_custom_##SUF(n, w, l); } \
      static void free(void* p) { sdft_hip_free_##SUF(p); }                                         \
      static void reset(void* p) { sdft_hip_reset_##SUF(p); }                                       \
      static void sdft(void* p, TD x, void* d) { sdft_hip_sdft_##SUF(p, x, d); }                    \
      static void sdft_n(void* p, std::size_t n, const TD* x, void* d) { sdft_hip_sdft_n_##SUF(p, n, x, d); } \
      static void sdft_nd(void* p, std::size_t n, const TD* x, void** d) { sdft_hip_sdft_nd_##SUF(p, n, x, d); } \
      static TD isdft(void* p, const void* d) { return sdft_hip_isdft_##SUF(p, d); }                \
      static void isdft_n(void* p, std::size_t n, const void* d, TD* y) { sdft_hip_isdft_n_##SUF(p, n, d, y); } \
      static void isdft_nd(void* p, std::size_t n, const void** d, TD* y) { sdft_hip_isdft_nd_##SUF(p, n, d, y); } \
    };
    SDFT_HPP_ABI(f32f64, float, double)
    SDFT_HPP_ABI(f32f32, float, float)
    SDFT_HPP_ABI(f64f64, double, double)
    SDFT_HPP_ABI(f64f32, double, float)
#undef SDFT_HPP_ABI
  }

  /**
   * Sliding Discrete Fourier Transform (SDFT) on the GPU.
   * @tparam T Time domain data type: float (default) or double.
   * @tparam F Frequency domain data type: float or double (default and recommended).
   **/
  template <typename T = float, typename F = double>
  class SDFT
  {
    using api = detail::abi<T, F>;

  public:

    /** Creates a new SDFT plan (reference :61). Throws if the GPU cannot be set up. */
    SDFT(const std::size_t dftsize, const Window window = Window::Hann, const double latency = 1) :
      dftsize_(dftsize), window_(window), latency_(latency),
      plan_(api::alloc(dftsize, static_cast<int>(window), latency))
    {
      if (!plan_)
      {
        const char* e = sdft_hip_last_error();
        throw std::runtime_error(std::string("sdft::SDFT: ") + (e ? e : "plan allocation failed"));
      }
    }

    ~SDFT() { api::free(plan_); }

    SDFT(const SDFT&) = delete;
    SDFT& operator=(const SDFT&) = delete;
    SDFT(SDFT&& other) noexcept :
      dftsize_(other.dftsize_), window_(other.window_), latency_(other.latency_), plan_(other.plan_)
    {
      other.plan_ = nullptr;
    }

    /** Resets this SDFT plan instance to its initial state (reference :97). */
    void reset() { api::reset(plan_); }

    /** Returns the assigned number of DFT bins (reference :109). */
    std::size_t size() const { return dftsize_; }

    /** Returns the assigned analysis window type (reference :117). */
    Window window() const { return window_; }

    /** Returns the assigned synthesis latency factor (reference :125). */
    double latency() const { return latency_; }

    /** Estimates the DFT vector for the given sample (reference :135). */
    void sdft(const T sample, std::complex<F>* const dft) { api::sdft(plan_, sample, dft); }

    /** Estimates the DFT matrix (nsamples, dftsize) for the given sample array (reference :179). */
    void sdft(const std::size_t nsamples, const T* samples, std::complex<F>* const dfts)
    {
      api::sdft_n(plan_, nsamples, samples, dfts);
    }

    /** Same with an array of DFT row vectors (reference :193). */
    void sdft(const std::size_t nsamples, const T* samples, std::complex<F>** const dfts)
    {
      api::sdft_nd(plan_, nsamples, samples, reinterpret_cast<void**>(dfts));
    }

    /** Synthesizes a single sample from the given DFT vector (reference :205). */
    T isdft(const std::complex<F>* dft) { return api::isdft(plan_, dft); }

    /** Synthesizes the sample array from the given DFT matrix (reference :235). */
    void isdft(const std::size_t nsamples, const std::complex<F>* dfts, T* const samples)
    {
      api::isdft_n(plan_, nsamples, dfts, samples);
    }

    /** Same with an array of DFT row vectors (reference :249). */
    void isdft(const std::size_t nsamples, const std::complex<F>** dfts, T* const samples)
    {
      api::isdft_nd(plan_, nsamples, reinterpret_cast<const void**>(dfts), samples);
    }

    /** The underlying C-ABI plan (sdft_t*), for the additions declared in sdft/sdft_hip.h. */
    void* native_handle() const { return plan_; }

  private:

    std::size_t dftsize_;
    Window window_;
    double latency_;
    void* plan_;
  };
}
