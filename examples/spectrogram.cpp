// spectrogram.cpp -- C++ host in the shape of the reference's cpp/examples/analysis.cpp (without the
// plotting): sdft::SDFT<float, double> over a chirp, prints the strongest bin every 4000 samples.
//
//   make -C examples && ./examples/spectrogram

#include <sdft/sdft.h>      // resolves to include/cpp/sdft/sdft.h -> sdft/sdft.hpp

#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

int main()
{
  const size_t sr = 48000, n = 48000, m = 1000;
  std::vector<float> x(n);
  double phi = 0;
  for (size_t i = 0; i < n; ++i)
  {
    const double f = (double)i / n * sr / 4;            // 0 -> 12 kHz
    phi += 2.0 * 3.14159265358979323846 * f / sr;
    x[i] = (float)std::sin(phi);
  }
  sdft::SDFT<float, double> sdft(m, sdft::Window::Hann, 1);
  std::vector<std::complex<double>> dfts(n * m);
  sdft.sdft(n, x.data(), dfts.data());
  for (size_t t = 3999; t < n; t += 4000)
  {
    size_t best = 0;
    for (size_t k = 1; k < m; ++k)
      if (std::abs(dfts[t * m + k]) > std::abs(dfts[t * m + best])) best = k;
    std::printf("t=%6zu  peak bin %4zu  ~%7.1f Hz  (instantaneous sweep frequency %7.1f Hz, window centre ~%zu samples earlier)\n",
                t, best, (double)best * sr / (2.0 * m), (double)t / n * sr / 4, m);
  }
  return 0;
}
