// host_bench.cpp -- a C++ host in the shape of the reference's cpp/examples/bench.cpp:13-49:
// construct SDFT<T, F>(dftsize), analyse n samples into a std::vector<std::complex<F>> matrix,
// synthesise back.  Input from a raw file, outputs to raw files; tests/test_gpu_chost.py compares
// them with the oracle.  Also exercises the single-sample and row-vector overloads.
//
// usage: host_bench <dftsize> <window 0..3> <latency> <x.raw> <y.raw> <dfts.raw>

#include <sdft/sdft.h>

#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef HOST_T
#define HOST_T float
#endif
#ifndef HOST_F
#define HOST_F double
#endif

using sdft::SDFT;

int main(int argc, char* argv[])
{
  if (argc < 7) return 2;
  const size_t dftsize = (size_t)atol(argv[1]);
  const sdft::Window window = static_cast<sdft::Window>(atoi(argv[2]));
  const double latency = atof(argv[3]);

  FILE* f = fopen(argv[4], "rb");
  if (!f) return 3;
  fseek(f, 0, SEEK_END);
  const size_t n = (size_t)ftell(f) / sizeof(HOST_T);
  fseek(f, 0, SEEK_SET);
  std::vector<HOST_T> x(n), y(n);
  if (fread(x.data(), sizeof(HOST_T), n, f) != n) return 3;
  fclose(f);

  SDFT<HOST_T, HOST_F> sdft(dftsize, window, latency);
  if (sdft.size() != dftsize || sdft.window() != window || sdft.latency() != latency) return 4;
  const size_t m = sdft.size();
  std::vector<std::complex<HOST_F>> dfts(n * m);

  const size_t half = n / 2;
  sdft.sdft(half, x.data(), dfts.data());                       // dense overload
  std::vector<std::complex<HOST_F>*> rows(n - half - 1);
  for (size_t r = 0; r < rows.size(); ++r) rows[r] = dfts.data() + (half + r) * m;
  sdft.sdft(rows.size(), x.data() + half, rows.data());         // row-vector overload
  sdft.sdft(x[n - 1], dfts.data() + (n - 1) * m);               // single-sample overload

  sdft.isdft(n - 1, dfts.data(), y.data());
  y[n - 1] = sdft.isdft(dfts.data() + (n - 1) * m);

  f = fopen(argv[5], "wb"); fwrite(y.data(), sizeof(HOST_T), n, f); fclose(f);
  f = fopen(argv[6], "wb"); fwrite(dfts.data(), sizeof(std::complex<HOST_F>), n * m, f); fclose(f);

  sdft.reset();
  std::vector<std::complex<HOST_F>> again(m);
  sdft.sdft(x[0], again.data());
  if (memcmp(again.data(), dfts.data(), m * sizeof(std::complex<HOST_F>)) != 0) return 5;
  printf("CPP-HOST ok n=%zu m=%zu\n", n, m);
  return 0;
}
