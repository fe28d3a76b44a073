// C-ABI instantiation: time domain double, frequency domain float (reference macros SDFT_TD_* / SDFT_FD_*, sdft.h:21-37)
#define SDFT_TD double
#define SDFT_FD float
#define SDFT_SUFFIX f64f32
#include "sdft_capi.inc"
