// C-ABI instantiation: time domain double, frequency domain double (reference macros SDFT_TD_* / SDFT_FD_*, sdft.h:21-37)
#define SDFT_TD double
#define SDFT_FD double
#define SDFT_SUFFIX f64f64
#include "sdft_capi.inc"
