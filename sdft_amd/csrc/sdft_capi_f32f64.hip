// C-ABI instantiation: time domain float, frequency domain double (reference macros SDFT_TD_* / SDFT_FD_*, sdft.h:21-37)
#define SDFT_TD float
#define SDFT_FD double
#define SDFT_SUFFIX f32f64
#include "sdft_capi.inc"
