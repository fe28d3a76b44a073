// C-ABI instantiation: time domain float, frequency domain float (reference macros SDFT_TD_* / SDFT_FD_*, sdft.h:21-37)
#define SDFT_TD float
#define SDFT_FD float
#define SDFT_SUFFIX f32f32
#include "sdft_capi.inc"
