/*
 * ref_shim.c -- translation unit that turns the *genuine* reference header into a
 * shared library for the oracle harness.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference (c/src/sdft/sdft.h) defines all of its functions, non-static,
 * in the header, so including it once from here and compiling with
 *   -I$(SDFT_REF_DIR)/c/src -DSDFT_NO_COMPLEX_H [-DSDFT_TD_*] [-DSDFT_FD_*]
 * exports sdft_alloc_custom / sdft_sdft_n / sdft_isdft_n / ... unchanged.  No
 * reference source is copied into this repository: the header is read where it
 * lies (oracle/Makefile), and the product goes to oracle/_ref/ (git-ignored).
 *
 * SDFT_NO_COMPLEX_H is mandatory with gcc: in <complex.h> mode gcc drops the
 * imaginary part of the 2-element compound literal at sdft.h:245-248 and the
 * build is numerically wrong.
 *
 * The accessors below are ours; they only read public fields of the plan
 * struct (sdft.h:145-182) so that the tests can compare tables and state.
 */
#include <sdft/sdft.h>

const sdft_fdx_t* ref_analysis_twiddles(const sdft_t* p)  { return p->analysis.twiddles; }
const sdft_fdx_t* ref_synthesis_twiddles(const sdft_t* p) { return p->synthesis.twiddles; }
const sdft_fdx_t* ref_accoutput(const sdft_t* p)          { return p->analysis.accoutput; }
const sdft_fdx_t* ref_fiddles(const sdft_t* p)            { return p->analysis.fiddles; }
const sdft_td_t*  ref_input(const sdft_t* p)              { return p->analysis.input; }
size_t            ref_cursor(const sdft_t* p)             { return p->analysis.cursor; }
double            ref_analysis_weight(const sdft_t* p)    { return (double)p->analysis.weight; }
double            ref_synthesis_weight(const sdft_t* p)   { return (double)p->synthesis.weight; }
size_t            ref_sizeof_td(void)                     { return sizeof(sdft_td_t); }
size_t            ref_sizeof_fd(void)                     { return sizeof(sdft_fd_t); }
