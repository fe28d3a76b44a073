/*
 * sdft_oracle.c -- CPU restatement of the modulated Sliding DFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sdft_amd/ may import, link or call
 * this file; it is the checker for the HIP path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg) and never the thing that is shipped or measured.
 *
 * Parity status: PINNED.  tests/test_oracle.py proves this restatement
 * bit-identical to the reference C header compiled from /root/reference
 * (oracle/Makefile -> oracle/_ref/) and to the golden vectors that build wrote
 * into tests/golden/ (tests/golden/make_golden.py).
 *
 * What is restated (all citations into /root/reference/c/src/sdft/sdft.h):
 *   plan constants and tables ........ :413-450
 *   reset ............................ :517-529
 *   per-sample analysis recurrence ... :562-598   (delay line :186-191,:564)
 *   spectral window convolution ...... :350-402
 *   per-row synthesis ................ :635-657
 *   row loops ........................ :607-613, :666-672
 *
 * The restatement is organised differently from the reference on purpose (the
 * delay line is kept in time order, the halo is produced by an index
 * reflection instead of a padded scratch row, the window is applied by a
 * 5-point gather) but every floating point operation is applied to the same
 * operands in the same order, so results are bit-identical when compiled
 * without FMA contraction (-ffp-contract=off) -- see oracle/Makefile.
 *
 * Build-time parameters:  -DORACLE_TD=float|double  -DORACLE_FD=float|double
 */

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORACLE_TD
#define ORACLE_TD float
#endif
#ifndef ORACLE_FD
#define ORACLE_FD double
#endif

typedef ORACLE_TD td_t;
typedef ORACLE_FD fd_t;
typedef struct { fd_t re, im; } cx_t;   /* layout == C99 complex == std::complex (:84-99) */

enum { WIN_BOXCAR = 0, WIN_HANN = 1, WIN_HAMMING = 2, WIN_BLACKMAN = 3 };   /* :127-133 */

typedef struct oracle_plan
{
  size_t  nbins;       /* N = dftsize                                         */
  int     window;
  double  latency;
  fd_t    aweight;     /* 1/(2N)            :422                               */
  fd_t    sweight;     /* 2                 :423                               */
  cx_t*   tw;          /* analysis twiddles :444                               */
  cx_t*   syn;         /* synthesis twiddles:445                               */
  /* stream state */
  td_t*   ring;        /* last 2N samples, ring[(head+i)%2N] = x[t-2N+i]       */
  size_t  head;
  size_t  phase;       /* == reference cursor (:153), samples seen mod 2N      */
  cx_t*   acc;         /* accoutput :157                                       */
  cx_t*   rot;         /* fiddles   :159                                       */
  cx_t*   demod;       /* demodulated spectrum of the current sample (aux)     */
  fd_t    ghost[4];    /* only used when N == 1: stale halo cells, see below   */
} oracle_plan;

/* fd-typed libm, selected by the size of fd_t (:193-213, :333-348) */
static fd_t fd_cos (fd_t a) { return sizeof(fd_t) == sizeof(float) ? (fd_t)cosf((float)a)  : (fd_t)cos((double)a);  }
static fd_t fd_sin (fd_t a) { return sizeof(fd_t) == sizeof(float) ? (fd_t)sinf((float)a)  : (fd_t)sin((double)a);  }
static fd_t fd_acos(fd_t a) { return sizeof(fd_t) == sizeof(float) ? (fd_t)acosf((float)a) : (fd_t)acos((double)a); }

static cx_t cx(fd_t re, fd_t im) { cx_t z; z.re = re; z.im = im; return z; }
/* struct-complex formulas of the reference's SDFT_NO_COMPLEX_H mode (:265-331) */
static cx_t cx_add (cx_t a, cx_t b) { return cx(a.re + b.re, a.im + b.im); }
static cx_t cx_sub (cx_t a, cx_t b) { return cx(a.re - b.re, a.im - b.im); }
static cx_t cx_mul (cx_t a, cx_t b) { return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static cx_t cx_scale(cx_t a, fd_t s) { return cx(a.re * s, a.im * s); }
static cx_t cx_conj(cx_t a) { return cx(a.re, -a.im); }

oracle_plan* oracle_new(size_t nbins, int window, double latency)
{
  oracle_plan* p = (oracle_plan*)calloc(1, sizeof(oracle_plan));
  p->nbins   = nbins;
  p->window  = window;
  p->latency = latency;
  p->aweight = (fd_t)(1) / (nbins * 2);                                   /* :422 */
  p->sweight = (fd_t)(2);                                                 /* :423 */
  p->tw    = (cx_t*)calloc(nbins ? nbins : 1, sizeof(cx_t));
  p->syn   = (cx_t*)calloc(nbins ? nbins : 1, sizeof(cx_t));
  p->ring  = (td_t*)calloc(nbins ? nbins * 2 : 1, sizeof(td_t));
  p->acc   = (cx_t*)calloc(nbins ? nbins : 1, sizeof(cx_t));
  p->rot   = (cx_t*)calloc(nbins ? nbins : 1, sizeof(cx_t));
  p->demod = (cx_t*)calloc(nbins ? nbins : 1, sizeof(cx_t));

  /* :439-446 -- note the mixed fd_t/double arithmetic: `omega * nbins` and
     `omega * k * nbins` are fd_t products, `* latency` promotes to double and
     the argument is narrowed back to fd_t at the cos/sin call. */
  const fd_t omega = (fd_t)(-2) * fd_acos((fd_t)(-1)) / (nbins * 2);
  const fd_t gain  = (fd_t)(+2) / ((fd_t)(1) - fd_cos((fd_t)(omega * nbins * latency)));
  for (size_t k = 0; k < nbins; ++k)
  {
    const fd_t a = omega * k;
    const fd_t s = (fd_t)(omega * k * nbins * latency);
    p->tw[k]  = cx((fd_t)(1) * fd_cos(a), (fd_t)(1) * fd_sin(a));
    p->syn[k] = cx(gain * fd_cos(s), gain * fd_sin(s));
    p->rot[k] = cx(1, 0);
  }
  return p;
}

void oracle_free(oracle_plan* p)
{
  if (!p) return;
  free(p->tw); free(p->syn); free(p->ring); free(p->acc); free(p->rot); free(p->demod);
  free(p);
}

void oracle_reset(oracle_plan* p)                                          /* :517-529 */
{
  p->head = 0;
  p->phase = 0;
  memset(p->ring, 0, p->nbins * 2 * sizeof(td_t));
  memset(p->acc, 0, p->nbins * sizeof(cx_t));
  memset(p->demod, 0, p->nbins * sizeof(cx_t));
  memset(p->ghost, 0, sizeof(p->ghost));
  for (size_t k = 0; k < p->nbins; ++k) p->rot[k] = cx(1, 0);
}

/*
 * Spectrum with halo.  The reference mirrors two cells on each side of a padded
 * row (:589-595): X[-i] = conj X[i], X[N-1+i] = conj X[N-1-i], i = 1, 2, applied
 * in that order so that a mirrored cell may itself be the source of a later
 * mirror.  For N >= 2 this is an iterated reflection about bin 0 and bin N-1,
 * each reflection toggling a conjugation.  For N == 1 both mirrors read the
 * opposite halo cell *before* it is refreshed, i.e. last sample's value; with a
 * zeroed plan those cells therefore stay zero for ever -- kept here as `ghost`
 * so that the restatement follows the reference even in that corner.
 */
static cx_t halo_at(const oracle_plan* p, ptrdiff_t k)
{
  const ptrdiff_t n = (ptrdiff_t)p->nbins;
  if (n == 1 && k != 0)
    return cx(p->ghost[0], p->ghost[1]);    /* always zero, see above */
  int flip = 0;
  while (k < 0 || k > n - 1)
  {
    k = (k < 0) ? -k : 2 * (n - 1) - k;
    flip ^= 1;
  }
  return flip ? cx_conj(p->demod[k]) : p->demod[k];
}

static void window_row(const oracle_plan* p, cx_t* out)                    /* :350-402 */
{
  const fd_t w = p->aweight;
  for (ptrdiff_t k = 0; k < (ptrdiff_t)p->nbins; ++k)
  {
    const cx_t c = halo_at(p, k);
    switch (p->window)
    {
      case WIN_HANN:
      {
        const cx_t a = cx_add(c, c);
        const cx_t b = cx_add(halo_at(p, k - 1), halo_at(p, k + 1));
        out[k] = cx_scale(cx_sub(a, b), w * (fd_t)(0.25));
        break;
      }
      case WIN_HAMMING:
      {
        const cx_t a = cx_scale(c, (fd_t)(0.54));
        const cx_t b = cx_scale(cx_add(halo_at(p, k - 1), halo_at(p, k + 1)), (fd_t)(0.23));
        out[k] = cx_scale(cx_sub(a, b), w);
        break;
      }
      case WIN_BLACKMAN:
      {
        const cx_t a = cx_scale(c, (fd_t)(0.42));
        const cx_t b = cx_scale(cx_add(halo_at(p, k - 1), halo_at(p, k + 1)), (fd_t)(0.25));
        const cx_t d = cx_scale(cx_add(halo_at(p, k - 2), halo_at(p, k + 2)), (fd_t)(0.04));
        out[k] = cx_scale(cx_add(cx_sub(a, b), d), w);
        break;
      }
      default:
        out[k] = cx_scale(c, w);
        break;
    }
  }
}

/* One analysis step (:562-598): returns nothing, leaves the demodulated spectrum
   in p->demod and writes the windowed row to `out`. */
static void analyse_one(oracle_plan* p, td_t x, cx_t* out)
{
  const size_t span = p->nbins * 2;
  if (span == 0) return;

  /* delay line: subtraction in TD precision, then widened to FD (:564) */
  const td_t old = p->ring[p->head];
  p->ring[p->head] = x;
  p->head = (p->head + 1 == span) ? 0 : p->head + 1;
  const fd_t delta = x - old;

  const int wrap = (p->phase >= span - 1);                                /* :566 */
  p->phase = wrap ? 0 : p->phase + 1;

  for (size_t k = 0; k < p->nbins; ++k)
  {
    p->acc[k] = cx_add(p->acc[k], cx_scale(p->rot[k], delta));           /* :572,:583 */
    if (wrap)
    {
      p->rot[k]   = cx(1, 0);                                             /* :573 */
      p->demod[k] = p->acc[k];                                            /* :574 */
    }
    else
    {
      p->rot[k]   = cx_mul(p->rot[k], p->tw[k]);                          /* :584 */
      p->demod[k] = cx_mul(p->acc[k], cx_conj(p->rot[k]));                /* :585 */
    }
  }
  window_row(p, out);
}

void oracle_sdft_n(oracle_plan* p, size_t n, const td_t* x, cx_t* rows)    /* :607-613 */
{
  for (size_t t = 0; t < n; ++t)
    analyse_one(p, x[t], rows + t * p->nbins);
}

static td_t synthesise_one(const oracle_plan* p, const cx_t* row)          /* :635-657 */
{
  fd_t s = (fd_t)(0);
  if (p->latency == 1)
  {
    for (size_t k = 0; k < p->nbins; ++k)
      s += row[k].re * (k % 2 ? -1 : +1);
  }
  else
  {
    for (size_t k = 0; k < p->nbins; ++k)
      s += cx_mul(row[k], p->syn[k]).re;
  }
  s *= p->sweight;
  return (td_t)(s);
}

void oracle_isdft_n(const oracle_plan* p, size_t n, const cx_t* rows, td_t* y)  /* :666-672 */
{
  for (size_t t = 0; t < n; ++t)
    y[t] = synthesise_one(p, rows + t * p->nbins);
}

/*
 * Streaming digest for full-size parity checks: runs analysis (and synthesis)
 * over n samples without materialising the (n, N) matrix.  Per row it emits
 *   digest[4t+0] = sum_k re,  [4t+1] = sum_k im,  [4t+2] = sum_k (re^2+im^2),
 *   digest[4t+3] = sum_k (k+1) * re      (position-sensitive)
 * accumulated in double in bin order, and y[t] if y != NULL.
 */
void oracle_digest_n(oracle_plan* p, size_t n, const td_t* x, double* digest, td_t* y)
{
  cx_t* row = (cx_t*)malloc((p->nbins ? p->nbins : 1) * sizeof(cx_t));
  for (size_t t = 0; t < n; ++t)
  {
    analyse_one(p, x[t], row);
    double sr = 0, si = 0, sp = 0, sk = 0;
    for (size_t k = 0; k < p->nbins; ++k)
    {
      const double re = (double)row[k].re, im = (double)row[k].im;
      sr += re; si += im; sp += re * re + im * im; sk += (double)(k + 1) * re;
    }
    digest[4 * t + 0] = sr; digest[4 * t + 1] = si; digest[4 * t + 2] = sp; digest[4 * t + 3] = sk;
    if (y) y[t] = synthesise_one(p, row);
  }
  free(row);
}

/* introspection for the tests */
size_t oracle_size(const oracle_plan* p)  { return p ? p->nbins : 0; }
size_t oracle_phase(const oracle_plan* p) { return p->phase; }
size_t oracle_sizeof_td(void) { return sizeof(td_t); }
size_t oracle_sizeof_fd(void) { return sizeof(fd_t); }
void oracle_tables(const oracle_plan* p, cx_t* tw, cx_t* syn, fd_t* weights)
{
  memcpy(tw,  p->tw,  p->nbins * sizeof(cx_t));
  memcpy(syn, p->syn, p->nbins * sizeof(cx_t));
  weights[0] = p->aweight;
  weights[1] = p->sweight;
}
void oracle_state(const oracle_plan* p, cx_t* acc, cx_t* rot, td_t* hist)
{
  memcpy(acc, p->acc, p->nbins * sizeof(cx_t));
  memcpy(rot, p->rot, p->nbins * sizeof(cx_t));
  for (size_t i = 0; i < p->nbins * 2; ++i)           /* time order, oldest first */
    hist[i] = p->ring[(p->head + i) % (p->nbins * 2)];
}
