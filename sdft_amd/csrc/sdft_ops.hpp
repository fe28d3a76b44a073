// sdft_ops.hpp -- spectral operations of the fused call (gain, shift, gate, power law, the host's own statements) and the synthesis term
// Part of the kernel source of libsdft_hip.so (see sdft_kernels.hpp); citations are into /root/reference/c/src/sdft/sdft.h.

#pragma once

#include "sdft_forward_hop.hpp"

#pragma clang fp contract(off)

namespace sdfthip {

// ------------------------------------------------------------------------------------------
// Spectral operation between analysis and synthesis (fused path): what a host of the
// reference does to the (n, N) matrix between sdft_sdft_n and sdft_isdft_n (README.md:42-47),
// applied per bin while the row is in flight.
//   identity            X'_k = X_k
//   gain  g[N] (real)   X'_k = X_k * g_k          (complex times real: both parts scaled)
//   shift s (bins)      X'_k = X_{k-s}, zero where k-s falls outside [0, N)
// synth_term returns what sdft_isdft adds for SOURCE bin k (sdft.h:643 / :650).  The reference adds
// output bins in ascending order; a shift keeps source bins in the same order, and the bins it
// empties add +-0, which never changes a running sum that started at +0.
// ------------------------------------------------------------------------------------------
//   gate  (thr, floor)  X'_k = X_k if |X_k| >= thr, else X_k * floor            (not linear: windowed rows only)
//   power (p, scale)    X'_k = X_k * scale * |X_k|^(p-1), i.e. |X'_k| = scale * |X_k|^p with the phase kept
// Gains may change with time: `rows` gain vectors, row r for the call's samples [r*hop, (r+1)*hop), the last one for
// everything after it (what a host does when it recomputes its mask every hop; README.md:42-47 leaves that loop to it).
//   user  (expression)   X'_k = whatever the host's statements leave in (re, im): compiled at run time (user_op below)
enum : int { OP_IDENTITY = 0, OP_GAIN = 1, OP_SHIFT = 2, OP_CGAIN = 3, OP_GATE = 4, OP_POWER = 5, OP_USER = 6 };
template <typename FD> struct SpectralOp
{
  int kind;
  const FD* gain;             // OP_GAIN: [rows][N] real factors; OP_CGAIN: [rows][N] complex factors (re, im interleaved)
  long shift;                 // OP_SHIFT
  unsigned rows;              // gain vectors (<= 1: one for the whole call)
  size_t hop;                 // samples per gain vector
  size_t t0;                  // index, within the host's call, of the first row a launch sees (two-pass segments)
  FD p0, p1;                  // OP_GATE: threshold, floor; OP_POWER: exponent, scale
  FD pv[8];                   // OP_USER: up to eight parameters travel with the kernel arguments (more: `gain` points at them)
};
template <typename FD> SDFT_HD bool op_is_linear(int kind) { return kind <= OP_CGAIN; }
// the operation a kernel serves: the library's own build dispatches on SpectralOp::kind at run time; a run-time
// compilation (the host's statements) is for one operation, and every other branch leaves the code
#ifndef SDFT_FIXED_OP
#define SDFT_FIXED_OP -1
#endif
template <typename FD> SDFT_D int op_kind_of(const SpectralOp<FD>& op) { return SDFT_FIXED_OP >= 0 ? SDFT_FIXED_OP : op.kind; }
template <int V> struct OpTag { static constexpr int value = V; };   // an operation known where the code is generated (-1: not)
// the gain vector of row t of the launch
template <typename FD> SDFT_D const FD* gain_row(const SpectralOp<FD>& op, size_t t, unsigned nbins)
{
  if (op.rows <= 1 || !op.gain) return op.gain;
  size_t r = (op.t0 + t) / op.hop;
  if (r >= op.rows) r = op.rows - 1;
  return op.gain + r * (size_t)nbins * (op.kind == OP_CGAIN ? 2u : 1u);
}
// ... walked forward in time (the row-group kernels): one division at the start, additions afterwards
template <typename FD> struct GainCursor
{
  const FD* g; size_t next, hop, stride; unsigned left;     // next: launch-relative time at which the next vector starts
  SDFT_D void start(const SpectralOp<FD>& op, size_t t, unsigned nbins)
  {
    g = op.gain; next = ~(size_t)0; hop = op.hop; left = 0; stride = (size_t)nbins * (op.kind == OP_CGAIN ? 2u : 1u);
    if (op.rows <= 1 || !op.gain || (op.kind != OP_GAIN && op.kind != OP_CGAIN)) return;
    size_t r = (op.t0 + t) / op.hop;
    if (r >= op.rows) r = op.rows - 1;
    g = op.gain + r * stride;
    left = op.rows - 1 - (unsigned)r;
    if (left) next = (r + 1) * op.hop - op.t0;
  }
  SDFT_D void seek(size_t t)                               // t never decreases
  {
    while (left && t >= next) { g += stride; --left; next = left ? next + hop : ~(size_t)0; }
  }
};
// x^h for a positive, finite, normal double x: exp(h * ln x) with both functions written out -- ln x = e*ln2 + 2*atanh(z),
// z = (r - 1)/(r + 1) for the mantissa r in [sqrt(1/2), sqrt(2)), a polynomial of degree 10 in z^2; exp by k = rint(t/ln2),
// a Taylor polynomial of degree 13 on |s| <= ln2/2 and one v_ldexp_f64.  About 50 fp64 instructions and a dozen registers
// (the library's log and exp, which also serve arguments this caller never has, take three times both: the power law at
// N = 2048, where the kernel has no registers to spare, 18.8 -> x ms).  Relative error of the result: 2e-16 * (1 + |h ln x|).
// a double constant in a scalar register pair at the point of use (two s_mov_b32): left to itself the compiler keeps the 25
// polynomial coefficients below in 50 vector registers for the whole kernel -- and spills them
SDFT_D double scalar_const(double c) { asm volatile("" : "+s"(c)); return c; }
SDFT_D double pow_positive(double x, double h)
{
  const long long bits = __double_as_longlong(x);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  double r = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);      // [1, 2)
  if (r > 1.4142135623730951) { r *= 0.5; ++e; }
  // (no IEEE division: v_rcp_f64 and two Newton steps -- the divide expansion costs 15 instructions and two mode switches)
  const double den = r + 1.0;
  double inv = __builtin_amdgcn_rcp(den);
  inv = __builtin_fma(__builtin_fma(-den, inv, 1.0), inv, inv);
  inv = __builtin_fma(__builtin_fma(-den, inv, 1.0), inv, inv);
  const double z = (r - 1.0) * inv, w = z * z;
  double q = scalar_const(1.0 / 21.0);
  q = __builtin_fma(q, w, scalar_const(1.0 / 19.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 17.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 15.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 13.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 11.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 9.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 7.0)); q = __builtin_fma(q, w, scalar_const(1.0 / 5.0));
  q = __builtin_fma(q, w, scalar_const(1.0 / 3.0)); q = __builtin_fma(q, w, 1.0);
  const double ln = __builtin_fma((double)e, scalar_const(0.6931471805599453), 2.0 * z * q);
  const double t = h * ln;
  if (t > 709.0) return __builtin_huge_val();
  if (t < -745.0) return 0.0;
  const double k = __builtin_rint(t * scalar_const(1.4426950408889634));
  const double sred = __builtin_fma(-k, scalar_const(1.9082149292705877e-10), __builtin_fma(-k, scalar_const(0.6931471803691238), t));   // ln2 = hi + lo
  double p = scalar_const(1.0 / 6227020800.0);
  p = __builtin_fma(p, sred, scalar_const(1.0 / 479001600.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 39916800.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 3628800.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 362880.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 40320.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 5040.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 720.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 120.0));
  p = __builtin_fma(p, sred, scalar_const(1.0 / 24.0)); p = __builtin_fma(p, sred, scalar_const(1.0 / 6.0));
  p = __builtin_fma(p, sred, 0.5); p = __builtin_fma(p, sred, 1.0); p = __builtin_fma(p, sred, 1.0);
  return __builtin_ldexp(p, (int)k);
}

// the operations that are not linear in the spectrum, on one windowed bin
template <typename FD> SDFT_D cx<FD> op_pointwise(cx<FD> v, const SpectralOp<FD>& op, int kind)
{
  if (kind == OP_GATE)
  {
    const FD mag2 = v.re * v.re + v.im * v.im;
    return (mag2 < op.p0 * op.p0) ? cscale(v, op.p1) : v;
  }
  if (kind == OP_POWER)
  {
    const FD mag2 = v.re * v.re + v.im * v.im;
    // (FD float: |v| below 1e-19 -- a denormal square, which v_log_f32 would flush -- counts as zero)
    // (FD double: a square below the smallest normal double likewise -- |v| < 1.5e-154)
    if (!(mag2 > (sizeof(FD) == 8 ? (FD)2.2250738585072014e-308 : (FD)1.17549435e-38f))) return cmake<FD>((FD)0, (FD)0);
    if (!(mag2 < (FD)__builtin_huge_val())) return v;                                    // infinities and NaNs pass through
    // |v|^(p-1) = exp((p-1)/2 * ln |v|^2): mag2 is positive and finite here, so none of pow()'s case analysis is needed
    // (a third of its instructions and registers; 1e-15 / 1e-6 of the factor at FD double / float, the float one
    // through v_log_f32 / v_exp_f32)
    const FD h = (op.p0 - (FD)1) * (FD)0.5;
    FD f;
    if constexpr (sizeof(FD) == 8) f = op.p1 * pow_positive(mag2, h);
    else f = op.p1 * __builtin_amdgcn_exp2f(h * __builtin_amdgcn_logf(mag2));
    return cscale(v, f);
  }
  return v;
}

// the host's own operation (sdft_hip_process_n with sdft_hip_op_expr): its statements are the text of the header "sdft_user_expr.inc" of the
// run-time compilation, which defines SDFT_USER_EXPR; the library's own build has no such operation.
// In scope: re, im (sdft_fd_t, read and assign: the windowed value of bin k), k, nbins (unsigned), t (size_t: sample index
// within the call), ch (size_t: channel), p (const sdft_fd_t*: the call's parameters, device memory), and HIP's math.
#ifdef SDFT_USER_EXPR
// p[i]: the call's parameters -- out of the kernel arguments (up to eight: no copy, no launch in front of the kernel; a
// pageable 8-byte hipMemcpyAsync in front of every hop cost a synchronous host 110 us) or out of device memory through
// the constant address space (scalar loads that nothing in the kernel can alias, so they leave the loop)
template <typename FD> struct UserParams
{
  const FD* small; const SDFT_CONSTANT FD* big;
  SDFT_D FD operator[](size_t i) const { return big ? big[i] : small[i]; }
};
template <typename FD> SDFT_D cx<FD> user_op(cx<FD> v, unsigned k, unsigned nbins, size_t t, size_t ch, const SpectralOp<FD>& op)
{
  const UserParams<FD> p{op.pv, op.gain ? as_uniform(op.gain) : nullptr};
  typedef FD sdft_fd_t;
  FD re = v.re, im = v.im;
  {
#include "sdft_user_expr.inc"
  }
  return cmake<FD>(re, im);
}
// rows[ch][t][k] = user_op(rows[ch][t][k]): the two-pass route (rows that no workgroup holds, one-chunk calls)
template <typename FD>
__global__ __launch_bounds__(256) void user_rows_kernel(cx<FD>* mat, size_t stride, size_t rows, unsigned nbins, unsigned channels, SpectralOp<FD> op)
{
  const size_t per = rows * nbins, total = per * channels;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
  {
    const size_t ch = i / per, r = i - ch * per;
    const size_t t = r / nbins, k = r - t * nbins;
    cx<FD>* q = mat + ch * stride + r;
    *q = user_op(*q, (unsigned)k, nbins, op.t0 + t, ch, op);
  }
}
#endif

// grow: the gain vector of the row v belongs to (gain_row / GainCursor); unused by the other operations
template <typename FD, bool LAT1, bool OPS>
SDFT_D FD synth_term(cx<FD> v, unsigned k, const SpectralOp<FD>& op, const cx<FD>* syn, unsigned nbins, const FD* grow = nullptr)
{
  long ko = (long)k;                                       // output bin whose sign / twiddle applies
  if constexpr (OPS)                                       // (plain sdft_isdft_n instantiates without the checks)
  {
    if (op.kind == OP_GAIN) v = cscale(v, grow[k < nbins ? k : 0]);
    else if (op.kind == OP_CGAIN) v = cmul(v, reinterpret_cast<const cx<FD>*>(grow)[k < nbins ? k : 0]);
    else if (op.kind == OP_SHIFT)
    {
      ko += op.shift;
      if (ko < 0 || ko >= (long)nbins) return (FD)0;
    }
    else if (op.kind >= OP_GATE) v = op_pointwise(v, op, op.kind);
  }
  if constexpr (LAT1) return v.re * ((ko & 1) ? (FD)(-1) : (FD)(+1));               // sdft.h:643
  else { const cx<FD> sy = syn[ko < (long)nbins ? ko : 0]; return v.re * sy.re - v.im * sy.im; }   // re of :650
}

// Fused analysis -> operation -> synthesis (forward_rows_kernel with SYN != 0): the rows never leave the
// workgroup unless `store` asks for a copy of the processed spectrum.
template <typename TD, typename FD> struct FuseArgs
{
  TD* y;                      // [channels][n]
  size_t y_stride;
  const cx<FD>* syn;          // [N]
  FD sweight;
  SpectralOp<FD> op;
  int store;                  // also write the processed rows to ForwardArgs::out
  unsigned* walked;           // SYN = 2, float samples: counts the samples whose sum had to be walked in order (or nullptr)
};

}  // namespace sdfthip
