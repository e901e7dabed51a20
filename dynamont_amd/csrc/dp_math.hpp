// dp_math.hpp -- fp64 arithmetic of one lattice cell, shared by the HIP kernels and by a
// host-side accuracy test (tests/test_dp_math.py compiles it with g++).
//
// Reference arithmetic being reproduced (relative to /root/reference):
//   log_normal_pdf  src/cpp/aligner.cpp:287-292   -0.5*z*z - log(stdev) - 0.5*log(2*pi), z = (x-mean)/stdev
//   logPlus         src/cpp/aligner.cpp:276-285   hi + log1p(exp(lo - hi)), -inf operands pass through
//
// ROCm's ocml log1p(double) alone is ~135 VALU instructions (exp ~42); one logPlus per cell per
// pass is the dominant cost of the whole path, so softplus g(d) = log1p(exp(d)), d <= 0, is
// evaluated here in ~45 fp64 instructions with |error| <= ~1.5e-16 absolute (<= ~1 ulp of the
// values 0.35..0.69 it returns most often), i.e. the accuracy class of glibc's exp+log1p:
//   exp:   d = k ln2 + r, |r| <= ln2/2, exp(r) = 1 + r + r^2 Q9(r)              (Q9: degree 9)
//   log1p: e in [0,1]; e <= sqrt2-1: 2 atanh(e/(2+e)); else ln2 + 2 atanh((e-1)/(e+3));
//          atanh(t) = t + t s F6(s), s = t^2 <= 0.0295                           (F6: degree 6)
// Coefficients: Chebyshev-node interpolation in 60-digit arithmetic (mpmath), rounded to fp64.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DYN_HD __host__ __device__ __forceinline__
#else
#define DYN_HD inline
#endif

namespace dynmath {

constexpr double NEG_INF = -__builtin_huge_val();

DYN_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1/x to ~1 ulp for x in [2, 4]: hardware seed + two Newton steps (same structure the compiler
// uses for fp64 division, minus scaling/fixup that this range never needs).
DYN_HD double rcp_seed(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(x);
#else
  return (double)(1.0f / (float)x);
#endif
}

DYN_HD double div_2to4(double num, double den) {
  double r = rcp_seed(den);
  double e = fma_(-den, r, 1.0);
  r = fma_(r, e, r);
  e = fma_(-den, r, 1.0);
  r = fma_(r, e, r);
  double q = num * r;
  double rem = fma_(-den, q, num);
  return fma_(rem, r, q);
}

// exp(d) for d in [-1000, 0] (callers clamp). Underflows gracefully to denormal/0 via ldexp.
DYN_HD double exp_nonpos(double d) {
  const double LOG2E = 0x1.71547652b82fep+0;
  const double LN2_HI = 0x1.62e42fee00000p-1;
  const double LN2_LO = 0x1.a39ef35793c76p-33;
  double kf = __builtin_rint(d * LOG2E);
  double r = fma_(-kf, LN2_HI, d);
  r = fma_(-kf, LN2_LO, r);
  double q = 0x1.af390ba7e6f47p-26;
  q = fma_(q, r, 0x1.2891d4ffbb0f9p-22);
  q = fma_(q, r, 0x1.71de0d85293b8p-19);
  q = fma_(q, r, 0x1.a019b8ca26fcfp-16);
  q = fma_(q, r, 0x1.a01a01a7cebcdp-13);
  q = fma_(q, r, 0x1.6c16c1789d1d7p-10);
  q = fma_(q, r, 0x1.11111111109a6p-7);
  q = fma_(q, r, 0x1.5555555553d37p-5);
  q = fma_(q, r, 0x1.5555555555556p-3);
  q = fma_(q, r, 0x1.0000000000001p-1);
  double p = fma_(r * r, q, r) + 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ldexp(p, (int)kf);
#else
  return std::ldexp(p, (int)kf);
#endif
}

// log1p(e) for e in [0, 1]. c = 1 selects the upper branch (e > sqrt2 - 1).
DYN_HD double log1p_unit(double e) {
  const double SQRT2M1 = 0x1.a827999fcef32p-2;
  const double LN2 = 0x1.62e42fefa39efp-1;
  const double c = (e > SQRT2M1) ? 1.0 : 0.0;
  const double num = e - c;
  const double den = (e + 2.0) + c;
  const double t = div_2to4(num, den);
  const double s = t * t;
  double f = 0x1.2b6686d1072f3p-4;
  f = fma_(f, s, 0x1.39fd39474ad34p-4);
  f = fma_(f, s, 0x1.7462bd8e53c17p-4);
  f = fma_(f, s, 0x1.c71c62c63e016p-4);
  f = fma_(f, s, 0x1.2492492e09d1ap-3);
  f = fma_(f, s, 0x1.9999999995204p-3);
  f = fma_(f, s, 0x1.5555555555558p-2);
  const double a = fma_(t * s, f, t);  // atanh(t)
  return fma_(c, LN2, a + a);
}

// softplus for d <= 0 (NaN and -inf map to 0 through the clamp).
DYN_HD double softplus_nonpos(double d) {
  d = __builtin_fmax(d, -1000.0);
  return log1p_unit(exp_nonpos(d));
}

// aligner.cpp:276-285. -inf operands: d = -inf or NaN -> clamp -> g = +0 -> hi + 0 == hi, so
// logPlus(x, -inf) == x and logPlus(-inf, -inf) == -inf exactly, as in the reference.
DYN_HD double log_plus(double x, double y) {
  const double hi = __builtin_fmax(x, y);
  const double lo = __builtin_fmin(x, y);
  return hi + softplus_nonpos(lo - hi);
}

// Per-k-mer emission constants kept with each band slot.
struct Emis {
  double mean;
  double inv_stdev;  // 1/stdev rounded to nearest
  double log_norm;   // -log(stdev) - 0.5*log(2*pi); std::log evaluated on the host by the libm the reference calls
  double stdev;      // only for log_normal_pdf_exact (host tests)
};

constexpr double HALF_LOG_2PI = 0x1.d67f1c864beb4p-1;  // 0.5*log(2*pi) = 0.91893853320467274178

// aligner.cpp:287-292 with the reference's evaluation order: z = diff/stdev formed as diff*inv + one
// FMA residual correction (= the correctly rounded quotient, Markstein), then
// (-0.5*z*z - log(stdev)) - 0.5*log(2*pi). Bit-identical to the CPU expression (tests/test_dp_math.py);
// 8 fp64 operations. Kept as the yardstick for the 4-operation form the kernels run.
DYN_HD double log_normal_pdf_exact(double x, double mean, double stdev, double log_stdev) {
  const double inv = 1.0 / stdev;
  const double diff = x - mean;
  double z = diff * inv;
  const double rem = fma_(-z, stdev, diff);
  z = fma_(rem, inv, z);
  double t = -0.5 * z;
  t = t * z;
  t = t - log_stdev;
  return t - HALF_LOG_2PI;
}

// The kernels' emission: z = diff * (1/stdev), e = fma(-0.5*z, z, log_norm): 4 fp64 operations per
// cell instead of 8. Differs from the expression above by a few ulp of |z^2/2| + |log_norm| (the
// quotient is not corrected and the two constants are pre-added); the chip is power-limited on fp64
// operations, and the differences are 1e-16-level, like those of the softplus.
DYN_HD double log_normal_pdf(double x, const Emis& p) {
  const double z = (x - p.mean) * p.inv_stdev;
  return fma_(-0.5 * z, z, p.log_norm);
}

DYN_HD Emis make_emis(double mean, double stdev, double log_stdev) {
  Emis e;
  e.mean = mean;
  e.inv_stdev = 1.0 / stdev;
  e.log_norm = -log_stdev - HALF_LOG_2PI;
  e.stdev = stdev;
  return e;
}

// ---------------------------------------------------------------------------------------------
// "Vector across cells" forms. A wave owns M independent lattice cells per lane; each stage of
// the arithmetic is written as a loop over the M cells so that the M dependency chains are
// interleaved in program order. A single wave then issues back-to-back fp64 instructions
// (measured: ~4.8 cycles/instr at 8 independent chains vs 8.6 cycles for one dependent chain,
// tools/ubench/issue_rate.hip) instead of M serial Horner chains.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Table-driven softplus. g(d) = log1p(exp(d)) satisfies g1 = s (the logistic function) and every
// higher derivative is a polynomial in s:  with u = s(1-s), w = 1-2s
//   g2 = u,  g3 = u w,  g4 = u (1-6u),  g5 = u w (1-12u),  |g6| <= 1/4      (gk = k-th derivative)
// A table of (g, s) at the nodes d_i = -i/128, i = 0..5120 (81 936 bytes, one copy in the LDS of
// each CU) plus the degree-5 Taylor polynomial about the NEAREST node (|r| <= 1/256) gives
// |error| <= 1/4 (1/256)^6 / 720 = 1.2e-18 + the rounding of g_i and of the last FMA, i.e. the
// same <= ~1 ulp class as the polynomial form above at 24 instead of 45 fp64 instructions.
// d <= -40 (incl. -inf and NaN, clamped) maps to the last node, which holds (0, 0): g = 0 exactly.
// ---------------------------------------------------------------------------------------------
struct SoftplusNode {
  double g, s;
};
constexpr int SP_STEPS = 128;                   // nodes per unit of d
constexpr int SP_RANGE = 40;                    // table covers d in [-40, 0]
constexpr int SP_NODES = SP_STEPS * SP_RANGE + 1;

// Host-side generator (long double arithmetic, rounded once to double).
inline void softplus_build_table(SoftplusNode* t) {
  for (int i = 0; i < SP_NODES; ++i) {
    const long double d = -(long double)i / SP_STEPS;
    const long double e = expl(d);
    t[i].g = (i == SP_NODES - 1) ? 0.0 : (double)log1pl(e);
    t[i].s = (i == SP_NODES - 1) ? 0.0 : (double)(e / (1.0L + e));
  }
}

template <int M>
DYN_HD void softplus_table_vec(double (&d)[M], double (&g)[M], const SoftplusNode* __restrict__ tab) {
  double kf[M], r[M], g0[M], s[M], u[M], w[M], uw[M], p[M], q[M];
#pragma unroll
  for (int j = 0; j < M; ++j) d[j] = __builtin_fmax(d[j], -(double)SP_RANGE);  // also NaN -> -40
#pragma unroll
  for (int j = 0; j < M; ++j) kf[j] = __builtin_rint(d[j] * -(double)SP_STEPS);
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(kf[j], 1.0 / SP_STEPS, d[j]);          // exact
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const SoftplusNode nd = tab[(int)kf[j]];
    g0[j] = nd.g;
    s[j] = nd.s;
  }
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = 1.0 - s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = s[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = w[j] - s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) uw[j] = u[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], -12.0 / 120.0, 1.0 / 120.0);  // (1 - 12u)/120
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = uw[j] * q[j];                              // g5/5!
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], -6.0 / 24.0, 1.0 / 24.0);      // (1 - 6u)/24
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = u[j] * q[j];                               // g4/4!
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], r[j], q[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], r[j], uw[j] * (1.0 / 6.0));
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], r[j], u[j] * 0.5);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], r[j], s[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) g[j] = fma_(p[j], r[j], g0[j]);
}

// Two-phase form of the table-driven logPlus so that independent work (the next row's emission)
// can be placed between issuing the LDS lookups and consuming them:
//   SoftplusLookup<M> L;  log_plus_issue(x, y, L, tab);   ...independent code...   log_plus_finish(L, out);
template <int M>
struct SoftplusLookup {
  double hi[M], r[M], g0[M], s[M];
};

template <int M>
DYN_HD void log_plus_issue(const double (&x)[M], const double (&y)[M], SoftplusLookup<M>& L,
                           const SoftplusNode* __restrict__ tab) {
  double d[M], kf[M];
#pragma unroll
  for (int j = 0; j < M; ++j) L.hi[j] = __builtin_fmax(x[j], y[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) d[j] = __builtin_fmin(x[j], y[j]) - L.hi[j];
#pragma unroll
  for (int j = 0; j < M; ++j) d[j] = __builtin_fmax(d[j], -(double)SP_RANGE);  // also NaN -> -40
#pragma unroll
  for (int j = 0; j < M; ++j) kf[j] = __builtin_rint(d[j] * -(double)SP_STEPS);
#pragma unroll
  for (int j = 0; j < M; ++j) L.r[j] = fma_(kf[j], 1.0 / SP_STEPS, d[j]);  // exact
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const SoftplusNode nd = tab[(int)kf[j]];
    L.g0[j] = nd.g;
    L.s[j] = nd.s;
  }
}

template <int M>
DYN_HD void log_plus_finish(const SoftplusLookup<M>& L, double (&out)[M]) {
  double u[M], w[M], uw[M], p[M], q[M];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = 1.0 - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = L.s[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = w[j] - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) uw[j] = u[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], -12.0 / 120.0, 1.0 / 120.0);  // (1 - 12u)/120
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = uw[j] * q[j];                              // g5/5!
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], -6.0 / 24.0, 1.0 / 24.0);      // (1 - 6u)/24
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = u[j] * q[j];                               // g4/4!
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], q[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], uw[j] * (1.0 / 6.0));
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], u[j] * 0.5);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], L.s[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = L.hi[j] + fma_(p[j], L.r[j], L.g0[j]);
}

// log_plus_finish plus the logistic value sigma(d) = e^d / (1 + e^d) at the SAME argument d = lo - hi:
// sigma is the first derivative of the softplus, so it is the derivative of the same Taylor polynomial
// about the same node (4 more FMAs; truncation <= |g6|/120 * (1/256)^5 = 1.9e-15 * u, relative to a
// value in (0, 1/2]). exp(lo - logPlus(x, y)) = sigma and exp(hi - logPlus(x, y)) = 1 - sigma: the
// shares of the two operands in the sum, which is what the training pass needs (nt_kernels.hip,
// forward_train_sweep) -- without an exponential. d <= -40 (and -inf, NaN) gives sigma = 0 exactly.
template <int M>
DYN_HD void log_plus_finish_sigma(const SoftplusLookup<M>& L, double (&out)[M], double (&sig)[M]) {
  double u[M], w[M], uw[M], p[M], q[M], dp[M];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = 1.0 - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = L.s[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = w[j] - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) uw[j] = u[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], -12.0 / 120.0, 1.0 / 120.0);  // (1 - 12u)/120
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = uw[j] * q[j];                              // g5/5!
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], -6.0 / 24.0, 1.0 / 24.0);      // (1 - 6u)/24
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = u[j] * q[j];                               // g4/4!
#pragma unroll
  for (int j = 0; j < M; ++j) dp[j] = fma_(p[j] * 5.0, L.r[j], q[j] * 4.0);     // derivative: 5 c5 r + 4 c4
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], q[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) dp[j] = fma_(dp[j], L.r[j], uw[j] * 0.5);         // + 3 c3
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], uw[j] * (1.0 / 6.0));
#pragma unroll
  for (int j = 0; j < M; ++j) dp[j] = fma_(dp[j], L.r[j], u[j]);                // + 2 c2
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], u[j] * 0.5);
#pragma unroll
  for (int j = 0; j < M; ++j) sig[j] = fma_(dp[j], L.r[j], L.s[j]);             // + c1
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], L.s[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = L.hi[j] + fma_(p[j], L.r[j], L.g0[j]);
}

// exp(d) for M independent arguments, d clamped to [-1000, 700] (posterior exponents are <= ~0).
// Same reduction and degree-9 kernel as exp_nonpos above, written stage-by-stage across cells.
template <int M>
DYN_HD void exp_vec(double (&d)[M], double (&out)[M]) {
  const double LOG2E = 0x1.71547652b82fep+0;
  const double LN2_HI = 0x1.62e42fee00000p-1;
  const double LN2_LO = 0x1.a39ef35793c76p-33;
  double kf[M], r[M], q[M];
#pragma unroll
  for (int j = 0; j < M; ++j) d[j] = __builtin_fmin(__builtin_fmax(d[j], -1000.0), 700.0);
#pragma unroll
  for (int j = 0; j < M; ++j) kf[j] = __builtin_rint(d[j] * LOG2E);
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(-kf[j], LN2_HI, d[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(-kf[j], LN2_LO, r[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(0x1.af390ba7e6f47p-26, r[j], 0x1.2891d4ffbb0f9p-22);
#define DYN_STEP(c) _Pragma("unroll") for (int j = 0; j < M; ++j) q[j] = fma_(q[j], r[j], c);
  DYN_STEP(0x1.71de0d85293b8p-19)
  DYN_STEP(0x1.a019b8ca26fcfp-16)
  DYN_STEP(0x1.a01a01a7cebcdp-13)
  DYN_STEP(0x1.6c16c1789d1d7p-10)
  DYN_STEP(0x1.11111111109a6p-7)
  DYN_STEP(0x1.5555555553d37p-5)
  DYN_STEP(0x1.5555555555556p-3)
  DYN_STEP(0x1.0000000000001p-1)
#undef DYN_STEP
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(r[j] * r[j], q[j], r[j]) + 1.0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
    out[j] = __builtin_amdgcn_ldexp(q[j], (int)kf[j]);
#else
    out[j] = std::ldexp(q[j], (int)kf[j]);
#endif
  }
}

// Structure-of-arrays emission constants of the M cells of a lane.
template <int M>
struct EmisV {
  double mean[M], inv_stdev[M], log_norm[M];
  DYN_HD void set(int j, const Emis& e) {
    mean[j] = e.mean;
    inv_stdev[j] = e.inv_stdev;
    log_norm[j] = e.log_norm;
  }
};

template <int M>
DYN_HD void log_normal_pdf_vec(double x, const EmisV<M>& p, double (&out)[M]) {
  double diff[M], z[M], h[M];
#pragma unroll
  for (int j = 0; j < M; ++j) diff[j] = x - p.mean[j];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = diff[j] * p.inv_stdev[j];
#pragma unroll
  for (int j = 0; j < M; ++j) h[j] = -0.5 * z[j];
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = fma_(h[j], z[j], p.log_norm[j]);
}

}  // namespace dynmath
