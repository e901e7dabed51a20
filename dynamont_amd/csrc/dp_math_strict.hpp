// dp_math_strict.hpp -- the reference's per-cell arithmetic reproduced BIT FOR BIT ("strict mode").
//
// Why: where two neighbouring lattice columns carry the same k-mer (every RNA read starts with the polyA pad) the
// traceback's exact comparison (reference src/cpp/NT_aligner_api.cpp:448) is a tie in exact arithmetic and the
// reference's choice rests on the last bits of its sums. dp_math.hpp's table softplus is <= 1 ulp away from
// glibc's log1p(exp()) in 3.6 % of the calls, which flips 3 of 1 000 short tie-bearing reads (tests/tie_parity.py).
// The functions below follow the reference's expressions operation by operation, with the two libm calls replaced
// by restatements of the algorithms glibc 2.35 runs on x86-64 (what oracle/_ref links):
//
//   logPlus         src/cpp/aligner.cpp:276-285   x + log1p(exp(y - x)), -inf operands pass through
//   log_normal_pdf  src/cpp/aligner.cpp:287-292   (-0.5*z*z - log(stdev)) - 0.5*log(2 pi), z = (x - mean) / stdev
//   exp             glibc sysdeps/ieee754/dbl-64/e_exp.c (ARM optimized-routines; 2^(k/128) table, degree-5
//                   polynomial) AS COMPILED for the FMA ifunc variant (__exp_fma): which a*b+c were contracted was
//                   read off the disassembly of libm.so.6 -- kd = fma(x, InvLn2N, Shift), both reduction steps, the
//                   polynomial and scale + scale*tmp are FMAs; the product inside specialcase() is not.
//   log1p           glibc sysdeps/ieee754/dbl-64/s_log1p.c (fdlibm), no ifunc variant, no contraction.
// log(stdev) is evaluated on the host by the host's libm at model load (dp_math.hpp, Emis::neg_log_stdev).
//
// tests/test_dp_math.py compiles this header with g++ and compares both functions with libm on > 1e7 arguments
// (0 differing bits); tests/tie_parity.py mode 6 replays the oracle's control flow with them (0 differing reads).
// Every operation is a single IEEE fp64 operation, so the device build (hipcc -ffp-contract=off, IEEE division)
// produces the same bits.
#pragma once

#include <cstdint>

#include "dp_math.hpp"

namespace dynmath {

constexpr int STRICT_EXP_WORDS = 256;  // 128 x (tail, scale bits)

// Host copy of the table (uploaded behind the softplus nodes; staged into LDS by the DP workgroups).
inline const uint64_t* strict_exp_table() {
  static const uint64_t T[STRICT_EXP_WORDS] = {
#include "strict_exp_table.inc"
  };
  return T;
}

DYN_HD uint64_t bits_of(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__double_as_longlong(x);
#else
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
#endif
}

DYN_HD double of_bits(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)u);
#else
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
#endif
}

DYN_HD int32_t hi_word(double x) { return (int32_t)(bits_of(x) >> 32); }
DYN_HD double with_hi_word(double x, int32_t h) {
  return of_bits((bits_of(x) & 0xffffffffull) | ((uint64_t)(uint32_t)h << 32));
}

// ---- exp(x), x in [-inf, 0] -------------------------------------------------------------------
// |x| >= 512 (results below 2^-738) take glibc's specialcase() path; kept out of line of the common one.
DYN_HD double exp_strict_special(double x, uint64_t ki, uint64_t sbits, double tmp) {
  const uint32_t abstop = (uint32_t)(bits_of(x) >> 52) & 0x7ffu;
  if (abstop >= 0x409u) return 0.0;  // x <= -1024 (and -inf): underflow to +0
  // k < 0: the exponent of scale is raised by 1022, the product rescaled at the end
  (void)ki;
  sbits += 1022ull << 52;
  const double scale = of_bits(sbits);
  const double st = scale * tmp;  // NOT contracted in glibc's build
  double y = scale + st;
  if (y < 1.0) {
    double lo = (scale - y) + st;
    const double hi = 1.0 + y;
    lo = ((1.0 - hi) + y) + lo;
    y = (lo + hi) - 1.0;
    if (y == 0.0) y = 0.0;
  }
  return 0x1p-1022 * y;
}

template <int M>
DYN_HD void exp_strict_vec(const double (&x)[M], double (&out)[M], const uint64_t* __restrict__ tab) {
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  double kd[M], r[M], r2[M], p[M], q[M], tail[M];
  uint64_t ki[M], sb[M];
  bool any_special = false;
#pragma unroll
  for (int j = 0; j < M; ++j) kd[j] = fma_(x[j], InvLn2N, Shift);
#pragma unroll
  for (int j = 0; j < M; ++j) ki[j] = bits_of(kd[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) kd[j] = kd[j] - Shift;
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(kd[j], NegLn2hiN, x[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(kd[j], NegLn2loN, r[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const unsigned idx = 2u * ((unsigned)ki[j] & 127u);
    tail[j] = of_bits(tab[idx]);
    sb[j] = tab[idx + 1] + (ki[j] << 45);
  }
#pragma unroll
  for (int j = 0; j < M; ++j) r2[j] = r[j] * r[j];
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(r[j], C3, C2);
#pragma unroll
  for (int j = 0; j < M; ++j) tail[j] = tail[j] + r[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(r[j], C5, C4);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], r2[j], tail[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) r2[j] = r2[j] * r2[j];
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(r2[j], q[j], p[j]);  // tmp
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const double scale = of_bits(sb[j]);
    out[j] = fma_(scale, p[j], scale);
  }
  // |x| < 2^-54 ("tiny": 1.0 + x) needs no case of its own: the common path gives fma(1, x, 1) there.
#pragma unroll
  for (int j = 0; j < M; ++j) any_special |= !(x[j] > -512.0);
  if (__builtin_expect(any_special, 0)) {
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (!(x[j] > -512.0)) out[j] = exp_strict_special(x[j], ki[j], sb[j], p[j]);
  }
}

// exp(x) for the training sweeps (emission probabilities, posterior masses), x <= ~1, on glibc's 2^(i/128) values kept
// as plain doubles (exp128_build_table: 128 of them behind the softplus nodes in LDS): glibc's reduction, a degree-3
// polynomial and no tail correction: |r| <= ln2/256, truncation r^4/24 <= 2.3e-12 relative (tests/test_dp_math.py).
// That is what its users need: a posterior is a ratio of products of T such factors along paths of the same length,
// the common part of the error cancels and the rest adds up like sqrt(T) x 1e-12 -- fifty times below the rounding
// noise of the reference's log-space sums (tests/extended_precision_train.py), and Z moves by 1e-12 relative.
// 10 fp64/integer operations and one ds_read_b64. The power of two
// is applied with ldexp, so the result runs down through the denormals to an exact 0 below -745.2, like exp itself;
// -inf and NaN give 0 (the argument is clamped to -750 first). The exact zero matters to the linear-domain sweeps: a
// cell that cannot be represented must lose its mass visibly (the read is then redone in the log domain), not keep
// 1e-304 of it.
constexpr int EXP128_SIZE = 128;
constexpr int EXP128_NODES = EXP128_SIZE / 2;  // in units of SoftplusNode (two doubles)
inline void exp128_build_table(double* t) {
  const uint64_t* T = strict_exp_table();  // glibc keeps asuint64(2^(i/128)) - (i << 45)
  for (int i = 0; i < EXP128_SIZE; ++i) t[i] = of_bits(T[2 * i + 1] + ((uint64_t)i << 45));
}

DYN_HD double ldexp_(double v, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ldexp(v, e);
#else
  return __builtin_ldexp(v, e);
#endif
}

template <int M, int J0 = 0, int J1 = M>
DYN_HD void exp_table128_vec(double (&x)[M], double (&out)[M], const double* __restrict__ tab) {
  constexpr int K = J1 - J0;
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  double kd[K], r[K], r2[K], p[K], tv[K];
  int ke[K];
  const double shift = vreg_const(Shift), c6 = vreg_const(1.0 / 6.0);
  const double inv = sreg_const(InvLn2N), floor750 = vreg_const(-750.0);
#pragma unroll
  for (int j = 0; j < K; ++j) x[J0 + j] = max_hw(x[J0 + j], floor750);
#pragma unroll
  for (int j = 0; j < K; ++j) kd[j] = fma_(x[J0 + j], inv, shift);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int ki = (int)(unsigned)bits_of(kd[j]);  // the low word of kd: round(x 128 / ln2), two's complement
    tv[j] = tab[(unsigned)ki & 127u];  // 2^(i/128), i = ki mod 128
    ke[j] = ki >> 7;
  }
#pragma unroll
  for (int j = 0; j < K; ++j) kd[j] = kd[j] - shift;
#pragma unroll
  for (int j = 0; j < K; ++j) r[j] = fma_(kd[j], NegLn2hiN, x[J0 + j]);
#pragma unroll
  for (int j = 0; j < K; ++j) r[j] = fma_(kd[j], NegLn2loN, r[j]);
#pragma unroll
  for (int j = 0; j < K; ++j) r2[j] = r[j] * r[j];
#pragma unroll
  for (int j = 0; j < K; ++j) p[j] = fma_(r[j], c6, 0.5);
#pragma unroll
  for (int j = 0; j < K; ++j) p[j] = fma_(p[j], r2[j], r[j]);   // exp(r) - 1 = r + r^2 (1/2 + r/6) + O(r^4 / 24)
#pragma unroll
  for (int j = 0; j < K; ++j) out[J0 + j] = ldexp_(fma_(tv[j], p[j], tv[j]), ke[j]);
}

// ---- log1p(x), x in [0, 1] ----------------------------------------------------------------------
// The branches of s_log1p.c that such an argument can take, evaluated side by side and selected:
//   x < 2^-54                    -> x
//   x < 2^-29                    -> x - x*x*0.5
//   hx < 0x3FDA827A (x < ~0.4142) -> k = 0, f = x
//   otherwise                     -> u = 1 + x, c = (x - (u - 1)) / u  [k = 0 before normalisation; x == 1: k = 1 and
//                                    c = 1 - (u - x), the same value 0], u normalised into [sqrt2/2, sqrt2), f = u - 1
// and |f| < 2^-20 ("hu == 0") swaps the polynomial for R = hfsq*(1 - 2/3 f). That last case (|x - y| of the logPlus
// below ~4e-6) is rare and sits in an unlikely block.
DYN_HD double log1p_strict_small_f(double f, double hfsq, int k, double c) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  if (f == 0.0) {
    if (k == 0) return 0.0;
    c += (double)k * ln2_lo;
    return (double)k * ln2_hi + c;
  }
  const double R = hfsq * (1.0 - 0.66666666666666666 * f);
  if (k == 0) return f - R;
  return (double)k * ln2_hi - ((R - ((double)k * ln2_lo + c)) - f);
}

template <int M>
DYN_HD void log1p_strict_vec(const double (&x)[M], double (&out)[M]) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  double f[M], c[M], hfsq[M], s[M], z[M], R[M], w[M];
  int kk[M];
  bool big[M], smallf[M];
  bool any_smallf = false;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int32_t hx = hi_word(x[j]);
    big[j] = hx >= 0x3FDA827A;
    const double u = 1.0 + x[j];
    int32_t hu = hi_word(u);
    int k = (hu >> 20) - 1023;                         // 0, or 1 for x == 1
    double cc = (k > 0) ? 1.0 - (u - x[j]) : x[j] - (u - 1.0);
    cc = cc / u;
    hu &= 0x000fffff;
    double un;
    if (hu < 0x6a09e) {
      un = with_hi_word(u, hu | 0x3ff00000);
    } else {
      k += 1;
      un = with_hi_word(u, hu | 0x3fe00000);
      hu = (0x00100000 - hu) >> 2;
    }
    f[j] = big[j] ? un - 1.0 : x[j];
    c[j] = cc;
    kk[j] = big[j] ? k : 0;
    smallf[j] = big[j] && hu == 0;
  }
#pragma unroll
  for (int j = 0; j < M; ++j) hfsq[j] = (0.5 * f[j]) * f[j];
#pragma unroll
  for (int j = 0; j < M; ++j) s[j] = f[j] / (2.0 + f[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = s[j] * s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const double R1 = z[j] * Lp1, z2 = z[j] * z[j];
    const double R2 = Lp2 + z[j] * Lp3, z4 = z2 * z2;
    const double R3 = Lp4 + z[j] * Lp5, z6 = z4 * z2;
    const double R4 = Lp6 + z[j] * Lp7;
    R[j] = ((R1 + z2 * R2) + z4 * R3) + z6 * R4;
  }
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = s[j] * (hfsq[j] + R[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const double r_small = f[j] - (hfsq[j] - w[j]);                                            // k == 0
    const double kf = (double)kk[j];
    const double r_big = kf * ln2_hi - ((hfsq[j] - (w[j] + (kf * ln2_lo + c[j]))) - f[j]);     // k != 0
    double r = (kk[j] != 0) ? r_big : r_small;
    const int32_t hx = hi_word(x[j]);
    if (hx < 0x3e200000) r = x[j] - (x[j] * x[j]) * 0.5;  // x < 2^-29
    if (hx < 0x3c900000) r = x[j];                         // x < 2^-54 (incl. 0 and subnormals)
    out[j] = r;
    any_smallf |= smallf[j];
  }
  if (__builtin_expect(any_smallf, 0)) {
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (smallf[j]) out[j] = log1p_strict_small_f(f[j], hfsq[j], kk[j], c[j]);
  }
}

// aligner.cpp:276-285 for operands that are finite or -inf (the DP never produces +inf; NaN only from NaN samples,
// which the sweeps flag separately).
template <int M>
DYN_HD void log_plus_strict_vec(const double (&x)[M], const double (&y)[M], double (&out)[M],
                                const uint64_t* __restrict__ tab) {
  double hi[M], d[M], e[M], g[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const bool swap = x[j] < y[j];
    hi[j] = swap ? y[j] : x[j];
    const double lo = swap ? x[j] : y[j];
    // lo == -inf: the reference returns the other operand; exp(-inf) = 0 and log1p(0) = 0 give hi + 0 = hi as well,
    // except for (-inf) - (-inf) = NaN, hence the select
    d[j] = (lo == NEG_INF) ? NEG_INF : lo - hi[j];
  }
  exp_strict_vec<M>(d, e, tab);
  log1p_strict_vec<M>(e, g);
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = hi[j] + g[j];
}

DYN_HD double log_plus_strict(double x, double y, const uint64_t* __restrict__ tab) {
  double a[1] = {x}, b[1] = {y}, o[1];
  log_plus_strict_vec<1>(a, b, o, tab);
  return o[0];
}

// The same from hi = max(x, y) and diff = x - y (as rounded): lo - hi = -|diff|, a correctly rounded difference having
// the same magnitude either way round. diff = NaN is (-inf) - (-inf): the reference returns -inf, as does hi + anything.
DYN_HD double log_plus_strict_from(double hi, double diff, const uint64_t* __restrict__ tab) {
  double d[1] = {max_hw(-__builtin_fabs(diff), NEG_INF)}, e[1], g[1];  // max_hw: a NaN operand loses
  exp_strict_vec<1>(d, e, tab);
  log1p_strict_vec<1>(e, g);
  return hi + g[0];
}

// aligner.cpp:287-292, operation by operation. In strict mode EmisV::inv_stdev carries STDEV itself.
template <int M>
DYN_HD void log_normal_pdf_strict_vec(double x, const EmisV<M>& p, double (&out)[M]) {
  double z[M];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = x - p.mean[j];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = z[j] / p.inv_stdev[j];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = (-0.5 * z[j]) * z[j];
#pragma unroll
  for (int j = 0; j < M; ++j) z[j] = z[j] + p.neg_log_stdev[j];  // t - log(stdev) == t + (-log(stdev)) exactly
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = z[j] - HALF_LOG_2PI;
}

DYN_HD double log_normal_pdf_strict(double x, const Emis& p) {
  const double z = (x - p.mean) / p.stdev;
  return (((-0.5 * z) * z) + p.neg_log_stdev) - HALF_LOG_2PI;
}

// =====================================================================================================================
// CERTIFIED arithmetic (round 4): the reference's bits at close to the default arithmetic's price.
//
// Restating glibc operation by operation (above) costs ~130 instructions per logPlus and three IEEE divisions per
// cell. Almost none of that is needed to know the reference's RESULT:
//
//  * logPlus = hi + g, g = log1p(exp(d)) in (0, ln 2]. hi is a log-probability of magnitude 1e2 .. 1e5, so the sum is
//    rounded to ulp(hi) ~ 1e-14 .. 1e-11 while glibc's g and the table softplus of dp_math.hpp agree to ~1e-16. Let
//    g~ be the table value and delta = g~ 2^-50 + 2^-57. If RN(hi + (g~ - delta)) == RN(hi + (g~ + delta)) then, rounding
//    being monotonic, EVERY g in that interval gives the same sum -- glibc's included: the sum is certified to be the
//    reference's bit for bit without evaluating glibc's algorithm. Five fp64 operations per cell. Where the two sums
//    differ (a rounding boundary of hi's grid falls inside the interval: probability ~ 8 g / |hi| per cell) the cell is
//    AMBIGUOUS and its register (64 cells) is recomputed with the restated glibc algorithm, M = 1.
//    Why glibc's g lies in the interval: glibc's exp is within 0.511 ulp (e_exp.c, "worst-case error 0.511 ULP"
//    for the FMA build: 0.509), fdlibm's log1p within 1 ulp (s_log1p.c: "error < 1 ulp"), and d log1p(e)/de x e
//    <= log1p(e): |g_glibc - g| <= 1.511 x 2^-52 g. The table value: node rounded to nearest from an 80-bit
//    evaluation (<= 0.5003 x 2^-52 g, nodes within e^(1/256) of g), final FMA rounding (<= 0.5 x 2^-52 g), polynomial
//    roundings and truncation (<= 0.01 x 2^-52 g): |g~ - g| <= 1.01 x 2^-52 g. Together 2.52 x 2^-52 g; the interval is
//    4 x 2^-52 g~ wide on each side, minus the rounding of its own end points (0.5 x 2^-52 g each). Below d = -40 the
//    table returns exactly 0 where glibc returns exp(d) <= 4.25e-18 < 2^-57: the absolute term.
//    tests/test_dp_math_strict.py measures max |g~ - g_glibc| / g on 1e8 arguments (observed: 1.0 x 2^-52) and checks on
//    as many (hi, lo) pairs that a certified sum never differs from the reference expression.
//  * the emission's division by stdev: 1/stdev is a per-k-mer constant, correctly rounded on the host (y = RN(1/b)).
//    q0 = a y; r0 = a - q0 b; q1 = q0 + r0 y; r1 = a - q1 b; q2 = q1 + r1 y (one multiplication, four FMAs). q1 is
//    within 1/2 + 2^-52 ulp of a/b (faithful); Markstein's theorem (Markstein 1990; Muller et al., Handbook of
//    Floating-Point Arithmetic, Thm. 4.9) then makes q2 = RN(a/b) for every a, provided the significand of b is not
//    all ones (checked per model at load, dyn_aligner_set_strict refuses such a model) and nothing over- or underflows
//    (|a| = |x - mean| <= 1e300 is enforced by the sweeps; a is 0 or >= 2^-53 |x|).
// =====================================================================================================================
constexpr double CERT_LO = 1.0 - 0x1p-50, CERT_HI = 1.0 + 0x1p-50, CERT_ABS = 0x1p-57;

// a / b, correctly rounded, given y = RN(1/b) (see above)
DYN_HD double div_by_const(double a, double b, double y) {
  double q = a * y;
  double r = fma_(-q, b, a);
  q = fma_(r, y, q);
  r = fma_(-q, b, a);
  return fma_(r, y, q);
}

// The same quotient in FOUR operations (round 6). With y = RN(1/b) and y_lo = RN(RN(1 - y b) y) -- `recip_lo`: the residual
// 1 - y b is exact in one FMA, |1 - y b| <= 2^-53 -- the pair (y, y_lo) carries 1/b to ~2^-104 relative, so
//   q~ = fma(a, y, RN(a y_lo))
// is a/b rounded ONCE up to ~2^-104 |a/b|: faithful for every a (it is RN(a/b) itself except for the handful of dividends per
// divisor whose quotient lies within 2^-104 of a rounding midpoint -- which no run-time test tells apart for less than it
// saves). One step of Markstein's iteration on a FAITHFUL quotient, r = a - q~ b (exact), q = RN(q~ + r y),
// is RN(a/b) for every a (Handbook of Floating-Point Arithmetic, Thm. 4.9; same excluded divisor, same no-overflow condition
// as above): mul, fma, fma, fma instead of mul, fma, fma, fma, fma.
DYN_HD double recip_lo(double b, double y) { return fma_(-y, b, 1.0) * y; }
DYN_HD double div_by_const4(double a, double b, double y, double y_lo) {
  const double t = a * y_lo;
  const double q = fma_(a, y, t);
  const double r = fma_(-q, b, a);
  return fma_(r, y, q);
}

// is the significand of b all ones? (the one case Markstein's theorem excludes)
inline bool div_by_const_excluded(double b) { return (bits_of(b) & 0x000fffffffffffffull) == 0x000fffffffffffffull; }

// aligner.cpp:287-292 bit for bit without a division instruction: needs BOTH stdev and 1/stdev per cell
template <int M>
DYN_HD void log_normal_pdf_cert_vec(double x, const EmisV<M>& p, const double (&stdev)[M], double (&out)[M]) {
  double a[M], q[M], r[M];
#pragma unroll
  for (int j = 0; j < M; ++j) a[j] = x - p.mean[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = a[j] * p.inv_stdev[j];
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(-q[j], stdev[j], a[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(r[j], p.inv_stdev[j], q[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(-q[j], stdev[j], a[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(r[j], p.inv_stdev[j], q[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = (-0.5 * q[j]) * q[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = q[j] + p.neg_log_stdev[j];
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = q[j] - HALF_LOG_2PI;
}

// the same with the four-operation quotient: y_lo[j] = recip_lo(stdev[j], p.inv_stdev[j])
template <int M>
DYN_HD void log_normal_pdf_cert4_vec(double x, const EmisV<M>& p, const double (&stdev)[M], const double (&y_lo)[M], double (&out)[M]) {
  double a[M], q[M], r[M];
#pragma unroll
  for (int j = 0; j < M; ++j) a[j] = x - p.mean[j];
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = a[j] * y_lo[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(a[j], p.inv_stdev[j], r[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) r[j] = fma_(-q[j], stdev[j], a[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(r[j], p.inv_stdev[j], q[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = (-0.5 * q[j]) * q[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = q[j] + p.neg_log_stdev[j];
#pragma unroll
  for (int j = 0; j < M; ++j) out[j] = q[j] - HALF_LOG_2PI;
}

DYN_HD double log_normal_pdf_cert(double x, const Emis& p) {
  const double z = div_by_const(x - p.mean, p.stdev, p.inv_stdev);
  return (((-0.5 * z) * z) + p.neg_log_stdev) - HALF_LOG_2PI;
}

// Second half of a certified logPlus (first half: log_plus_issue of dp_math.hpp, unchanged). lo[j] == hi_[j] certifies
// out[j] = lo[j] as the reference's sum; the caller recomputes the others with log_plus_strict.
template <int M>
DYN_HD void log_plus_finish_cert(const SoftplusLookup<M>& L, double (&lo)[M], double (&hi_)[M]) {
  double u[M], w[M], p[M], q[M];
  const double c120 = vreg_const(1.0 / 120.0), c24 = vreg_const(1.0 / 24.0);
  const double m10 = sreg_const(-12.0 / 120.0), m4 = sreg_const(-0.25);
  const double cabs = vreg_const(CERT_ABS), clo = sreg_const(CERT_LO), chi = sreg_const(CERT_HI);
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = 1.0 - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = L.s[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) w[j] = w[j] - L.s[j];
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(u[j], m10, c120);
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = q[j] * w[j];
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(u[j], m4, c24);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(q[j], L.r[j], p[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], w[j] * (1.0 / 6.0));
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], 0.5);
#pragma unroll
  for (int j = 0; j < M; ++j) u[j] = u[j] * L.r[j];
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], u[j], L.s[j]);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], L.r[j], L.g0[j]);  // the table softplus g~ (log_plus_finish adds hi here)
#pragma unroll
  for (int j = 0; j < M; ++j) q[j] = fma_(p[j], clo, -cabs);
#pragma unroll
  for (int j = 0; j < M; ++j) p[j] = fma_(p[j], chi, cabs);
#pragma unroll
  for (int j = 0; j < M; ++j) lo[j] = L.hi[j] + q[j];
#pragma unroll
  for (int j = 0; j < M; ++j) hi_[j] = L.hi[j] + p[j];
}

// host form (tests, CPU replay): the certified logPlus with its fallback; *ambiguous counts the fallbacks
inline double log_plus_cert(double x, double y, const SoftplusNode* sp_tab, const uint64_t* exp_tab, long* ambiguous) {
  double a[1] = {x}, b[1] = {y}, lo[1], hi_[1];
  SoftplusLookup<1> L;
  log_plus_issue<1>(a, b, L, sp_tab);
  log_plus_finish_cert<1>(L, lo, hi_);
  if (lo[0] == hi_[0]) return lo[0];
  if (ambiguous) ++*ambiguous;
  return log_plus_strict_from(L.hi[0], L.diff[0], exp_tab);  // what the kernels' fallback calls
}

}  // namespace dynmath
