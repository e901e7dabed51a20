"""CPU: accuracy of the per-cell fp64 arithmetic (dynamont_amd/csrc/dp_math.hpp), compiled for
the host with g++. softplus vs 40-digit mpmath; log_normal_pdf bit-identical to the oracle."""
import ctypes as C
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle

SRC = r'''
#include "%s/dynamont_amd/csrc/dp_math_strict.hpp"
#include <vector>
extern "C" {
void eval_softplus(const double* d, double* out, long n) { for (long i = 0; i < n; ++i) out[i] = dynmath::softplus_nonpos(d[i]); }
void eval_logplus(const double* x, const double* y, double* out, long n) { for (long i = 0; i < n; ++i) out[i] = dynmath::log_plus(x[i], y[i]); }
static std::vector<dynmath::SoftplusNode> TAB;
void init_table() { TAB.resize(dynmath::SP_NODES); dynmath::softplus_build_table(TAB.data()); }
// the product path: table-driven logPlus in its two-phase, 7-cells-at-a-time form
void eval_logplus_table(const double* x, const double* y, double* out, long n) {
  for (long i = 0; i + 7 <= n; i += 7) {
    double a[7], b[7], o[7];
    for (int j = 0; j < 7; ++j) { a[j] = x[i + j]; b[j] = y[i + j]; }
    dynmath::SoftplusLookup<7> L;
    dynmath::log_plus_issue<7>(a, b, L, TAB.data());
    dynmath::log_plus_finish<7>(L, o);
    for (int j = 0; j < 7; ++j) out[i + j] = o[j];
  }
}
// logPlus + the logistic share sigma(lo - hi) from the same table lookup (training pass)
void eval_softplus_table(const double* d, double* out, long n) {
  for (long i = 0; i + 7 <= n; i += 7) {
    double a[7], g[7];
    for (int j = 0; j < 7; ++j) a[j] = d[i + j];
    dynmath::softplus_table_vec<7>(a, g, TAB.data());
    for (int j = 0; j < 7; ++j) out[i + j] = g[j];
  }
}
void eval_softplus_table3(const double* d, double* out, long n) {
  for (long i = 0; i + 7 <= n; i += 7) {
    double a[7], g[7];
    for (int j = 0; j < 7; ++j) a[j] = d[i + j];
    dynmath::softplus_table3_vec<7>(a, g, TAB.data());
    for (int j = 0; j < 7; ++j) out[i + j] = g[j];
  }
}
void eval_exp_vec(const double* d, double* out, long n) {   // exp_table128_vec: the training sweeps' exponential
  for (long i = 0; i + 7 <= n; i += 7) {
    double a[7], g[7];
    for (int j = 0; j < 7; ++j) a[j] = d[i + j];
    static double ET[dynmath::EXP128_SIZE]; static bool init = false;
    if (!init) { dynmath::exp128_build_table(ET); init = true; }
    dynmath::exp_table128_vec<7>(a, g, ET);
    for (int j = 0; j < 7; ++j) out[i + j] = g[j];
  }
}
void eval_pdf(const double* x, const double* mean, const double* sd, double* out, long n) {
  for (long i = 0; i < n; ++i) out[i] = dynmath::log_normal_pdf_exact(x[i], mean[i], sd[i], std::log(sd[i])); }
void eval_pdf_fast(const double* x, const double* mean, const double* sd, double* out, long n) {
  for (long i = 0; i + 7 <= n; i += 7) {
    dynmath::EmisV<7> p; double o[7];
    for (int j = 0; j < 7; ++j) p.set(j, dynmath::make_emis(mean[i + j], sd[i + j], std::log(sd[i + j])));
    // the kernels evaluate one sample against the 7 cells of a lane; here one sample per cell
    for (int j = 0; j < 7; ++j) { dynmath::log_normal_pdf_vec<7>(x[i + j], p, o); out[i + j] = o[j]; }
    for (int j = 0; j < 7; ++j) if (out[i + j] != dynmath::log_normal_pdf(x[i + j], dynmath::make_emis(mean[i + j], sd[i + j], std::log(sd[i + j])))) out[i + j] = 1e300;
  } }
}
''' % ROOT

dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def mathlib(tmp_path_factory):
    d = tmp_path_factory.mktemp("dpmath")
    src = d / "t.cpp"
    src.write_text(SRC)
    so = d / "libt.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    lib = C.CDLL(str(so))
    lib.init_table()
    return lib


def test_softplus_accuracy(mathlib):
    rng = np.random.default_rng(0)
    d = np.concatenate([-np.abs(rng.standard_normal(1500)) * 3, -rng.uniform(0, 45, 1500),
                        -10.0 ** rng.uniform(-12, 2.9, 1500), [0.0, -0.8813735870195429, -1e-300, -745.0, -1000.0]])
    out = np.empty_like(d)
    mathlib.eval_softplus(d.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(len(d)))
    mp.mp.dps = 40
    worst = max(abs(mp.mpf(float(y)) - mp.log1p(mp.exp(mp.mpf(float(x))))) for x, y in zip(d, out))
    assert worst < 2.0e-16, worst


def test_logplus_special_values(mathlib):
    inf = np.inf
    x = np.array([-inf, 3.0, -inf, -5.0, 1e4, -2000.0])
    y = np.array([-inf, -inf, -7.5, -5.0, 1e4 - 800.0, 2000.0])
    out = np.empty_like(x)
    mathlib.eval_logplus(x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(len(x)))
    assert out[0] == -inf and out[1] == 3.0 and out[2] == -7.5      # aligner.cpp:278-281
    assert abs(out[3] - (-5.0 + np.log(2.0))) < 1e-15
    assert out[4] == 1e4 and out[5] == 2000.0


def test_logplus_matches_oracle_closely(mathlib, oracle_built):
    L = C.CDLL(pyoracle.ORACLE_SO)
    L.nto_log_plus.restype = C.c_double
    L.nto_log_plus.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(1)
    x = rng.uniform(-5000, 100, 20000)
    y = x + rng.uniform(-60, 60, 20000)
    out = np.empty_like(x)
    mathlib.eval_logplus(x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(len(x)))
    ref = np.array([L.nto_log_plus(a, b) for a, b in zip(x, y)])
    assert np.abs(out - ref).max() <= 2 * np.spacing(np.abs(ref)).max()
    assert np.mean(out == ref) > 0.95


def test_log_normal_pdf_bit_identical(mathlib, oracle_built):
    L = C.CDLL(pyoracle.ORACLE_SO)
    L.nto_log_normal_pdf.restype = C.c_double
    L.nto_log_normal_pdf.argtypes = [C.c_double] * 3
    rng = np.random.default_rng(2)
    n = 50000
    x = rng.standard_normal(n) * 2
    mean = rng.standard_normal(n)
    sd = rng.uniform(0.05, 0.5, n)
    sd[: n // 2] = 0.15
    sd[n // 2: n // 2 + 10000] = 0.25
    out = np.empty(n)
    mathlib.eval_pdf(x.ctypes.data_as(dp), mean.ctypes.data_as(dp), sd.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(n))
    ref = np.array([L.nto_log_normal_pdf(a, b, c) for a, b, c in zip(x, mean, sd)])
    assert np.array_equal(out, ref)


def test_kernel_emission_within_a_few_ulp_of_the_reference_expression(mathlib):
    """The kernels run the 5-operation emission (uncorrected quotient, the reference's rounding sequence after
    it); it must stay within a few ulp of the terms it adds and be bit-identical most of the time."""
    rng = np.random.default_rng(3)
    n = 49 * 1000
    x = rng.standard_normal(n) * 3
    mean = rng.standard_normal(n)
    sd = rng.uniform(0.05, 0.5, n)
    sd[: n // 2] = 0.15
    x[:100] = mean[:100] + 40 * sd[:100]                      # far tails
    exact, fast = np.empty(n), np.empty(n)
    args = (x.ctypes.data_as(dp), mean.ctypes.data_as(dp), sd.ctypes.data_as(dp))
    mathlib.eval_pdf(*args, exact.ctypes.data_as(dp), C.c_long(n))
    mathlib.eval_pdf_fast(*args, fast.ctypes.data_as(dp), C.c_long(n))
    assert fast.max() < 1e299                                 # vector form == scalar form, bit for bit
    z2 = 0.5 * ((x - mean) / sd) ** 2
    scale = np.spacing(z2 + np.abs(np.log(sd)) + 0.92)
    err = np.abs(fast - exact) / scale
    assert err.max() <= 4.0 and np.mean(fast == exact) > 0.6, (err.max(), np.mean(fast == exact))


def _grid7(rng):
    d = np.concatenate([-np.abs(rng.standard_normal(1400)) * 3, -rng.uniform(0, 45, 1400),
                        -10.0 ** rng.uniform(-12, 2.9, 1393), [0.0, -1 / 256, -1 / 128, -39.999, -40.0, -40.01, -1e9]])
    return d[: len(d) // 7 * 7].copy()


def test_table_softplus_accuracy(mathlib):
    """The form the kernels run: (g, sigma) nodes every 1/128 + degree-5 Taylor about the nearest node."""
    d = _grid7(np.random.default_rng(3))
    out = np.empty_like(d)
    mathlib.eval_softplus_table(d.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(len(d)))
    mp.mp.dps = 40
    worst = max(abs(mp.mpf(float(y)) - mp.log1p(mp.exp(mp.mpf(float(x))))) for x, y in zip(d, out))
    assert worst < 1.5e-16, worst
    assert out[-1] == 0.0 and out[-2] == 0.0     # d <= -40 -> exactly 0


def test_table_softplus_degree3_for_the_training_backward_sweep(mathlib):
    """log_plus_finish3: the same nodes, degree-3 Taylor (9 operations instead of 14) -- 1.2e-12, for train() only."""
    d = _grid7(np.random.default_rng(3))
    out = np.empty_like(d)
    mathlib.eval_softplus_table3(d.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(len(d)))
    mp.mp.dps = 40
    worst = max(abs(mp.mpf(float(y)) - mp.log1p(mp.exp(mp.mpf(float(x))))) for x, y in zip(d, out))
    assert worst < 1.3e-12, worst
    assert out[-1] == 0.0 and out[-2] == 0.0     # d <= -40 -> exactly 0


def test_table_logplus_special_values_and_oracle(mathlib, oracle_built):
    inf = np.inf
    x = np.array([-inf, 3.0, -inf, -5.0, 1e4, -2000.0, 0.25])
    y = np.array([-inf, -inf, -7.5, -5.0, 1e4 - 800.0, 2000.0, 0.25])
    out = np.empty_like(x)
    mathlib.eval_logplus_table(x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(7))
    assert out[0] == -inf and out[1] == 3.0 and out[2] == -7.5 and out[4] == 1e4 and out[5] == 2000.0
    assert abs(out[3] - (-5.0 + np.log(2.0))) < 1e-15
    L = C.CDLL(pyoracle.ORACLE_SO)
    L.nto_log_plus.restype = C.c_double
    L.nto_log_plus.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(4)
    a = rng.uniform(-5000, 100, 7000)
    b = a + rng.uniform(-60, 60, 7000)
    o = np.empty_like(a)
    mathlib.eval_logplus_table(a.ctypes.data_as(dp), b.ctypes.data_as(dp), o.ctypes.data_as(dp), C.c_long(len(a)))
    ref = np.array([L.nto_log_plus(p, q) for p, q in zip(a, b)])
    assert np.abs(o - ref).max() <= 2 * np.spacing(np.abs(ref)).max()
    assert np.mean(o == ref) > 0.95


def test_exp_table128_vec_accuracy_and_exact_zero(mathlib):
    """2.5e-12 relative (degree-3 polynomial on |r| <= ln2/256) down to the smallest normal number, gradual underflow below it (like exp itself: half a unit of
    the denormal spacing), exactly 0 from -745.2 on and for -inf / NaN -- the posterior chain's stay probability of a cell
    nothing can reach (its stored exponent is (-inf) - (-inf)) must be 0, not 1e-304."""
    rng = np.random.default_rng(5)
    d = np.concatenate([-rng.uniform(0, 60, 3500), -rng.uniform(600, 708, 1400), rng.uniform(-1e-9, 1e-9, 700), rng.uniform(0, 5, 686),
                        -rng.uniform(708, 746, 700), [0.0, -745.0, -745.2, -999.0, -1e9, 1e-300, -1e-300, -0.5, -np.inf, np.nan, -746.0, -750.0, -751.0, -1e300]])
    d = d[: len(d) // 7 * 7].copy()
    out = np.empty_like(d)
    mathlib.eval_exp_vec(d.ctypes.data_as(dp), out.ctypes.data_as(dp), C.c_long(len(d)))
    mp.mp.dps = 40
    worst = 0.0
    tiny = mp.mpf(2) ** -1074
    for x, y in zip(d, out):
        if not np.isfinite(x) or x < -745.14:
            assert y == 0.0, (x, y)
            continue
        t = mp.exp(mp.mpf(float(x)))
        if t > mp.mpf(2) ** -1022:
            worst = max(worst, float(abs(mp.mpf(float(y)) - t) / t))
        else:
            assert abs(mp.mpf(float(y)) - t) <= tiny * mp.mpf("0.51") + t * mp.mpf("3e-12"), (x, y)
    assert worst < 2.5e-12, worst
