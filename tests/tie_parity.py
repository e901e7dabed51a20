"""Replay of the oracle's control flow with the PRODUCT's arithmetic (test infrastructure).

Where two neighbouring lattice columns carry the same k-mer (a homopolymer of k+1 bases: every RNA read starts with
the polyA pad) the traceback's exact comparison (NT_aligner_api.cpp:448) is a tie in exact arithmetic, and the
reference's choice rests on the last bits of its sums. Whether a GPU build reproduces those choices is therefore a
property of its arithmetic primitives (dp_math.hpp), not of the kernels' indexing -- and it can be measured without a
GPU: ``oracle/nt_oracle.c`` compiled with ``-DNTO_ARITH_HOOKS`` takes its two primitives (log-normal density, logPlus)
from this module's C++ shim, which calls dp_math.hpp compiled for the host.

    python tests/tie_parity.py [n_reads]        # table: reads whose borders differ from the reference, per arithmetic

Modes of the shim: 0 = libm (must reproduce the oracle exactly), 1 = the product (dp_math.hpp as it stands),
2 = the 4-operation emission of rounds 1-2 (constants pre-added, one FMA), 3 = a 3-operation emission
(1/(stdev sqrt 2) folded into the constant), 4 = the product's logPlus with the reference's emission,
5 = the product's emission with the difference folded into an FMA (z = fma(x, 1/stdev, -mean/stdev), one operation
fewer), 6 = the STRICT arithmetic (dp_math_strict.hpp: glibc's exp and log1p restated bit for bit, the reference's
emission expression), 7 = the CERTIFIED arithmetic of round 4 (table softplus + rounding certificate, the restated glibc
only for ambiguous sums; division-free exact emission) -- what the kernels run for flagged reads, 8 = strict mode "ties"
read by read (Replay.set_strict_rows: the certified arithmetic exactly where the kernels use it, the plain one elsewhere).
Measured on 1 000 reads: 0 / 3 / 11 / 17 / 3 / 14 / 0 / 0 reads differ.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dynamont_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

SHIM = r'''
#include "%(root)s/dynamont_amd/csrc/dp_math.hpp"
#include "%(root)s/dynamont_amd/csrc/dp_math_strict.hpp"
#include <cmath>
#include <vector>
static std::vector<dynmath::SoftplusNode> TAB;
static int g_mode = 1;
static long g_calls = 0, g_ambiguous = 0;
// mode 8: what a handle in strict mode "ties" does with ONE read -- strict_rows as dyn_tie_rows returns it (0: the read
// carries no structural tie and runs in the plain arithmetic throughout; else certified backward sweep, certified forward
// rows in whole 64-row blocks while the block starts at a row <= strict_rows, each row's emission in the flavour of the
// row before it: nt_kernels.hip, forward_sweep)
static unsigned long g_strict_rows = 0;
static bool g_lp_cert = false, g_pdf_cert = false;
extern "C" void replay_strict_rows(unsigned long r) { g_strict_rows = r; }
extern "C" void nto_hook_row(int phase, uint64_t t) {
  if (g_mode != 8) return;
  if (g_strict_rows == 0) { g_lp_cert = g_pdf_cert = false; return; }
  if (phase == 1) { g_lp_cert = g_pdf_cert = true; return; }
  auto block_start = [](uint64_t r) { return 1 + 64 * ((r - 1) / 64); };
  g_lp_cert = block_start(t) <= g_strict_rows;
  g_pdf_cert = t == 1 ? g_strict_rows >= 1 : block_start(t - 1) <= g_strict_rows;
}
extern "C" void replay_counts(long* out) { out[0] = g_calls; out[1] = g_ambiguous; g_calls = g_ambiguous = 0; }
extern "C" void replay_mode(int m) {
  g_mode = m;
  if (TAB.empty()) { TAB.resize(dynmath::SP_NODES); dynmath::softplus_build_table(TAB.data()); }
}
static double libm_pdf(double x, double mean, double sd) {
  const double diff = x - mean; const double z = diff / sd;
  return -0.5 * z * z - std::log(sd) - 0.5 * std::log(2.0 * M_PI);
}
extern "C" double nto_hook_pdf(double x, double mean, double sd) {
  switch (g_mode) {
    case 0: case 4: return libm_pdf(x, mean, sd);
    case 6: return dynmath::log_normal_pdf_strict(x, dynmath::make_emis(mean, sd, std::log(sd)));
    case 7: return dynmath::log_normal_pdf_cert(x, dynmath::make_emis(mean, sd, std::log(sd)));
    case 8: return g_pdf_cert ? dynmath::log_normal_pdf_cert(x, dynmath::make_emis(mean, sd, std::log(sd)))
                              : dynmath::log_normal_pdf(x, dynmath::make_emis(mean, sd, std::log(sd)));
    case 2: { const double z = (x - mean) * (1.0 / sd); return dynmath::fma_(-0.5 * z, z, -std::log(sd) - dynmath::HALF_LOG_2PI); }
    case 3: { const double k = (double)(1.0L / ((long double)sd * 1.41421356237309504880168872420969808L));
              const double y = (x - mean) * k; return dynmath::fma_(-y, y, -std::log(sd) - dynmath::HALF_LOG_2PI); }
    case 5: { const double inv = 1.0 / sd; const double z = dynmath::fma_(x, inv, -(mean * inv)); const double t = z * z;
              return dynmath::fma_(t, -0.5, -std::log(sd)) - dynmath::HALF_LOG_2PI; }
    default: return dynmath::log_normal_pdf(x, dynmath::make_emis(mean, sd, std::log(sd)));
  }
}
extern "C" double nto_hook_log_plus(double x, double y) {
  if (g_mode == 0) {
    if (std::isinf(x)) return y;
    if (std::isinf(y)) return x;
    if (x < y) { const double t = x; x = y; y = t; }
    return x + std::log1p(std::exp(y - x));
  }
  if (g_mode == 6) return dynmath::log_plus_strict(x, y, dynmath::strict_exp_table());
  if (g_mode == 7 || (g_mode == 8 && g_lp_cert)) { ++g_calls; return dynmath::log_plus_cert(x, y, TAB.data(), dynmath::strict_exp_table(), &g_ambiguous); }
  double a[1] = {x}, b[1] = {y}, o[1];
  dynmath::SoftplusLookup<1> L;
  dynmath::log_plus_issue<1>(a, b, L, TAB.data());
  dynmath::log_plus_finish<1>(L, o);
  return o[0];
}
'''


def build_replay(outdir: str) -> str:
    """gcc/g++ -> <outdir>/libnt_replay.so (oracle control flow + hooked primitives)."""
    shim = os.path.join(outdir, "replay_shim.cpp")
    with open(shim, "w") as f:
        f.write(SHIM % {"root": ROOT})
    obj_c, obj_s, so = (os.path.join(outdir, n) for n in ("nt_replay.o", "replay_shim.o", "libnt_replay.so"))
    subprocess.run(["gcc", "-std=c11", "-O2", "-fPIC", "-ffp-contract=off", "-DNTO_ARITH_HOOKS", "-c",
                    os.path.join(ROOT, "oracle", "nt_oracle.c"), "-o", obj_c], check=True)
    subprocess.run(["g++", "-O2", "-fPIC", "-ffp-contract=off", "-c", shim, "-o", obj_s], check=True)
    subprocess.run(["g++", "-shared", "-o", so, obj_c, obj_s, "-lm"], check=True)
    return so


class Replay(pyoracle.Oracle):
    """pyoracle.Oracle over the replay library; ``mode`` as in the module docstring."""

    def __init__(self, so: str, model_path: str, pore: int, band: int = 400, mode: int = 1):
        saved = pyoracle.ORACLE_SO
        pyoracle.ORACLE_SO = so
        try:
            super().__init__(model_path, pore, band)
        finally:
            pyoracle.ORACLE_SO = saved
        self.lib.replay_mode(int(mode))

    def set_mode(self, mode: int):
        self.lib.replay_mode(int(mode))

    def set_strict_rows(self, rows: int):
        """mode 8: dyn_tie_rows of the read that is aligned next (0 = plain arithmetic throughout)"""
        self.lib.replay_strict_rows(C.c_ulong(int(rows)))

    def counts(self):
        """mode 7: (logPlus calls, ambiguous ones) since the last call"""
        out = (C.c_long * 2)()
        self.lib.replay_counts(out)
        return int(out[0]), int(out[1])


def tie_reads(n: int, mean, sd, pore: str = "rna002", seed: int = 5000):
    """Short RNA reads (k+1 .. 60 bases, dwell 2 / 3.5 / 10): each starts with the 9-A pad, i.e. with five identical
    5-mers -- every read contains structural ties, and they make up a large share of its few decisions."""
    rng = np.random.default_rng(7)
    k = synth.PORES[pore][2]
    return [synth.make_reads(seed + i, 1, pore, mean, sd, int(rng.integers(k + 1, 60)),
                             dwell=float(rng.choice([2.0, 3.5, 10.0])))[0] for i in range(n)]


def start_tie_reads(n: int, mean, sd, pore: str, seed: int, bases=None):
    """Short reads (k+2 .. 70 bases, or ``bases`` = (lo, hi)) that START with a homopolymer of k+1 .. k+4 bases: columns 1
    and 2 (and more) carry the same k-mer, the symmetric read-start tie, for any pore (DNA reads have no pad; RNA 9-mer
    reads need pad + A). With 450+ bases the band is narrower than the read and the tied columns leave it part-way through:
    strict mode "ties" then certifies a PREFIX of the forward rows only."""
    _, rna, k = synth.PORES[pore]
    mean_c, sd_c = synth.code_order_table(mean, sd, k, rna)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nb = int(rng.integers(k + 2, 71)) if bases is None else int(rng.integers(bases[0], bases[1] + 1))
        digits = rng.integers(0, 4, size=nb)
        run = int(rng.integers(k + 1, min(nb, k + 5)))
        digits[:run] = 0 if rna else int(rng.integers(0, 4))
        if rna:
            digits[:9] = 0
        out.append(synth.read_from_digits(rng, digits, mean_c, sd_c, k, float(rng.choice([2.0, 3.5, 10.0]))))
    return out


def internal_homopolymer_reads(n: int, mean, sd, pore: str, seed: int):
    """Reads of 150 .. 400 bases with ONE internal homopolymer run of 20 .. 120 bases (>= k+1: a stretch of identical
    k-mers away from the read start, where the history on both sides of a tie is no longer symmetric)."""
    _, rna, k = synth.PORES[pore]
    mean_c, sd_c = synth.code_order_table(mean, sd, k, rna)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nb = int(rng.integers(150, 401))
        digits = rng.integers(0, 4, size=nb)
        if rna:
            digits[:9] = 0
            digits[9] = int(rng.integers(1, 4))  # no read-start tie in this family
        run = int(rng.integers(20, min(121, nb - 40)))
        at = int(rng.integers(20, nb - run - 10))
        digits[at:at + run] = int(rng.integers(0, 4))
        out.append(synth.read_from_digits(rng, digits, mean_c, sd_c, k, 10.0 if rna else 12.5))
    return out


def offset_tie_reads(n: int, mean, sd, pore: str, seed: int):
    """Short reads with a homopolymer of k+1 .. k+3 bases that starts 1, 2 or 3 bases into the read (DNA) / behind the
    polyA pad (RNA; the bases in between are neither A nor the run's base, so the pad ties nothing itself): k-mers j and
    j+1 are equal for a j close to the read start that is NOT the pair (0, 1) round 3's rule looked at."""
    _, rna, k = synth.PORES[pore]
    mean_c, sd_c = synth.code_order_table(mean, sd, k, rna)
    rng = np.random.default_rng(seed)
    lead = 9 if rna else 0
    out = []
    for _ in range(n):
        nb = int(rng.integers(lead + k + 8, lead + 81))
        digits = rng.integers(0, 4, size=nb)
        j = lead + int(rng.integers(1, 4))  # first base of the run
        run = int(rng.integers(k + 1, k + 4))
        b = int(rng.integers(1, 4)) if rna else int(rng.integers(0, 4))
        if rna:
            digits[:9] = 0
            others = [x for x in (1, 2, 3) if x != b]
            digits[9:j] = rng.choice(others, size=j - 9)
        elif digits[j - 1] == b:
            digits[j - 1] = (b + 1 + int(rng.integers(0, 3))) % 4
        digits[j:j + run] = b
        if digits[j + run] == b:
            digits[j + run] = (b + 1 + int(rng.integers(0, 3))) % 4
        out.append(synth.read_from_digits(rng, digits, mean_c, sd_c, k, float(rng.choice([2.0, 3.5, 10.0]))))
    return out


def late_tie_reads(n: int, mean, sd, pore: str, seed: int):
    """Reads of 300 .. 420 bases, dwell 10 (S ~ 3 000 .. 4 500), whose one homopolymer run (k+1 .. 40 bases) starts at
    base 150 or later: every tied decision lies beyond lattice row 1 024, round 3's fixed number of strict rows."""
    _, rna, k = synth.PORES[pore]
    mean_c, sd_c = synth.code_order_table(mean, sd, k, rna)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nb = int(rng.integers(300, 421))
        digits = rng.integers(0, 4, size=nb)
        if rna:
            digits[:9] = 0
            digits[9] = int(rng.integers(1, 4))
        run = int(rng.integers(k + 1, 41))
        at = int(rng.integers(150, nb - run - 10))
        digits[at:at + run] = int(rng.integers(0, 4))
        out.append(synth.read_from_digits(rng, digits, mean_c, sd_c, k, 10.0))
    return out


def random_reads(n: int, mean, sd, pore: str, seed: int, bases=(40, 160)):
    """Random short reads as they come (RNA: behind the pad and one non-A base, so that the pad itself ties nothing)."""
    _, rna, k = synth.PORES[pore]
    mean_c, sd_c = synth.code_order_table(mean, sd, k, rna)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nb = int(rng.integers(bases[0], bases[1] + 1))
        digits = rng.integers(0, 4, size=nb)
        if rna:
            digits[:9] = 0
            digits[9] = int(rng.integers(1, 4))
        out.append(synth.read_from_digits(rng, digits, mean_c, sd_c, k, float(rng.choice([3.5, 10.0]))))
    return out


# Fixture G12 (tests/golden/make_golden_g12.py), the widened tie evidence of round 4: family -> (pore, model, generator).
# "syn5_r1": the 5-mer model with its means rounded to one decimal -- distinct k-mers with IDENTICAL (mean, stdev), so
# that neighbouring columns tie without a homopolymer (what a quantised real table does).
G12_MODELS = {"syn5": (5, 0.25, None), "syn9": (9, 0.15, None), "syn5_sd015": (5, 0.15, None), "syn5_r1": (5, 0.15, 1)}
G12_FAMILIES = {
    "rna002_internal": ("rna002", "syn5_sd015", lambda m, s: internal_homopolymer_reads(3000, m, s, "rna002", 7100)),
    "rna004_internal": ("rna004", "syn9", lambda m, s: internal_homopolymer_reads(3000, m, s, "rna004", 7200)),
    "dna_r10_400_internal": ("dna_r10_400bps", "syn9", lambda m, s: internal_homopolymer_reads(3000, m, s, "dna_r10_400bps", 7300)),
    "dna_r9_internal": ("dna_r9", "syn5_sd015", lambda m, s: internal_homopolymer_reads(3000, m, s, "dna_r9", 7400)),
    "dna_r9_offset": ("dna_r9", "syn5_sd015", lambda m, s: offset_tie_reads(1500, m, s, "dna_r9", 7500)),
    "rna004_offset": ("rna004", "syn9", lambda m, s: offset_tie_reads(1000, m, s, "rna004", 7600)),
    "dna_r10_400_offset": ("dna_r10_400bps", "syn9", lambda m, s: offset_tie_reads(1000, m, s, "dna_r10_400bps", 7700)),
    "dna_r9_late": ("dna_r9", "syn5_sd015", lambda m, s: late_tie_reads(1000, m, s, "dna_r9", 7800)),
    "rna004_late": ("rna004", "syn9", lambda m, s: late_tie_reads(1000, m, s, "rna004", 7900)),
    "rna004_start_long": ("rna004", "syn9", lambda m, s: start_tie_reads(400, m, s, "rna004", 7950, (450, 600))),
    "dna_r9_start_long": ("dna_r9", "syn5_sd015", lambda m, s: start_tie_reads(400, m, s, "dna_r9", 7960, (450, 600))),
    "dna_r9_rounded_model": ("dna_r9", "syn5_r1", lambda m, s: random_reads(2000, m, s, "dna_r9", 8000)),
    "dna_r9_random": ("dna_r9", "syn5_sd015", lambda m, s: random_reads(3000, m, s, "dna_r9", 8100)),
    "rna004_random": ("rna004", "syn9", lambda m, s: random_reads(2000, m, s, "rna004", 8200)),
}


def polya_tail_reads(n: int, mean, sd, pore: str, seed: int):
    """What a real direct-RNA read looks like to the aligner: the 3' polyA tail is sequenced first, so the reversed
    basecall STARTS with a long homopolymer -- pad + 20 .. 150 more A's here, then 150 .. 500 random bases. Dozens of
    neighbouring columns carry the same k-mer from column 0 on (the read-start tie of the G10 families, many columns wide)."""
    _, rna, k = synth.PORES[pore]
    mean_c, sd_c = synth.code_order_table(mean, sd, k, rna)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        run = int(rng.integers(20, 151))
        nb = 9 + run + int(rng.integers(150, 501))
        digits = rng.integers(0, 4, size=nb)
        digits[:9 + run] = 0
        digits[9 + run] = int(rng.integers(1, 4))
        out.append(synth.read_from_digits(rng, digits, mean_c, sd_c, k, float(rng.choice([5.0, 10.0]))))
    return out


# families beyond the committed fixture: run in full on the device by tests/tie_device_full.py (reference results from
# tests/golden/make_g12_full.py, not committed)
EXTRA_FAMILIES = {
    "rna004_polya_tail": ("rna004", "syn9", lambda m, s: polya_tail_reads(1000, m, s, "rna004", 8100)),
    "rna002_polya_tail": ("rna002", "syn5_sd015", lambda m, s: polya_tail_reads(1000, m, s, "rna002", 8200)),
}


def g12_model_paths(outdir: str) -> dict:
    out = {}
    for name, (k, sd, decimals) in G12_MODELS.items():
        path = os.path.join(outdir, name + ".model")
        if decimals is None:
            synth.write_model(path, k, seed=7, stdev=sd)
        else:
            mean, sdv = synth.model_values(k, 7, sd)
            synth.write_model_values(path, k, np.round(mean, decimals), sdv)
        out[name] = path
    return out


# Fixture G10 (tests/golden/make_golden_g10.py): family -> (pore, model, generator). Models: the synthetic ones of
# conftest ("syn5": 5-mer, stdev 0.25; "syn9": 9-mer, stdev 0.15) and "syn5_sd015", the 5-mer model with stdev 0.15 that
# `python tests/tie_parity.py` uses -- the configuration in which the default arithmetic is known to deviate.
G10_MODELS = {"syn5": (5, 0.25), "syn9": (9, 0.15), "syn5_sd015": (5, 0.15)}
G10_FAMILIES = {
    "rna002_start": ("rna002", "syn5", lambda m, s: tie_reads(1000, m, s, "rna002")),
    "rna002_start_sd015": ("rna002", "syn5_sd015", lambda m, s: tie_reads(1000, m, s, "rna002")),
    "rna004_start": ("rna004", "syn9", lambda m, s: start_tie_reads(300, m, s, "rna004", 6100)),
    "dna_r10_400_start": ("dna_r10_400bps", "syn9", lambda m, s: start_tie_reads(300, m, s, "dna_r10_400bps", 6200)),
    "dna_r9_start": ("dna_r9", "syn5", lambda m, s: start_tie_reads(200, m, s, "dna_r9", 6300)),
    "dna_r9_start_sd015": ("dna_r9", "syn5_sd015", lambda m, s: start_tie_reads(300, m, s, "dna_r9", 6350)),
    "rna002_internal": ("rna002", "syn5", lambda m, s: internal_homopolymer_reads(100, m, s, "rna002", 6400)),
    "rna004_internal": ("rna004", "syn9", lambda m, s: internal_homopolymer_reads(100, m, s, "rna004", 6500)),
    "dna_r10_400_internal": ("dna_r10_400bps", "syn9", lambda m, s: internal_homopolymer_reads(100, m, s, "dna_r10_400bps", 6600)),
}


def g10_model_paths(outdir: str) -> dict:
    return {name: synth.write_model(os.path.join(outdir, name + ".model"), k, seed=7, stdev=sd) for name, (k, sd) in G10_MODELS.items()}


def borders_equal(a: dict, b: dict) -> bool:
    return (np.array_equal(a["signal_positions"], b["signal_positions"])
            and np.array_equal(a["sequence_positions"], b["sequence_positions"]))


def differing_reads(engine, reads, want) -> list[int]:
    out = []
    for i, (r, w) in enumerate(zip(reads, want)):
        if w is None:
            continue
        if not borders_equal(engine.align(r.signal, r.sequence, True), w):
            out.append(i)
    return out


def reference_results(orc, reads):
    want = []
    for r in reads:
        try:
            want.append(orc.align(r.signal, r.sequence, True))
        except RuntimeError:
            want.append(None)
    return want


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    d = tempfile.mkdtemp(prefix="tie_parity_")
    pore = "rna002"
    path = synth.write_model(os.path.join(d, "m.model"), synth.PORES[pore][2])
    _, mean, sd = synth.read_model_file(path)
    reads = tie_reads(n, mean, sd, pore)
    want = reference_results(pyoracle.Oracle(path, synth.PORES[pore][0], 400), reads)
    rp = Replay(build_replay(d), path, synth.PORES[pore][0], 400)
    for mode, name in ((0, "libm primitives (sanity: the oracle itself)"), (1, "product arithmetic (dp_math.hpp)"),
                       (4, "reference emission + product logPlus"), (2, "4-operation emission of rounds 1-2 + product logPlus"),
                       (3, "3-operation emission + product logPlus"), (5, "z = fma(x, 1/sd, -mean/sd), reference tail + product logPlus"),
                       (6, "strict arithmetic (dp_math_strict.hpp)"), (7, "certified arithmetic (table softplus + certificate)")):
        rp.set_mode(mode)
        bad = differing_reads(rp, reads, want)
        print(f"{name:55s}: {len(bad):3d} of {sum(w is not None for w in want)} reads with borders differing from the reference")
