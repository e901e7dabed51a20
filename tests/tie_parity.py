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
fewer). Measured on 1 000 reads: 0 / 3 / 11 / 17 / 3 / 14 reads differ.
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
#include <cmath>
#include <vector>
static std::vector<dynmath::SoftplusNode> TAB;
static int g_mode = 1;
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


def tie_reads(n: int, mean, sd, pore: str = "rna002", seed: int = 5000):
    """Short RNA reads (k+1 .. 60 bases, dwell 2 / 3.5 / 10): each starts with the 9-A pad, i.e. with five identical
    5-mers -- every read contains structural ties, and they make up a large share of its few decisions."""
    rng = np.random.default_rng(7)
    k = synth.PORES[pore][2]
    return [synth.make_reads(seed + i, 1, pore, mean, sd, int(rng.integers(k + 1, 60)),
                             dwell=float(rng.choice([2.0, 3.5, 10.0])))[0] for i in range(n)]


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
                       (3, "3-operation emission + product logPlus"), (5, "z = fma(x, 1/sd, -mean/sd), reference tail + product logPlus")):
        rp.set_mode(mode)
        bad = differing_reads(rp, reads, want)
        print(f"{name:55s}: {len(bad):3d} of {sum(w is not None for w in want)} reads with borders differing from the reference")
