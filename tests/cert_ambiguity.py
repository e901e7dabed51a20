"""How often is a certified logPlus ambiguous on the DP's own operands? (test infrastructure; CPU only)

The certified arithmetic (dynamont_amd/csrc/dp_math_strict.hpp, round 4) recomputes a sum with the restated glibc only
when a rounding boundary of the result's grid falls inside the certificate's interval. This script replays the oracle's
control flow with that arithmetic (tests/tie_parity.py mode 7) on full-size reads of BASELINE's workloads, checks that
the results are the oracle's bit for bit (borders, Z, probabilities) and writes the ambiguity rates to
profiles/r04/cert_ambiguity.json: per logPlus, and -- assuming independence -- per 64-cell register and per 448-cell row,
the granularities at which a kernel can fall back.

    python tests/cert_ambiguity.py [n_reads_per_workload]
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynamont_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402
import tie_parity  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    d = tempfile.mkdtemp(prefix="cert_amb_")
    so = tie_parity.build_replay(d)
    out = {}
    for name, pore, k, sd, bases in (("cfg2 (rna004 9-mer, ~20 k samples)", "rna004", 9, 0.15, 2000),
                                     ("cfg3 (dna_r10_400bps 9-mer, 10 k - 100 k samples)", "dna_r10_400bps", 9, 0.15, (800, 8000)),
                                     ("cfg1 (rna002 5-mer, ~2 k samples)", "rna002", 5, 0.25, 200)):
        path = synth.write_model(os.path.join(d, f"m{k}_{sd}.model"), k, seed=7, stdev=sd)
        _, mean, sdv = synth.read_model_file(path)
        reads = synth.make_reads(4242, n, pore, mean, sdv, bases)
        enum = synth.PORES[pore][0]
        orc = pyoracle.Oracle(path, enum, 400)
        rp = tie_parity.Replay(so, path, enum, 400, mode=7)
        rp.counts()
        calls = amb = 0
        zmax = 0.0
        for r in reads:
            want = orc.align(r.signal, r.sequence, True)
            got = rp.align(r.signal, r.sequence, True)
            assert np.array_equal(got["signal_positions"], want["signal_positions"])
            assert np.array_equal(got["sequence_positions"], want["sequence_positions"])
            assert got["Z"] == want["Z"], (got["Z"], want["Z"])
            assert np.array_equal(got["probabilities"], want["probabilities"])
            c, a = rp.counts()
            calls += c
            amb += a
            zmax = max(zmax, abs(want["Z"]))
        p = amb / calls
        out[name] = dict(reads=n, log_plus_calls=calls, ambiguous=amb, rate_per_call=p,
                         rate_per_64_cells=1 - (1 - p) ** 64, rate_per_448_cell_row=1 - (1 - p) ** 448, max_abs_Z=zmax,
                         results="bit-identical to the oracle (borders, Z, probabilities)")
        print(name, out[name], flush=True)
    with open(os.path.join(ROOT, "profiles", "r04", "cert_ambiguity.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
