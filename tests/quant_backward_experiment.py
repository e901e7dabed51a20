#!/usr/bin/env python
"""EXPERIMENT (test infrastructure, CPU only): would a float32 backward lattice keep the integer columns?

VERDICT r2 item 4 proposed storing the backward rows as row-normalised float32 (+ one fp64 offset per row): 12.2 instead
of 20.1 bytes per lattice cell, both dominant HBM streams of k_read_queue<JOB_ALIGN> halved. The acceptance bar is
integer-exact segment borders on every read. That question does not need a GPU: oracle/nt_oracle.c compiled with
-DNTO_QUANT_BE replaces bE by offset + float(bE - offset) after the backward pass (and rebuilds bM from the stored
values, as the fused forward sweep does) -- everything else is the oracle, i.e. the reference's own arithmetic.

    python tests/quant_backward_experiment.py [out.json]

Reads: the G10 families (tie-bearing) and 64 reads of BASELINE config 2 (2 000 bases, ~20 k samples).
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynamont_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402
import tie_parity  # noqa: E402


def build(outdir):
    so = os.path.join(outdir, "libnt_quant.so")
    subprocess.run(["gcc", "-std=c11", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-DNTO_QUANT_BE", "-o", so,
                    os.path.join(ROOT, "oracle", "nt_oracle.c"), "-lm"], check=True)
    return so


_Q = _O = None


def _init(so, model, pore):
    global _Q, _O
    saved = pyoracle.ORACLE_SO
    pyoracle.ORACLE_SO = so
    _Q = pyoracle.Oracle(model, pore, 400)
    pyoracle.ORACLE_SO = saved
    _O = pyoracle.Oracle(model, pore, 400)


def _one(job):
    sig, seq = job
    a, b = _O.align(sig, seq, True), _Q.align(sig, seq, True)
    same = np.array_equal(a["signal_positions"], b["signal_positions"])
    moved = int((a["signal_positions"] != b["signal_positions"]).sum()) if len(a["signal_positions"]) == len(b["signal_positions"]) else -1
    dp = float(np.abs(a["probabilities"] - b["probabilities"]).max()) if same and len(a["probabilities"]) else None
    return same, moved, len(a["signal_positions"]), dp


def main():
    d = tempfile.mkdtemp(prefix="quant_")
    so = build(d)
    paths = tie_parity.g10_model_paths(d)
    sets = {}
    for fam, (pore, mkey, gen) in tie_parity.G10_FAMILIES.items():
        _, mean, sd = synth.read_model_file(paths[mkey])
        sets["G10 " + fam] = (pore, paths[mkey], gen(mean, sd))
    _, mean, sd = synth.read_model_file(paths["syn9"])
    cfg = synth.CONFIGS["cfg2"]
    sets["cfg2 (first 64 reads)"] = (cfg["pore"], paths["syn9"], synth.make_reads(cfg["seed"], 64, cfg["pore"], mean, sd, cfg["n_bases"]))
    report = {"experiment": "backward rows stored as offset + float32(bE - offset), decisions otherwise in the oracle's fp64", "sets": {}}
    for name, (pore, model, reads) in sets.items():
        with Pool(min(8, os.cpu_count() or 1), initializer=_init, initargs=(so, model, synth.PORES[pore][0])) as pool:
            res = pool.map(_one, [(r.signal, r.sequence) for r in reads], chunksize=4)
        bad = [i for i, r in enumerate(res) if not r[0]]
        dps = [r[3] for r in res if r[3] is not None]
        report["sets"][name] = dict(reads=len(reads), segments=int(sum(r[2] for r in res)), reads_with_moved_borders=len(bad),
                                    first_read_ids=bad[:20], borders_moved=int(sum(max(0, r[1]) for r in res)),
                                    max_abs_dprob_on_unchanged_reads=max(dps) if dps else None)
        print(name, report["sets"][name], flush=True)
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03", "ab_fp32_backward_rows.json")
    json.dump(report, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
