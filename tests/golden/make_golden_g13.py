#!/usr/bin/env python
"""Generate golden fixture G13 -- the traceback's DECISION MARGIN where it could be small -- and
profiles/r05/decision_margin_families.json. Authoring container only (compiled reference: `make -C oracle ref`; the
reference's model files are read in place from /root/reference/models and only their (mean, stdev) VALUES go into the fixture).

    python tests/golden/make_golden_g13.py [workers]

The default strict mode ("ties") certifies a read bit for bit only when two neighbouring columns carry identical emission
parameters; every other read relies on the on-path margin |vM(t-1,n) - vE(t-1,n)| of the comparison the traceback takes
(reference src/cpp/NT_aligner_api.cpp:445-448) staying far above the ~1e-13 by which the table softplus can move it. Round 2
measured that margin on ONE workload (cfg2: synthetic 9-mer table, N(0,1) means, one stdev): 1.04e-6. This fixture measures it
on the tables where distinct k-mers lie closest together:
  rna002_real     the reference's models/rna/rna002/rna002_5mer.model (closest non-identical neighbours 2e-4 apart)
  rna002_trained  models/rna/rna002/trained_rna002_5mer.model: a stdev per k-mer
  clustered9      a 9-mer table (RNA004 pore) made of models/rna/rna004/rna004_5mer.model: every 9-mer takes the entry of its
                  central 5-mer, moved by 1e-4, 1e-5, 1e-6 or 1e-7 (256 nine-mers share each 5-mer's level: what a 262 144-entry
                  table with real-like clustering looks like) -- stdev of the 5-mer
  dna_cfg3        BASELINE configs[2]: DNA r10.4.1 400 bps, reads of up to 100 k samples (5x cfg2's rows per read)
For every read: the COMPILED REFERENCE's borders and Z; the oracle's (bit-equal, tests/test_oracle_golden.py) smallest on-path
margin between columns with DIFFERENT parameters. tests/test_gpu_parity.py runs every read on the device in the default
mode against these borders."""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynamont_amd import synth  # noqa: E402
from oracle.pyoracle import Oracle, Reference  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF_MODELS = "/root/reference/models/rna"

FAMILIES = {
    # name: (pore, table, reads, bases, seed)
    "rna002_real": ("rna002", "rna002_real", 240, (200, 600), 1301),
    "rna002_trained": ("rna002", "rna002_trained", 240, (200, 600), 1302),
    "clustered9": ("rna004", "clustered9", 240, (800, 2000), 1303),
    "dna_cfg3": ("dna_r10_400bps", "syn9", 96, (800, 8000), 1304),
}


def clustered9(mean5, sd5):
    """every 9-mer gets the entry of its central 5-mer (code digits 2..6 of 9, base 4), moved by +-10^-(4 + code % 4)"""
    code = np.arange(4 ** 9, dtype=np.int64)
    central = (code // 16) % 1024
    sign = np.where((code * 2654435761 >> 7) & 1, 1.0, -1.0)
    jitter = sign * 10.0 ** -(4 + code % 4)
    return mean5[central] + jitter, sd5[central].copy()


def tables():
    t = {}
    for name, path in (("rna002_real", f"{REF_MODELS}/rna002/rna002_5mer.model"), ("rna002_trained", f"{REF_MODELS}/rna002/trained_rna002_5mer.model"),
                       ("rna004_5", f"{REF_MODELS}/rna004/rna004_5mer.model")):
        _, mean, sd = synth.read_model_file(path)
        t[name] = (np.asarray(mean, dtype=np.float64), np.asarray(sd, dtype=np.float64))
    return t


_W = {}


def _init(model, pore):
    enum = synth.PORES[pore][0]
    _W["ref"] = Reference(model, enum, 400)
    _W["orc"] = Oracle(model, enum, 400)


def _one(job):
    sig, seq = job
    want = _W["ref"].align(sig, seq, True)
    got = _W["orc"].align(sig, seq, True)
    assert np.array_equal(want["signal_positions"], got["signal_positions"]) and want["Z"] == got["Z"]
    return (np.asarray(want["signal_positions"], dtype=np.int32), float(want["Z"]), float(_W["orc"].last_decision_margin_distinct()),
            float(_W["orc"].last_decision_margin()))


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    tb = tables()
    tmp = tempfile.mkdtemp(prefix="g13_")
    fixture = {"rna002_real_mean": tb["rna002_real"][0], "rna002_real_sd": tb["rna002_real"][1],
               "rna002_trained_mean": tb["rna002_trained"][0], "rna002_trained_sd": tb["rna002_trained"][1],
               "rna004_5_mean": tb["rna004_5"][0], "rna004_5_sd": tb["rna004_5"][1]}
    report = {"what": "min over on-path traceback decisions between columns with DIFFERENT emission parameters of |vM(t-1,n) - vE(t-1,n)| "
                      "(oracle = compiled reference bit for bit); `all` includes structural ties (identical parameters: certified reads)",
              "families": {}}
    for fam, (pore, table, n, bases, seed) in FAMILIES.items():
        k = synth.PORES[pore][2]
        if table == "syn9":
            model = synth.write_model(os.path.join(tmp, "syn9.model"), 9, seed=7, stdev=0.15)
            _, mean, sd = synth.read_model_file(model)
        else:
            mean, sd = clustered9(*tb["rna004_5"]) if table == "clustered9" else tb[table]
            model = synth.write_model_values(os.path.join(tmp, f"{fam}.model"), k, mean, sd)
            _, mean, sd = synth.read_model_file(model)  # as the aligner will read them
        reads = synth.make_reads(seed, n, pore, np.asarray(mean), np.asarray(sd), bases)
        t0 = time.time()
        with mp.get_context("fork").Pool(workers, initializer=_init, initargs=(model, pore)) as pool:
            res = pool.map(_one, [(r.signal, r.sequence) for r in reads], chunksize=2)
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(r[0]) for r in res], out=offs[1:])
        fixture[f"{fam}_signal_positions"] = np.concatenate([r[0] for r in res])
        fixture[f"{fam}_offsets"] = offs
        fixture[f"{fam}_Z"] = np.array([r[1] for r in res])
        md = np.array([r[2] for r in res])
        ma = np.array([r[3] for r in res])
        fixture[f"{fam}_margin_distinct"] = md
        samples = int(sum(len(r.signal) for r in reads))
        report["families"][fam] = {
            "pore": pore, "table": table, "reads": n, "bases": list(bases), "seed": seed, "samples": samples, "segments": int(offs[-1]),
            "distinct_parameter_decisions": {"min": float(md.min()), "argmin_read": int(md.argmin()),
                                             "percentiles": {str(p): float(np.percentile(md, p)) for p in (1, 10, 50)},
                                             "reads_below_1e-6": int((md < 1e-6).sum()), "reads_below_1e-9": int((md < 1e-9).sum())},
            "all_decisions": {"min": float(ma.min()), "reads_below_1e-9": int((ma < 1e-9).sum())},
            "wall_s": round(time.time() - t0, 1)}
        print(fam, json.dumps(report["families"][fam]), flush=True)
    fixture["families"] = np.array(json.dumps({f: {"pore": v[0], "table": v[1], "reads": v[2], "bases": list(v[3]), "seed": v[4]} for f, v in FAMILIES.items()}))
    np.savez_compressed(os.path.join(OUT, "g13_margin_families.npz"), **fixture)
    os.makedirs(os.path.join(ROOT, "profiles", "r05"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "profiles", "r05", "decision_margin_families.json"), "w"), indent=1)
    print("written", os.path.getsize(os.path.join(OUT, "g13_margin_families.npz")), "bytes")


if __name__ == "__main__":
    main()
