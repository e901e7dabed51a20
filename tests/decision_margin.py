#!/usr/bin/env python
"""TEST INFRASTRUCTURE (uses the CPU oracle): decision-margin report for a bench workload.

    python tests/decision_margin.py [--workload cfg2] [--reads 1024] [--procs 8] --out profiles/r02/decision_margin_cfg2.json

The integer columns of a segmentation are decided by exact floating-point comparisons in the traceback
(`E[tBb] == M[tBb + eShift] + logScore`, NT_aligner_api.cpp:448, i.e. vM(t-1,n) >= vE(t-1,n)). The GPU
path evaluates the same expressions with its own softplus / emission arithmetic, which differs from
glibc's by rounding errors (~1e-13 accumulated on |vM - vE| at worst). It takes the same decisions as
long as the smallest on-path |vM - vE| is far above that. This script measures that margin with the
oracle over every read of a workload and records the distribution.

Structural ties are reported apart: where two neighbouring lattice columns carry the SAME k-mer (a
homopolymer of k+1 bases -- the polyA pad followed by an A in a quarter of the synthetic RNA reads, any
polyA tail in real data) "enter the column now" and "stay in it" are symmetric, the margin is zero in
exact arithmetic and the reference's own choice rests on rounding noise (0 or ~1e-11, all within the
first ~20 rows here). Those reads are listed so that the GPU parity test compares every one of them.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_O = None


def _init(model, pore_id):
    global _O
    from oracle.pyoracle import Oracle
    _O = Oracle(model, pore_id)


def _work(job):
    sig, seq = job
    r = _O.align(sig, seq, True)
    return _O.last_decision_margin(), len(r["signal_positions"]), _O.last_decision_margin_distinct(), _O.last_decision_margin_at()[:2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from dynamont_amd import synth
    cfg = synth.CONFIGS[a.workload]
    pore_id, _, k = synth.PORES[cfg["pore"]]
    tmp = tempfile.mkdtemp(prefix="margin_")
    model = synth.write_model(os.path.join(tmp, f"syn{k}.model"), k, seed=7, stdev=0.25 if k == 5 else 0.15)
    _, mean, sd = synth.read_model_file(model)
    n = a.reads or cfg["n_reads"]
    reads = synth.make_reads(cfg["seed"], n, cfg["pore"], mean, sd, cfg["n_bases"])
    t0 = time.time()
    with mp.get_context("fork").Pool(a.procs, initializer=_init, initargs=(model, pore_id)) as pool:
        res = pool.map(_work, [(r.signal, r.sequence) for r in reads], chunksize=4)
    m = np.array([x[0] for x in res])
    md = np.array([x[2] for x in res])
    segs = int(sum(x[1] for x in res))
    low = [int(i) for i in np.nonzero(m < 1e-6)[0]]
    out = {
        "workload": f"{a.workload}: first {n} reads (seed {cfg['seed']}), {cfg['pore']}, synthetic {k}-mer model",
        "what": "min over on-path traceback decisions of |vM(t-1,n) - vE(t-1,n)| per read (oracle, glibc arithmetic)",
        "reads": int(n), "segments": segs, "decisions": int(sum(len(r.signal) for r in reads)),
        "distinct_kmer_decisions": {"min": float(md.min()), "argmin_read": int(md.argmin()),
                                    "percentiles": {str(p): float(np.percentile(md, p)) for p in (0.1, 1, 10, 50)},
                                    "reads_below_1e-6": int((md < 1e-6).sum())},
        "all_decisions": {"min": float(m.min()), "reads_below_1e-6": len(low), "reads_below_1e-9": int((m < 1e-9).sum()),
                          "percentiles": {str(p): float(np.percentile(m, p)) for p in (0.1, 1, 10, 50)}},
        # reads with a structural tie (same k-mer in neighbouring columns): read index, margin, (row, column), first bases
        "structural_tie_reads": [{"read": i, "margin": float(m[i]), "row_col": list(res[i][3]), "start": reads[i].sequence[:12]} for i in low],
        "wall_s": round(time.time() - t0, 1),
    }
    assert out["distinct_kmer_decisions"]["min"] >= 1e-9, out
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
