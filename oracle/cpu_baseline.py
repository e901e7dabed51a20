#!/usr/bin/env python
"""TEST/BENCH INFRASTRUCTURE ONLY -- the CPU baseline leg of bench.py.

Times NTAligner::align(calc_probabilities=true) (or ::train with --mode train) on a bounded sample
of the bench workload on the host cores of the box it runs on, one single-threaded aligner per
process (the reference scales by processes, segment.py:301-316). Uses the compiled reference
(oracle/_ref, kind "reference") when that binary is present, else the C restatement
(oracle/_build, kind "port"). The all-cores figure is the MEDIAN of --reps repetitions over the same
reads; a 1-core figure (one process, the first --one-core-reads reads) is reported beside it.
Never touches the GPU.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_AL = None
_MODE = "align"
_NK = 0


def _init(kind, model, pore_id, mode, num_kmers):
    global _AL, _MODE, _NK
    from oracle import pyoracle
    _AL = pyoracle.Reference(model, pore_id) if kind == "reference" else pyoracle.Oracle(model, pore_id)
    _MODE, _NK = mode, num_kmers


def _work(args):
    sig, seq = args
    t0 = time.perf_counter()
    if _MODE == "train":
        from oracle import pyoracle
        _AL.train(sig, seq, _NK) if isinstance(_AL, pyoracle.Reference) else _AL.train(sig, seq, dense=True)
    else:
        _AL.align(sig, seq, True)
    return time.perf_counter() - t0, len(sig)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--mode", default="align", choices=["align", "train"])
    ap.add_argument("--reads", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--one-core-reads", type=int, default=4)
    ap.add_argument("--procs", type=int, default=min(os.cpu_count() or 1, 16))
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    from dynamont_amd import synth
    from oracle import pyoracle
    kind = "reference" if pyoracle.reference_available() else "port"
    if kind == "port":
        pyoracle.build("oracle")
    cfgname = {"cfg2_small": "cfg2", "cfg4_share": "cfg4", "cfg5_share": "cfg5"}.get(a.workload, a.workload)
    cfg = synth.CONFIGS[cfgname]
    pore_id, _, k = synth.PORES[cfg["pore"]]
    _, mean, sd = synth.read_model_file(a.model)
    n = min(a.reads, cfg["n_reads"])
    reads = synth.make_reads(cfg["seed"], n, cfg["pore"], mean, sd, cfg["n_bases"], polya=cfg.get("polya"))  # first n reads of the workload
    jobs = [(r.signal, r.sequence) for r in reads]
    procs = max(1, min(a.procs, n))
    init = (kind, a.model, pore_id, a.mode, 4 ** k)
    walls, core_s = [], []
    with mp.get_context("fork").Pool(procs, initializer=_init, initargs=init) as pool:
        pool.map(_work, jobs[:procs])  # warm: page in libm, first-touch allocator
        for _ in range(max(1, a.reps)):
            t0 = time.perf_counter()
            res = pool.map(_work, jobs, chunksize=1)
            walls.append(time.perf_counter() - t0)
            core_s.append(sum(r[0] for r in res))
    samples = sum(len(j[0]) for j in jobs)
    wall = statistics.median(walls)
    # one process alone on the machine
    n1 = max(1, min(a.one_core_reads, n))
    _init(*init)
    _work(jobs[0])
    t0 = time.perf_counter()
    for j in jobs[:n1]:
        _work(j)
    wall1 = time.perf_counter() - t0
    samples1 = sum(len(j[0]) for j in jobs[:n1])
    what = "align(calc_probabilities=true)" if a.mode == "align" else "train()"
    out = {
        "value": round(samples / wall / 1e6, 5), "unit": "Msamp/s", "cores": procs, "kind": kind,
        "sample": f"first {n} reads of {a.workload} ({samples} samples), {what}, {procs} single-threaded "
                  f"processes, median wall of {len(walls)} repetitions {wall:.2f} s "
                  f"(all: {', '.join(f'{w:.2f}' for w in walls)}), {statistics.median(core_s):.1f} core-s",
        "reads_per_s": round(n / wall, 3),
        "per_core_reads_per_s": round(n / statistics.median(core_s), 3),
        "one_core": {"value": round(samples1 / wall1 / 1e6, 5), "unit": "Msamp/s", "reads_per_s": round(n1 / wall1, 4),
                     "sample": f"first {n1} reads, one process alone, wall {wall1:.2f} s"},
    }
    json.dump(out, open(a.out, "w"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
