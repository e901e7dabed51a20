#!/usr/bin/env python
"""TEST/BENCH INFRASTRUCTURE ONLY -- the CPU baseline leg of bench.py.

Times NTAligner::align(calc_probabilities=true) on a bounded sample of the bench workload on the
host cores of the box it runs on, one single-threaded aligner per process (the reference scales
by processes, segment.py:301-316). Uses the compiled reference (oracle/_ref, kind "reference")
when that binary is present, else the C restatement (oracle/_build, kind "port").
Never touches the GPU.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_AL = None


def _init(kind, model, pore_id):
    global _AL
    from oracle import pyoracle
    _AL = pyoracle.Reference(model, pore_id) if kind == "reference" else pyoracle.Oracle(model, pore_id)


def _work(args):
    sig, seq = args
    t0 = time.perf_counter()
    r = _AL.align(sig, seq, True)
    return time.perf_counter() - t0, len(sig), len(r["sequence_positions"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--reads", type=int, default=16)
    ap.add_argument("--procs", type=int, default=min(os.cpu_count() or 1, 16))
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    from dynamont_amd import synth
    from oracle import pyoracle
    kind = "reference" if pyoracle.reference_available() else "port"
    if kind == "port":
        pyoracle.build("oracle")
    cfgname = "cfg2" if a.workload == "cfg2_small" else a.workload
    cfg = synth.CONFIGS[cfgname]
    pore_id = synth.PORES[cfg["pore"]][0]
    _, mean, sd = synth.read_model_file(a.model)
    n = min(a.reads, cfg["n_reads"])
    reads = synth.make_reads(cfg["seed"], n, cfg["pore"], mean, sd, cfg["n_bases"])  # first n reads of the workload
    jobs = [(r.signal, r.sequence) for r in reads]
    procs = max(1, min(a.procs, n))
    with mp.get_context("fork").Pool(procs, initializer=_init, initargs=(kind, a.model, pore_id)) as pool:
        pool.map(_work, jobs[:procs])  # warm: page in libm, first-touch allocator
        t0 = time.perf_counter()
        res = pool.map(_work, jobs, chunksize=1)
        wall = time.perf_counter() - t0
    samples = sum(r[1] for r in res)
    out = {
        "value": round(samples / wall / 1e6, 5), "unit": "Msamp/s", "cores": procs, "kind": kind,
        "sample": f"first {n} reads of {a.workload} ({samples} samples), align(calc_probabilities=true), "
                  f"{procs} single-threaded processes, wall {wall:.2f} s, "
                  f"{sum(r[0] for r in res):.1f} core-s",
        "reads_per_s": round(n / wall, 3),
        "per_core_reads_per_s": round(n / sum(r[0] for r in res), 3),
    }
    json.dump(out, open(a.out, "w"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
