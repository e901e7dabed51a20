#!/usr/bin/env python
"""The compiled reference's segment borders and Z for EVERY read of fixture G12's fourteen families (25 300 reads), and of
G10's (3 400) -- not a committed fixture (tens of MB): written to tests/golden/_full/ (git-ignored, travels with a gpurun
snapshot), so that tests/tie_device_full.py can run ALL of them on the device in the default configuration, where the
committed fixture and `pytest -m gpu` keep 48 per family. Authoring container only (oracle/_ref).

    python tests/golden/make_g12_full.py [workers]
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dynamont_amd import synth  # noqa: E402
from oracle.pyoracle import Reference  # noqa: E402
import tie_parity  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_full")
_W = {}


def _init(model, pore):
    _W["ref"] = Reference(model, synth.PORES[pore][0], 400)


def _one(job):
    sig, seq = job
    try:
        want = _W["ref"].align(sig, seq, True)
    except RuntimeError:
        return None
    return want["signal_positions"].astype(np.uint32), want["sequence_positions"].astype(np.uint32), want["Z"]


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    os.makedirs(OUT, exist_ok=True)
    d = tempfile.mkdtemp(prefix="g12full_")
    mpaths = tie_parity.g12_model_paths(d)
    store = {}
    for fam, (pore, mkey, gen) in {**tie_parity.G12_FAMILIES, **tie_parity.EXTRA_FAMILIES}.items():
        _, mean, sd = synth.read_model_file(mpaths[mkey])
        reads = gen(mean, sd)
        with mp.get_context("fork").Pool(workers, initializer=_init, initargs=(mpaths[mkey], pore)) as pool:
            res = pool.map(_one, [(r.signal, r.sequence) for r in reads], chunksize=8)
        seg_off = np.zeros(len(reads) + 1, dtype=np.int64)
        for i, r in enumerate(res):
            seg_off[i + 1] = seg_off[i] + (0 if r is None else len(r[0]))
        store[fam + "_S"] = np.array([len(r.signal) for r in reads], dtype=np.int32)  # guards the regenerated inputs
        store[fam + "_seg_off"] = seg_off
        store[fam + "_sigpos"] = np.concatenate([np.zeros(0, np.uint32)] + [r[0] for r in res if r is not None])
        store[fam + "_seqpos"] = np.concatenate([np.zeros(0, np.uint32)] + [r[1] for r in res if r is not None])
        store[fam + "_Z"] = np.array([np.nan if r is None else r[2] for r in res])
        print(fam, len(reads), "reads,", int(seg_off[-1]), "segments", flush=True)
    np.savez_compressed(os.path.join(OUT, "g12_all.npz"), **store)
    print("wrote", os.path.join(OUT, "g12_all.npz"), os.path.getsize(os.path.join(OUT, "g12_all.npz")) >> 20, "MiB")


if __name__ == "__main__":
    main()
