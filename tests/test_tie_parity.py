"""Structural ties (same k-mer in neighbouring columns): how faithfully the product's arithmetic reproduces the
reference's rounding-noise decisions. CPU: the oracle's control flow replayed with dp_math.hpp (tests/tie_parity.py).
GPU: the kernels on the same reads -- they must agree with the replay (which is what makes the CPU measurement
meaningful) and differ from the reference on no more reads than it does."""
import os

import numpy as np
import pytest

from conftest import model_for
from dynamont_amd import synth
from oracle import pyoracle
import tie_parity

PORE = "rna002"
N_READS = 300


@pytest.fixture(scope="module")
def tie_setup(models, oracle_built, tmp_path_factory):
    path = model_for(models, PORE)
    _, mean, sd = synth.read_model_file(path)
    reads = tie_parity.tie_reads(N_READS, mean, sd, PORE)
    orc = pyoracle.Oracle(path, synth.PORES[PORE][0], 400)
    want = tie_parity.reference_results(orc, reads)
    so = tie_parity.build_replay(str(tmp_path_factory.mktemp("replay")))
    return path, reads, want, tie_parity.Replay(so, path, synth.PORES[PORE][0], 400)


def test_replay_with_libm_primitives_is_the_oracle(tie_setup):
    _, reads, want, rp = tie_setup
    rp.set_mode(0)
    assert tie_parity.differing_reads(rp, reads, want) == []


def test_product_arithmetic_reproduces_tie_decisions(tie_setup):
    """Measured on 1 000 such reads: 3 differ (the <= 1 ulp of the logPlus; the reference's own emission gives the same 3);
    the 4-operation emission of rounds 1-2 gave 11, a 3-operation form 17."""
    _, reads, want, rp = tie_setup
    rp.set_mode(1)
    bad = tie_parity.differing_reads(rp, reads, want)
    assert len(bad) <= 3, bad
    rp.set_mode(4)   # reference emission + product logPlus: the emission is not what is left
    assert len(tie_parity.differing_reads(rp, reads, want)) >= len(bad) - 1


@pytest.mark.gpu
def test_gpu_tie_decisions_follow_the_replay(tie_setup):
    from dynamont_amd import Aligner
    path, reads, want, rp = tie_setup
    rp.set_mode(1)
    al = Aligner(path, PORE, band=400, device=0)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    vs_ref, vs_replay = [], []
    for i, (r, w) in enumerate(zip(reads, want)):
        if w is None:
            assert res.status[i] != 0
            continue
        assert res.status[i] == 0, res.error(i)
        got = res.read(i)
        if not tie_parity.borders_equal(got, w):
            vs_ref.append(i)
        if not tie_parity.borders_equal(got, rp.align(r.signal, r.sequence, True)):
            vs_replay.append(i)
        assert np.abs(got["probabilities"] - w["probabilities"]).max() <= 1e-6 or i in vs_ref
    print(f"tie reads: GPU differs from the reference on {len(vs_ref)}, from the replay on {len(vs_replay)} of {len(reads)}")
    assert vs_replay == [], vs_replay      # the kernels take every decision the way the replayed arithmetic does
    assert len(vs_ref) <= 3, vs_ref
