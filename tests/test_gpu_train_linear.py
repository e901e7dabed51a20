"""train() runs its sweeps in the LINEAR domain (probabilities with exact power-of-two row scaling, nt_kernels.hip
backward_train_lin / forward_train_lin); a read whose rows need more range than fp64 has is done again in the log
domain. These tests push on exactly that: samples far outside every k-mer's density, level changes that move all the
forward mass of a row from one state vector to the other, stalls, mis-fitting signal, unnormalised transition weights
(dna_r9: m1 = e2 = 1), and a forced redo of ordinary reads. The checker is the oracle (reference arithmetic).

Tolerances: the oracle's own weights carry ~sqrt(T) x 7e-12 x |log Z| / 4e4 of rounding noise (80-bit adjudication:
tests/extended_precision_train.py), so weights compare at 1e-7 relative; Z at 1e-9 relative as everywhere.
"""
import numpy as np
import pytest

from dynamont_amd import Aligner, synth
from oracle.pyoracle import Oracle

pytestmark = pytest.mark.gpu


def _check(al, orc, reads, tag):
    t = None
    with al.batch([r.signal for r in reads], [r.sequence for r in reads]) as b:
        b.train()
        tr = b.fetch_train()
        t = b.timing()
    ok = 0
    for i, r in enumerate(reads):
        where = (tag, i, len(r.signal), len(r.sequence))
        try:
            want = orc.train(r.signal, r.sequence)
        except RuntimeError as e:
            assert tr.error(i) == str(e), where
            continue
        assert tr.status[i] == 0, (where, tr.error(i))
        assert abs(tr.Z[i] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"])), where
        # the oracle's noise floor: its log-space values are ~Z, spaced eps |Z| apart, and a posterior is the exp of a
        # sum of T of them (a far-out sample makes |Z| 1e6 .. 1e9 and the reference's statistics correspondingly coarse)
        tol = max(1e-7, 256 * 2.2e-16 * abs(want["Z"]))
        assert abs(tr.transitions[3 * i] - want["m1"]) <= tol, where
        assert abs(tr.transitions[3 * i + 2] - want["e2"]) <= tol, where
        code, m, sdev = tr.sparse(i)
        a = int(tr.em_offsets[i])
        got_w = np.zeros(len(want["weight"]))
        got_w[code] = tr.em_weight[a:a + len(code)]
        # (a k-mer the oracle gives weight 1e-300 and the product 0, or the other way round, is not a difference)
        assert np.abs(got_w - want["weight"]).max() <= tol * max(1.0, want["weight"].max()), where
        # every sample carries weight 1 -- in exact arithmetic; the oracle itself drifts by ~T ulp(Z) per row once a
        # far-out sample has made |Z| huge, so the sum is held to S only for reads whose Z is of ordinary size
        assert abs(got_w.sum() - want["weight"].sum()) <= tol * len(r.signal), where
        if abs(want["Z"]) < 1e6:
            assert abs(got_w.sum() - len(r.signal)) <= 1e-8 * len(r.signal), where
        # mean and stdev of k-mers that carry real weight (a k-mer holding 1e-9 of a sample has no stable mean)
        heavy = code[want["weight"][code] > 1e-3]
        dense_m, dense_s = np.zeros(len(got_w)), np.zeros(len(got_w))
        dense_m[code], dense_s[code] = m, sdev
        assert np.abs(dense_m[heavy] - want["mean"][heavy]).max() <= 10 * tol, where
        assert np.abs(dense_s[heavy] - want["stdev"][heavy]).max() <= 100 * tol, where
        ok += 1
    return ok, t


def _variants(base, rng, sd_typ):
    out = []
    for r in base:
        s = r.signal
        spiky = s.copy()
        spiky[rng.integers(0, len(s), size=len(s) // 37)] += 60 * sd_typ      # every cell of those rows below e^-300
        out.append(synth.SynthRead(spiky, r.sequence))
        far = s.copy()
        far[len(s) // 3] = 300.0                                                # ~2 000 sd away: exponent shift of 2^-2.9e6
        far[len(s) // 2] = -5e3                                                 # 2^-8e8
        out.append(synth.SynthRead(far, r.sequence))
        out.append(synth.SynthRead(np.ascontiguousarray(s[rng.permutation(len(s))]), r.sequence))   # fits nothing
        out.append(synth.SynthRead(np.ascontiguousarray(s[::-1]), r.sequence))
        flat = np.full(len(s), float(np.median(s)))                             # no information at all
        out.append(synth.SynthRead(flat, r.sequence))
        # all the k-mers of the first half squeezed into a tenth of the samples: the path hugs the band edge
        cut = len(s) // 2
        squeezed = np.concatenate([s[:cut:5], np.repeat(s[cut:], 2)[: len(s) - len(s[:cut:5])]])
        out.append(synth.SynthRead(np.ascontiguousarray(squeezed), r.sequence))
    return out


@pytest.mark.parametrize("pore,nb", [("rna004", (250, 420)), ("dna_r9", (150, 400)), ("rna002", (100, 300))])
def test_train_linear_domain_under_stress(models, pore, nb):
    from test_gpu_parity import model_for
    path = model_for(models, pore)
    pid, rna, k = synth.PORES[pore]
    _, mean, sd = synth.read_model_file(path)
    rng = np.random.default_rng(950)
    base = synth.make_reads(77, 5, pore, mean, sd, nb)
    al = Aligner(path, pore, device=0)
    orc = Oracle(path, pid)
    n_ok, t = _check(al, orc, base + _variants(base, rng, float(np.median(sd))), pore)
    assert n_ok >= 20
    al.close()


def test_train_sample_beyond_the_linear_guard_is_redone_in_the_log_domain(models):
    """One sample 7e4 model standard deviations out (log density -2e9; the guard of the linear sweeps is -1e9): the linear sweeps declare the row dead, the weights no longer sum
    to T - 1, the read is redone in the log domain and answers like the oracle (which still trains on it)."""
    path = models["syn9"]
    _, mean, sd = synth.read_model_file(path)
    base = synth.make_reads(78, 3, "rna004", mean, sd, (200, 300))
    reads = []
    for j, r in enumerate(base):
        s = r.signal.copy()
        s[len(s) // (j + 2)] = 1e4
        reads.append(synth.SynthRead(s, r.sequence))
    al = Aligner(path, "rna004", device=0)
    n_ok, t = _check(al, Oracle(path, 1), base + reads, "far")
    assert t["reads_log_redo"] == 3
    al.close()


def test_train_log_domain_redo_gives_the_same_statistics(models, monkeypatch):
    """DYN_LIN_PARK=-700 leaves the linear sweeps e^230 of range: ordinary 20 k-sample reads lose posterior mass, are
    caught by their weight sum and redone. Results must not depend on which way a read went."""
    path = models["syn9"]
    _, mean, sd = synth.read_model_file(path)
    cfg = synth.CONFIGS["cfg5"]
    reads = synth.make_reads(cfg["seed"], 48, cfg["pore"], mean, sd, cfg["n_bases"])
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    al = Aligner(path, "rna004", device=0)
    with al.batch(sigs, seqs) as b:
        b.train()
        want, t0 = b.fetch_train(), b.timing()
    assert t0["reads_log_redo"] == 0 and (want.status == 0).all()
    monkeypatch.setenv("DYN_LIN_PARK", "-700")
    with al.batch(sigs, seqs) as b:
        b.train()
        got, t1 = b.fetch_train(), b.timing()
    monkeypatch.delenv("DYN_LIN_PARK")
    assert t1["reads_log_redo"] >= 24, t1
    assert (got.status == 0).all()
    assert np.allclose(got.Z, want.Z, rtol=1e-11)
    assert np.array_equal(got.em_offsets, want.em_offsets) and np.array_equal(got.em_count, want.em_count)
    assert np.allclose(got.em_weight, want.em_weight, rtol=1e-7, atol=1e-12)
    assert np.allclose(got.trans_counts, want.trans_counts, rtol=1e-7, atol=1e-9)
    al.close()
