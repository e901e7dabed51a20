"""train()'s forward sweep is the posterior chain (nt_kernels.hip forward_train_chain): posteriors propagated with the
stay probability exp(e2 + e(t,n) + bE(t,n) - bE(t-1,n)) of every cell, one exponential per cell, everything in [0, 1].
These tests push on what could break it: samples far outside every k-mer's density, signal that fits nothing, stalls and
squeezed dwell times (the path at the band's edge), basecalls that disagree with the signal, unnormalised transition
weights (dna_r9: m1 = e2 = 1). The checker is the oracle (reference arithmetic).

Tolerances: the oracle's own weights carry ~sqrt(T) x 7e-12 x |log Z| / 4e4 of rounding noise (80-bit adjudication:
tests/extended_precision_train.py), and so does the chain, which works from log-space backward values; weights compare
at 1e-7 relative, Z at 1e-9 relative as everywhere.
"""
import numpy as np
import pytest

from dynamont_amd import Aligner, synth
from oracle.pyoracle import Oracle

pytestmark = pytest.mark.gpu


def _check(al, orc, reads, tag):
    tr = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
    ok = 0
    for i, r in enumerate(reads):
        where = (tag, i, len(r.signal), len(r.sequence))
        try:
            want = orc.train(r.signal, r.sequence)
        except RuntimeError as e:
            if str(e) == "Training failed: alignment scores do not match" and tr.status[i] == 0:
                # the reference's check compares two ROUNDINGS of Z (forward, backward); with |Z| ~ 1e12 (a sample 1e6
                # standard deviations out) their noise exceeds its threshold and it gives up. The chain has one Z and
                # conserves mass whatever the magnitudes: it trains the read. Stated deviation.
                assert abs(tr.Z[i]) > 1e11, where
                a, c = int(tr.em_offsets[i]), int(tr.em_count[i])
                assert abs(tr.em_weight[a:a + c].sum() - len(r.signal)) <= 1e-9 * len(r.signal), where
                continue
            assert tr.error(i) == str(e), where
            continue
        assert tr.status[i] == 0, (where, tr.error(i))
        assert abs(tr.Z[i] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"])), where
        # the oracle's noise floor: its log-space values are ~Z, spaced eps |Z| apart, and a posterior is the exp of a
        # sum of T of them (a far-out sample makes |Z| 1e6 .. 1e9 and the reference's statistics correspondingly coarse)
        # (measured on the -5e3 sample below: the oracle's weights sum to 2502.94 for 2503 samples, 2.4e-5 short, the
        # chain's to 2503.000000000001)
        tol = max(1e-7, 1024 * 2.2e-16 * abs(want["Z"]))
        assert abs(tr.transitions[3 * i] - want["m1"]) <= tol, where
        assert abs(tr.transitions[3 * i + 2] - want["e2"]) <= tol, where
        code, m, sdev = tr.sparse(i)
        a = int(tr.em_offsets[i])
        got_w = np.zeros(len(want["weight"]))
        got_w[code] = tr.em_weight[a:a + len(code)]
        # (a k-mer the oracle gives weight 1e-300 and the product 0, or the other way round, is not a difference)
        assert np.abs(got_w - want["weight"]).max() <= tol * max(1.0, want["weight"].max()), where
        # every sample carries weight 1 -- by construction in the chain; the oracle itself drifts by ~T ulp(Z) per row
        # once a far-out sample has made |Z| huge
        assert abs(got_w.sum() - len(r.signal)) <= 1e-9 * len(r.signal), where
        assert abs(got_w.sum() - want["weight"].sum()) <= tol * len(r.signal), where
        # mean and stdev of k-mers that carry real weight (a k-mer holding 1e-9 of a sample has no stable mean)
        heavy = code[want["weight"][code] > 1e-3]
        dense_m, dense_s = np.zeros(len(got_w)), np.zeros(len(got_w))
        dense_m[code], dense_s[code] = m, sdev
        scale = max(1.0, float(np.abs(r.signal).max()))  # a k-mer that owns the far-out sample has a mean of that size
        assert np.abs(dense_m[heavy] - want["mean"][heavy]).max() <= 10 * tol * scale, where
        assert np.abs(dense_s[heavy] - want["stdev"][heavy]).max() <= 100 * tol * scale, where
        ok += 1
    return ok


def _variants(base, rng, sd_typ):
    out = []
    for r in base:
        s = r.signal
        spiky = s.copy()
        spiky[rng.integers(0, len(s), size=len(s) // 37)] += 60 * sd_typ      # every cell of those rows below e^-300
        out.append(synth.SynthRead(spiky, r.sequence))
        far = s.copy()
        far[len(s) // 3] = 300.0                                                # ~2 000 sd away
        far[len(s) // 2] = -5e3                                                 # log density -5e8
        out.append(synth.SynthRead(far, r.sequence))
        out.append(synth.SynthRead(np.ascontiguousarray(s[rng.permutation(len(s))]), r.sequence))   # fits nothing
        out.append(synth.SynthRead(np.ascontiguousarray(s[::-1]), r.sequence))
        flat = np.full(len(s), float(np.median(s)))                             # no information at all
        out.append(synth.SynthRead(flat, r.sequence))
        # all the k-mers of the first half squeezed into a tenth of the samples: the path hugs the band edge
        cut = len(s) // 2
        squeezed = np.concatenate([s[:cut:5], np.repeat(s[cut:], 2)[: len(s) - len(s[:cut:5])]])
        out.append(synth.SynthRead(np.ascontiguousarray(squeezed), r.sequence))
        # what a basecaller does: 5 % substitutions, 3 % insertions / deletions against the signal's true sequence
        seq = list(r.sequence)
        called = seq[:9]
        for ch in seq[9:]:
            u = rng.random()
            if u < 0.015:
                continue
            if u < 0.03:
                called.append("ACGT"[rng.integers(0, 4)])
            called.append("ACGT"[rng.integers(0, 4)] if rng.random() < 0.05 else ch)
        out.append(synth.SynthRead(s.copy(), "".join(called)))
    return out


@pytest.mark.parametrize("pore,nb", [("rna004", (250, 420)), ("dna_r9", (150, 400)), ("rna002", (100, 300))])
def test_train_posterior_chain_under_stress(models, pore, nb):
    from test_gpu_parity import model_for
    path = model_for(models, pore)
    pid, rna, k = synth.PORES[pore]
    _, mean, sd = synth.read_model_file(path)
    rng = np.random.default_rng(950)
    base = synth.make_reads(77, 5, pore, mean, sd, nb)
    al = Aligner(path, pore, device=0)
    orc = Oracle(path, pid)
    n_ok = _check(al, orc, base + _variants(base, rng, float(np.median(sd))), pore)
    assert n_ok >= 25
    al.close()


def test_train_sample_at_the_edge_of_double_precision(models):
    """Samples 7e4 and 1e6 model standard deviations out (log densities -2e9 and -9e11): log-space values that large
    leave the reference's posteriors 1e-7 .. 1e-4 of resolution. At -2e9 the reads train like the oracle within that; at
    -9e11 the reference fails its own Z check on rounding noise and the chain still trains (see _check)."""
    path = models["syn9"]
    _, mean, sd = synth.read_model_file(path)
    base = synth.make_reads(78, 3, "rna004", mean, sd, (200, 300))
    reads = []
    for j, r in enumerate(base):
        s = r.signal.copy()
        s[len(s) // (j + 2)] = 1e4 if j < 2 else 2e5
        reads.append(synth.SynthRead(s, r.sequence))
    al = Aligner(path, "rna004", device=0)
    assert _check(al, Oracle(path, 1), base + reads, "far") == 5
    al.close()


def test_train_reference_zcheck_refuses_what_the_reference_refuses(models):
    """dyn_aligner_set_train_zcheck: with the reference's own rule on top (|Zf - Zb| / size > 1e-8, NT_aligner_api.cpp:
    619-625), the read with a sample 1e6 standard deviations out is refused with the reference's message -- as the oracle
    refuses it -- and ordinary reads train exactly as without the switch."""
    path = models["syn9"]
    _, mean, sd = synth.read_model_file(path)
    base = synth.make_reads(78, 3, "rna004", mean, sd, (200, 300))
    far = base[2].signal.copy()
    far[len(far) // 4] = 2e5
    reads = base + [synth.SynthRead(far, base[2].sequence)]
    orc = Oracle(path, 1)
    with pytest.raises(RuntimeError, match="Training failed: alignment scores do not match"):
        orc.train(far, base[2].sequence)
    al = Aligner(path, "rna004", device=0)
    plain = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
    assert plain.status.tolist() == [0, 0, 0, 0]            # the chain trains all four (stated deviation)
    al.set_train_zcheck(True)
    strict = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
    assert strict.status.tolist()[:3] == [0, 0, 0] and strict.status[3] != 0
    assert strict.error(3) == "Training failed: alignment scores do not match"
    for i in range(3):
        assert strict.Z[i] == plain.Z[i]
        a, c = int(plain.em_offsets[i]), int(plain.em_count[i])
        assert np.array_equal(strict.em_weight[a:a + c], plain.em_weight[a:a + c])
    al.close()


def test_train_long_dna_reads(models):
    """cfg3-shaped reads through train(): 8 DNA r10.4.1 reads of 40 k - 100 k samples (the lattice of one read spans
    hundreds of pages, the band moves every ~12 rows); every read conserves its mass, two are compared with the oracle."""
    import multiprocessing as mp
    from test_gpu_parity import model_for, _orc_init, _orc_train
    path = model_for(models, "dna_r10_400bps")
    _, mean, sd = synth.read_model_file(path)
    reads = synth.make_reads(303, 8, "dna_r10_400bps", mean, sd, (3200, 8000))
    al = Aligner(path, "dna_r10_400bps", device=0)
    tr = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
    assert (tr.status == 0).all()
    assert max(len(r.signal) for r in reads) > 80000
    for i, r in enumerate(reads):
        a, c = int(tr.em_offsets[i]), int(tr.em_count[i])
        assert abs(tr.em_weight[a:a + c].sum() - len(r.signal)) <= 1e-9 * len(r.signal), i
    picks = [int(np.argmax([len(r.signal) for r in reads])), 0]
    with mp.get_context("fork").Pool(2, initializer=_orc_init, initargs=(path, synth.PORES["dna_r10_400bps"][0])) as pool:
        want = pool.map(_orc_train, [(reads[i].signal, reads[i].sequence) for i in picks], chunksize=1)
    for i, (Z, m1, e2, w, s1, s2) in zip(picks, want):
        assert abs(tr.Z[i] - Z) <= 1e-9 * abs(Z)
        assert abs(tr.transitions[3 * i] - m1) <= 1e-8 and abs(tr.transitions[3 * i + 2] - e2) <= 1e-8
        code, m, s = tr.sparse(i)
        touched = np.nonzero(w > 0)[0]
        assert np.array_equal(code, touched)
        a = int(tr.em_offsets[i])
        assert np.allclose(tr.em_weight[a:a + len(code)], w[touched], rtol=3e-7, atol=1e-12)
    al.close()
