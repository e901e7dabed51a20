"""GPU (-m gpu): the HIP path, called through the C ABI, against
  * the golden vectors generated from the compiled reference (G1-G5, G7),
  * the CPU oracle on the same seeded inputs,
  * size-independent properties at BASELINE.json's full sizes.
Bar (BASELINE.json north_star): start/end/basepos/state bit-exact, posterior within 1e-4
(observed ~2e-8: path posteriors are stored as fp32 log-probabilities), Z within 1e-9 relative."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_matches_golden, golden, model_for
from dynamont_amd import Aligner, synth
from oracle.pyoracle import Oracle

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("native_lib", "oracle_built")]

PROB_TOL = 1e-4   # north_star tolerance
PROB_TIGHT = 1e-6  # what the fp32 LP storage actually delivers, guarded so regressions show


@pytest.fixture(scope="module")
def al5(models):
    return Aligner(models["syn5"], "rna002", device=0)


@pytest.fixture(scope="module")
def al9(models):
    return Aligner(models["syn9"], "rna004", device=0)


def test_native_library_is_what_runs(native_lib):
    # the extension must be the in-tree .so, loaded in this process
    maps = open("/proc/self/maps").read()
    assert "dynamont_amd/libdynamont_mi.so" in maps


def test_g1_cfg1(models, al5, tmp_path):
    g = golden("g1_cfg1.npz")
    res = al5.align(g["syn_signal"], str(g["syn_sequence"]), True)
    assert_matches_golden(res, g, "syn_", PROB_TIGHT)
    assert res["signal_positions"].dtype == np.uint64 and res["sequence_positions"].dtype == np.uint64
    assert res["polishes"] == [""] * len(res["states"])
    real = tmp_path / "rna002_5mer.model"
    with open(real, "w") as w:
        w.write("kmer\tlevel_mean\tlevel_stdv\n")
        for n, m, s in zip(synth.kmer_strings(5), g["real_model_mean"], g["real_model_stdev"]):
            w.write(f"{n}\t{float(m)!r}\t{float(s)!r}\n")
    res = Aligner(str(real), "rna002", device=0).align(g["real_signal"], str(g["real_sequence"]), True)
    assert_matches_golden(res, g, "real_", PROB_TIGHT)
    # calc_probabilities=False: Z only, no segments (NT_aligner_api.cpp:293-295)
    z = al5.align(g["syn_signal"], str(g["syn_sequence"]))
    assert len(z["probabilities"]) == 0 and abs(z["Z"] - float(g["syn_Z"])) <= 1e-9 * abs(float(g["syn_Z"]))


def test_g2_rna004_batch(models, al9):
    g = golden("g2_rna004.npz")
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(int(g["seed"]), int(g["n_reads"]), "rna004", mean, sd, (200, 2000))
    res = al9.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    assert (res.status == 0).all()
    for i in range(len(reads)):
        assert_matches_golden(res.read(i), g, f"r{i}_", PROB_TIGHT)


def test_g3_short_reads_bandwidth_clamp_and_bands(models):
    g = golden("g3_short.npz")
    groups = {}
    for i in range(int(g["n_cases"])):
        p = f"c{i}_"
        band = int(g[p + "band"]) if p + "band" in g else 400
        groups.setdefault((str(g[p + "pore"]), band), []).append(p)
    for (pore, band), ps in groups.items():
        al = Aligner(model_for(models, pore), pore, band=band, device=0)
        res = al.align_batch([g[p + "signal"] for p in ps], [str(g[p + "sequence"]) for p in ps], True)
        for j, p in enumerate(ps):
            assert_matches_golden(res.read(j), g, p, PROB_TIGHT)


def test_g4_dna_long_reads(models):
    g = golden("g4_dna_long.npz")
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(int(g["seed"]), int(g["n_reads"]), "dna_r10_400bps", mean, sd, (7000, 8000))
    al = Aligner(models["syn9"], "dna_r10_400bps", device=0)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    for i in range(len(reads)):
        assert_matches_golden(res.read(i), g, f"r{i}_", PROB_TIGHT)


def test_g5_failures_are_isolated_per_read(al5):
    d = json.load(open(os.path.join(GOLDEN, "g5_failures.json")))
    cases = [c for c in d["align"] if c["name"] != "calc_false"]
    res = al5.align_batch([np.array(c["signal"]) for c in cases], [c["sequence"] for c in cases], True)
    for i, c in enumerate(cases):
        if c["ok"]:
            r = res.read(i)
            assert abs(r["Z"] - c["Z"]) <= 1e-9 * max(1.0, abs(c["Z"])) and len(r["states"]) == c["nseg"]
        else:
            assert res.error(i) == c["message"]
            with pytest.raises(RuntimeError) as e:
                res.read(i)
            assert str(e.value) == c["message"]
    # single-read surface raises like the reference's Aligner.align
    bad = next(c for c in cases if c["name"] == "base_N")
    with pytest.raises(RuntimeError, match="Invalid nucleotide: N"):
        al5.align(np.array(bad["signal"]), bad["sequence"], True)


@pytest.mark.parametrize("pore,nb", [("rna002", (30, 300)), ("rna004", (100, 700)), ("dna_r9", (20, 250)),
                                     ("dna_r10_260bps", (300, 900)), ("dna_r10_400bps", (450, 600))])
def test_random_reads_against_oracle(models, pore, nb):
    path = model_for(models, pore)
    _, mean, sd = synth.read_model_file(path)
    reads = synth.make_reads(4242, 12, pore, mean, sd, nb)
    # a noise read and a read with an outlier burst as well
    rng = np.random.default_rng(9)
    reads[1].signal[:] = rng.standard_normal(len(reads[1].signal))
    reads[2].signal[50:60] += 25.0
    al = Aligner(path, pore, device=0)
    orc = Oracle(path, synth.PORES[pore][0])
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    for i, r in enumerate(reads):
        try:
            want = orc.align(r.signal, r.sequence, True)
        except RuntimeError as e:
            assert res.error(i) == str(e)
            continue
        got = res.read(i)
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"])
        assert np.array_equal(got["signal_positions"], want["signal_positions"])
        assert got["states"] == want["states"]
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT
        assert abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))


@pytest.mark.parametrize("pore,nb", [("rna002", (30, 300)), ("rna004", (100, 700)), ("dna_r9", (20, 250))])
def test_z_only_alignment_against_oracle(models, pore, nb):
    """align(calc_probabilities=False), the reference's default call: Z and the check that both sweeps agree on it, no
    segments. The sweeps run the cheap arithmetic here (no decision hangs on them): Z to 1e-9 relative like everywhere,
    error texts like the reference, on ordinary, noise and outlier reads."""
    path = model_for(models, pore)
    _, mean, sd = synth.read_model_file(path)
    reads = synth.make_reads(777, 16, pore, mean, sd, nb)
    rng = np.random.default_rng(10)
    reads[1].signal[:] = rng.standard_normal(len(reads[1].signal))
    reads[2].signal[40:50] += 25.0
    reads[3].signal[len(reads[3].signal) // 2] = 3e3
    reads[4] = synth.SynthRead(reads[4].signal[: 2 * (len(reads[4].sequence) - synth.PORES[pore][2] + 1) - 1], reads[4].sequence)  # too short
    al = Aligner(path, pore, device=0)
    orc = Oracle(path, synth.PORES[pore][0])
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], False)
    n_ok = 0
    for i, r in enumerate(reads):
        try:
            want = orc.align(r.signal, r.sequence, False)
        except RuntimeError as e:
            assert res.error(i) == str(e), i
            continue
        assert res.status[i] == 0, (i, res.error(i))
        assert abs(res.Z[i] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"])), (i, res.Z[i], want["Z"])
        assert res.read(i)["signal_positions"].size == 0
        n_ok += 1
    assert n_ok >= 14
    al.close()


@pytest.mark.parametrize("pore,nb", [("rna002", (60, 400)), ("rna004", (150, 700)), ("dna_r10_400bps", (300, 700))])
def test_models_with_a_stdev_per_kmer(models, tmp_path, pore, nb):
    """The seeded synthetic models give every k-mer the same stdev; real ones do not (rna004: 0.05 .. 0.5). A model with
    log-normal stdevs over a factor of 25 -- different 1/stdev, log stdev and density constant in every cell of a
    lane -- through align(calc=true) and train() against the oracle."""
    pid, rna, k = synth.PORES[pore]
    rng = np.random.default_rng(2025 + k)
    mean = rng.standard_normal(4 ** k)
    sd = np.clip(0.15 * np.exp(0.6 * rng.standard_normal(4 ** k)), 0.03, 0.75)
    path = synth.write_model_values(str(tmp_path / "varsd.model"), k, mean, sd)
    reads = synth.make_reads(77 + k, 10, pore, mean, sd, nb)
    al = Aligner(path, pore, device=0)
    orc = Oracle(path, pid)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    tr = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
    for i, r in enumerate(reads):
        want, got = orc.align(r.signal, r.sequence, True), res.read(i)
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"]), i
        assert np.array_equal(got["signal_positions"], want["signal_positions"]), i
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT, i
        assert abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"])), i
        wt = orc.train(r.signal, r.sequence)
        assert tr.status[i] == 0 and abs(tr.Z[i] - wt["Z"]) <= 1e-9 * max(1.0, abs(wt["Z"])), i
        assert abs(tr.transitions[3 * i] - wt["m1"]) <= 1e-8 and abs(tr.transitions[3 * i + 2] - wt["e2"]) <= 1e-8, i
        code, m, sdev = tr.sparse(i)
        touched = np.nonzero(wt["weight"] > 0)[0]
        assert np.array_equal(code, touched), i
        a = int(tr.em_offsets[i])
        assert np.allclose(tr.em_weight[a:a + len(code)], wt["weight"][touched], rtol=1e-7, atol=1e-12), i
        heavy = wt["weight"][touched] > 1e-3
        assert np.abs(m[heavy] - wt["mean"][touched][heavy]).max() <= 1e-7, i
        assert np.abs(sdev[heavy] - wt["stdev"][touched][heavy]).max() <= 1e-6, i
    al.close()


@pytest.mark.parametrize("pore,band,dwell", [("rna004", 400, 0.5), ("dna_r9", 400, 2.5), ("rna004", 100, 0.5),
                                             ("dna_r10_260bps", 446, 3.0), ("rna002", 30, 1.0)])
def test_dense_reads_moving_window(models, pore, band, dwell):
    """Reads longer than the band with 2-3 samples per base: the band window moves in (almost)
    every other row, so the slot hand-overs of the kernels' window-move blocks (leaving column ->
    column lo+P, entering column gets its k-mer parameters one row ahead) run back to back."""
    path = model_for(models, pore)
    _, mean, sd = synth.read_model_file(path)
    reads = synth.make_reads(99, 6, pore, mean, sd, (band + 30, band + 500), dwell=dwell)
    ratio = [(len(r.sequence) - synth.PORES[pore][2] + 2) / (len(r.signal) + 1) for r in reads]
    assert max(ratio) > 0.3
    al = Aligner(path, pore, band=band, device=0)
    orc = Oracle(path, synth.PORES[pore][0], band)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    checked = 0
    for i, r in enumerate(reads):
        try:
            want = orc.align(r.signal, r.sequence, True)
        except RuntimeError as e:
            assert res.error(i) == str(e)
            continue
        got = res.read(i)
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"])
        assert np.array_equal(got["signal_positions"], want["signal_positions"])
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT
        assert abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))
        checked += 1
    assert checked >= 3


@pytest.mark.parametrize("pore,band", [("dna_r9", 4), ("dna_r9", 7), ("rna002", 10), ("dna_r9", 31), ("rna004", 60),
                                       ("dna_r10_260bps", 199), ("rna002", 446)])
def test_small_bands_and_tiny_reads_fuzz(models, pore, band):
    """Property check of the band logic: many short reads of every length class (sequence barely
    longer than k, band wider than / equal to / much narrower than the read, 2-4 samples per base up
    to the default dwell) against the oracle, one batch per band."""
    path = model_for(models, pore)
    k = synth.PORES[pore][2]
    _, mean, sd = synth.read_model_file(path)
    rng = np.random.default_rng(band)
    reads = []
    for i in range(48):
        nb = int(rng.choice([k, k + 1, k + 2, k + 5, band // 2 + k, band + k - 1, band + k, band + k + 1, 2 * band + k + 3, 3 * band + 40]))
        dwell = float(rng.choice([0.5, 2.0, 3.5, 10.0]))
        reads += synth.make_reads(1000 * band + i, 1, pore, mean, sd, max(nb, k), dwell=dwell)
    al = Aligner(path, pore, band=band, device=0)
    orc = Oracle(path, synth.PORES[pore][0], band)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    zs = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], False)
    tr = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
    ok = 0
    for i, r in enumerate(reads):
        tag = (i, len(r.sequence), len(r.signal))
        try:
            want = orc.align(r.signal, r.sequence, True)
        except RuntimeError as e:
            assert res.error(i) == str(e), tag
            assert zs.error(i) == str(e), tag
            continue
        assert res.status[i] == 0, (tag, res.error(i))
        got = res.read(i)
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"]), tag
        assert np.array_equal(got["signal_positions"], want["signal_positions"]), tag
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT
        assert abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))
        assert zs.status[i] == 0 and abs(zs.Z[i] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"])), tag
        # the training sweeps (backward with the stored stay exponent, posterior chain) through the same band logic
        wt = orc.train(r.signal, r.sequence)
        assert tr.status[i] == 0, (tag, tr.error(i))
        assert abs(tr.Z[i] - wt["Z"]) <= 1e-9 * max(1.0, abs(wt["Z"])), tag
        assert abs(tr.transitions[3 * i] - wt["m1"]) <= 1e-8 and abs(tr.transitions[3 * i + 2] - wt["e2"]) <= 1e-8, tag
        code, m, sdev = tr.sparse(i)
        touched = np.nonzero(wt["weight"] > 0)[0]
        assert np.array_equal(code, touched), tag
        a = int(tr.em_offsets[i])
        assert np.allclose(tr.em_weight[a:a + len(code)], wt["weight"][touched], rtol=1e-7, atol=1e-12), tag
        ok += 1
    assert ok >= 24


def test_order_and_chunking_invariance(models, al9):
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(77, 24, "rna004", mean, sd, (150, 900))
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    with al9.batch(sigs, seqs) as b:
        b.align(True)
        base = b.fetch()
        tm = b.timing()
        assert tm["launches"] == 1 and tm["lp_inplace"] == 0 and tm["n_static"] == len(reads)
    perm = np.random.default_rng(3).permutation(len(reads))
    shuf = al9.align_batch([sigs[i] for i in perm], [seqs[i] for i in perm], True)
    small = Aligner(models["syn9"], "rna004", device=0)
    small.set_mem_budget(80 << 20)  # a pool for 2-4 reads at a time (each read needs 20-30 MB): the rest queue for pages
    with small.batch(sigs, seqs) as b:
        b.align(True)
        chunked = b.fetch()
        # a pool that cannot serve every wave keeps the posteriors in place (8 instead of 12 B per slot);
        # still one launch: reads past the first round take their pages from the free list on the device
        tm = b.timing()
        assert tm["launches"] == 1 and tm["lp_inplace"] == 1 and 0 < tm["n_static"] < len(reads)
    os.environ["DYN_FORCE_LAYOUT"] = "separate"   # the same starved pool with the separate LPE layout
    try:
        with small.batch(sigs, seqs) as b:
            b.align(True)
            starved_sep = b.fetch()
            tm = b.timing()
            assert tm["lp_inplace"] == 0 and 0 < tm["n_static"] < len(reads) and tm["wave_wait_share"] > 0
    finally:
        del os.environ["DYN_FORCE_LAYOUT"]
    for i in range(len(reads)):
        a, s2 = base.read(i), starved_sep.read(i)
        assert np.array_equal(a["signal_positions"], s2["signal_positions"]) and a["Z"] == s2["Z"]
        assert np.array_equal(a["probabilities"], s2["probabilities"])  # same layout, same arithmetic: bitwise
    for j, i in enumerate(perm):
        a, s, c = base.read(int(i)), shuf.read(j), chunked.read(int(i))
        assert np.array_equal(a["signal_positions"], s["signal_positions"])
        assert np.array_equal(a["probabilities"], s["probabilities"])  # bitwise: deterministic kernels
        assert a["Z"] == s["Z"]
        # the other posterior layout: same path, same Z; segment-start posteriors are rebuilt from
        # LPE + backward values there and read back as float LPM here
        assert np.array_equal(a["signal_positions"], c["signal_positions"]) and a["Z"] == c["Z"]
        assert np.abs(a["probabilities"] - c["probabilities"]).max() <= PROB_TIGHT


def test_a_read_whose_lattice_exceeds_the_budget_fails_alone(models):
    """A read whose lattice alone does not fit the handle's memory budget gets its own status (the reference would raise
    std::bad_alloc for that read only and report its line, segment.py:172-176); every other read of the batch is computed and
    equals the oracle -- synchronous batch, train() and the asynchronous pipeline; the Z-only call keeps no lattice and computes
    that read too."""
    path = models["syn9"]
    _, mean, sd = synth.read_model_file(path)
    small_reads = synth.make_reads(811, 6, "rna004", mean, sd, (150, 300))
    big = synth.make_reads(812, 1, "rna004", mean, sd, 2000)[0]       # ~20 k samples: ~70-110 MB of lattice
    reads = small_reads[:3] + [big] + small_reads[3:]
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    al = Aligner(path, "rna004", device=0)
    al.set_mem_budget(48 << 20)
    orc = Oracle(path, synth.PORES["rna004"][0])
    msg = "Read too large for the device memory budget"

    def check(res, probs=True):
        for i, r in enumerate(reads):
            if i == 3 and probs:  # (the Z-only call stores no lattice: it computes this read as well)
                assert res.status[i] == 8 and res.error(i) == msg
                continue
            assert res.status[i] == 0, (i, res.error(i))
            want = orc.align(r.signal, r.sequence, probs)
            assert abs(res.Z[i] - want["Z"]) <= 1e-9 * abs(want["Z"])
            if probs:
                got = res.read(i)
                assert np.array_equal(got["sequence_positions"], want["sequence_positions"])
                assert np.array_equal(got["signal_positions"], want["signal_positions"])
                assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT

    check(al.align_batch(sigs, seqs, True))
    check(al.align_batch(sigs, seqs, False), probs=False)
    tr = al.train_batch(sigs, seqs)
    assert tr.status[3] == 8 and tr.error(3) == msg and all(tr.status[i] == 0 for i in range(len(reads)) if i != 3)
    with al.align_async(*synth.pack_reads(reads), True) as t:
        check(t.wait())
    al.close()


@pytest.mark.parametrize("pore", ["rna004", "dna_r9"])
def test_in_place_posterior_layout_against_oracle(models, pore):
    """The layout of footprint-limited batches, forced with a small memory budget."""
    path = model_for(models, pore)
    _, mean, sd = synth.read_model_file(path)
    reads = synth.make_reads(505, 10, pore, mean, sd, (200, 700))
    al = Aligner(path, pore, device=0)
    al.set_mem_budget(60 << 20)
    orc = Oracle(path, synth.PORES[pore][0])
    with al.batch([r.signal for r in reads], [r.sequence for r in reads]) as b:
        b.align(True)
        res = b.fetch()
        assert b.timing()["lp_inplace"] == 1
    for i, r in enumerate(reads):
        want, got = orc.align(r.signal, r.sequence, True), res.read(i)
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"])
        assert np.array_equal(got["signal_positions"], want["signal_positions"])
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT


_ORC = None


def _orc_init(model, pore_id):
    global _ORC
    _ORC = Oracle(model, pore_id)


def _orc_align(job):
    sig, seq = job
    r = _ORC.align(sig, seq, True)
    return r["Z"], r["signal_positions"], r["sequence_positions"], r["probabilities"]


def _oracle_parity(model, pore_id, reads, res, picks, procs=16):
    """Full parity of the reads `picks` against the oracle, run in a process pool (the oracle is
    single-threaded and takes ~2 s per 20 k-sample read)."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(procs, len(picks), os.cpu_count() or 1), initializer=_orc_init,
                                     initargs=(model, pore_id)) as pool:
        want = pool.map(_orc_align, [(reads[i].signal, reads[i].sequence) for i in picks], chunksize=1)
    for i, (Z, sigpos, seqpos, prob) in zip(picks, want):
        got = res.read(i)
        assert np.array_equal(got["signal_positions"], sigpos), i
        assert np.array_equal(got["sequence_positions"], seqpos), i
        assert np.abs(got["probabilities"] - prob).max() <= PROB_TIGHT, i
        assert abs(got["Z"] - Z) <= 1e-9 * max(1.0, abs(Z)), i


def _full_size_properties(reads, res, k):
    """Size-independent properties of a segmentation: one segment per k-mer, starts strictly increasing by
    at least the 2 rows a segment needs, first at 0, base positions 0..Kc-1 (+k/2), posteriors in [0, 1]."""
    assert (res.status == 0).all()
    for i, r in enumerate(reads):
        a, n = int(res.seg_offsets[i]), int(res.n_segments[i])
        assert n == len(r.sequence) - k + 1
        sp = res.signal_positions[a:a + n]
        assert sp[0] == 0 and np.all(np.diff(sp.astype(np.int64)) >= 2) and sp[-1] <= len(r.signal) - 2
        assert np.array_equal(res.sequence_positions[a:a + n], np.arange(n, dtype=np.uint64) + k // 2)
        p = res.probabilities[a:a + n]
        assert np.all((p >= 0) & (p <= 1.0 + 1e-6))
    assert np.all(np.isfinite(res.Z))


def _resident_queue_gives_the_same(al, packed, res):
    """The same batch as an asynchronous ticket -- the RESIDENT read queue (no launch of its own) -- must be the one-launch
    result bit for bit: Z bits, integer columns, probabilities. (What bench.py times is this path.)"""
    t = al.align_async(*packed, True)
    got = t.wait()
    tm = t.timing()
    assert tm["launches"] == 0 and tm["reads_ok"] == int((res.status == 0).sum())
    assert np.array_equal(got.status, res.status) and np.array_equal(got.Z.view(np.uint64), res.Z.view(np.uint64))
    assert np.array_equal(got.n_segments, res.n_segments) and np.array_equal(got.seg_offsets, res.seg_offsets)
    m = int(res.seg_offsets[-1])
    assert np.array_equal(got.signal_positions[:m], res.signal_positions[:m])
    assert np.array_equal(got.sequence_positions[:m], res.sequence_positions[:m])
    assert np.array_equal(got.probabilities[:m].view(np.uint64), res.probabilities[:m].view(np.uint64))
    t.close()
    assert al.session_stats()["aborted"] == 0


def test_cfg2_full_size_properties_and_parity_on_64_reads(models, al9):
    """BASELINE configs[1]: 1 024 RNA004 reads x ~20 k samples. Size-independent properties on
    all reads, full parity against the oracle on every 33rd read and on every read with a structural-tie decision (63 reads)."""
    _, mean, sd = synth.read_model_file(models["syn9"])
    cfg = synth.CONFIGS["cfg2"]
    reads = synth.make_reads(cfg["seed"], cfg["n_reads"], cfg["pore"], mean, sd, cfg["n_bases"])
    sig, so, sq, qo = synth.pack_reads(reads)
    with al9.batch_packed(sig, so, sq, qo) as b:
        b.align(True)
        res = b.fetch()
        tm = b.timing()
    assert tm["samples"] == int(so[-1]) and tm["reads_ok"] == len(reads)
    assert tm["launches"] == 1 and tm["lp_inplace"] == 0 and tm["n_static"] == len(reads)
    _full_size_properties(reads, res, al9.kmer_size)
    _resident_queue_gives_the_same(al9, (sig, so, sq, qo), res)
    # every 33rd read, and EVERY read of this workload in which the reference's traceback takes a decision on a
    # structural tie (neighbouring columns with the same k-mer: margin 0 in exact arithmetic, 0 .. 1e-8 in the
    # reference's floating point; tests/decision_margin.py -> tests/golden/g9_cfg2_decision_margin.json). Decisions
    # between different k-mers have a margin >= 1e-6 over all 20.4 M decisions of the workload.
    g9 = json.load(open(os.path.join(GOLDEN, "g9_cfg2_decision_margin.json")))
    assert g9["distinct_kmer_decisions"]["min"] >= 1e-9
    ties = [t["read"] for t in g9["structural_tie_reads"]]
    _oracle_parity(models["syn9"], 1, reads, res, sorted(set(list(range(0, 1024, 33)) + [1023] + ties)))


def test_cfg4_share_full_size(models, al9):
    """BASELINE configs[3]'s per-GPU share: 4 096 RNA004 reads x ~20 k samples in ONE batch -- four rounds of
    the persistent waves, every read past the first 1 024 taken off the queue on the device."""
    _, mean, sd = synth.read_model_file(models["syn9"])
    cfg = synth.CONFIGS["cfg4"]
    reads = synth.make_reads(cfg["seed"], 4096, cfg["pore"], mean, sd, cfg["n_bases"])
    sig, so, sq, qo = synth.pack_reads(reads)
    with al9.batch_packed(sig, so, sq, qo) as b:
        b.align(True)
        res = b.fetch()
        tm = b.timing()
    assert tm["reads_ok"] == 4096 and tm["launches"] == 1 and tm["lp_inplace"] == 0 and tm["n_static"] == 1024
    _full_size_properties(reads, res, al9.kmer_size)
    _resident_queue_gives_the_same(al9, (sig, so, sq, qo), res)
    # 64 reads against the oracle: 48 spread evenly over the queue (longest first: the first round, the rounds taken
    # off the queue on the device, the tail) + 16 of the reads that START with a structural tie (pad + A: the first
    # two 9-mers are equal), spread the same way
    order = np.argsort([-len(r.signal) for r in reads], kind="stable")
    picks = {int(order[j]) for j in np.linspace(0, 4095, 48).astype(int)} | {int(order[j]) for j in (1023, 1024)}
    ties = [int(i) for i in order if reads[i].sequence.startswith("A" * 10)]
    assert len(ties) > 500
    picks |= {ties[j] for j in np.linspace(0, len(ties) - 1, 16).astype(int)}
    _oracle_parity(models["syn9"], 1, reads, res, sorted(picks))


def test_cfg3_full_size(models):
    """BASELINE configs[2]: 4 096 DNA r10.4.1 reads of 10 k - 100 k samples in ONE batch: the pool cannot hold a
    lattice for every wave, so part of the waves start without pages and wait for surplus pages of others."""
    _, mean, sd = synth.read_model_file(models["syn9"])
    cfg = synth.CONFIGS["cfg3"]
    reads = synth.make_reads(cfg["seed"], cfg["n_reads"], cfg["pore"], mean, sd, cfg["n_bases"])
    sig, so, sq, qo = synth.pack_reads(reads)
    al = Aligner(models["syn9"], cfg["pore"], device=0)
    with al.batch_packed(sig, so, sq, qo) as b:
        b.align(True)
        res = b.fetch()
        tm = b.timing()
    assert tm["reads_ok"] == 4096 and tm["launches"] == 1 and tm["samples"] == int(so[-1])
    # page-limited: the pool is smaller than 1 024 lattices of the longest reads, so part of the first round runs
    # short reads (host plan) and later reads change arena size on the device
    assert tm["lp_inplace"] == 1 and 0 < tm["n_static"] <= 1024
    _full_size_properties(reads, res, al.kmer_size)
    # the same batch as a ticket of the resident queue: a PAGED session (the pool's pages shared through the free list,
    # posteriors in place), bit for bit the launch's results
    t = al.align_async(sig, so, sq, qo, True)
    got = t.wait()
    tmr = t.timing()
    assert tmr["launches"] == 0 and tmr["lp_inplace"] == 1 and tmr["reads_ok"] == 4096
    m = int(res.seg_offsets[-1])
    assert np.array_equal(got.status, res.status) and np.array_equal(got.Z.view(np.uint64), res.Z.view(np.uint64))
    assert np.array_equal(got.signal_positions[:m], res.signal_positions[:m])
    assert np.array_equal(got.probabilities[:m].view(np.uint64), res.probabilities[:m].view(np.uint64))
    t.close()
    assert al.session_stats()["aborted"] == 0
    order = np.argsort([-len(r.signal) for r in reads], kind="stable")
    # 64 reads against the oracle, spread evenly over the length order (the longest and the shortest included; up to
    # ~12 s and 2.6 GB of oracle per read: 8 processes), plus every read that starts with a homopolymer of k+1 bases
    picks = {int(order[j]) for j in np.linspace(0, 4095, 64).astype(int)}
    picks |= {i for i, r in enumerate(reads) if len(set(r.sequence[:10])) == 1}
    _oracle_parity(models["syn9"], synth.PORES[cfg["pore"]][0], reads, res, sorted(picks), procs=8)
    al.close()


def test_g7_train_against_reference_golden(models):
    g = golden("g7_train.npz")
    for i in range(int(g["n_cases"])):
        p = f"t{i}_"
        pore = str(g[p + "pore"])
        al = Aligner(model_for(models, pore), pore, device=0)
        res = al.train_batch([g[p + "signal"]], [str(g[p + "sequence"])])
        assert res.status[0] == 0
        zg = float(g[p + "Z"])
        assert abs(res.Z[0] - zg) <= 1e-9 * max(1.0, abs(zg))
        assert np.abs(res.transitions[:3] - g[p + "trans"]).max() <= 1e-9
        code, mean, sd = res.sparse(0)
        assert np.array_equal(code.astype(np.int64), g[p + "codes"])
        assert np.abs(mean - g[p + "mean"]).max() <= 1e-9
        assert np.abs(sd - g[p + "stdev"]).max() <= 1e-7   # sqrt(var) amplifies cancellation in s2/w - mean^2
        # dense dict form of the reference binding
        dense = al.train(g[p + "signal"], str(g[p + "sequence"]))
        assert len(dense["emission_model"]) == al.num_kmers and dense["transition_params"]["e1"] == 1.0


def _read_with_exactly(rng, S, kc, k, mean_code, sd_code, rna):
    """A synthetic read with exactly S samples and kc k-mers (kc + k - 1 bases); dwell split as evenly as S allows."""
    digits = rng.integers(0, 4, size=kc + k - 1)
    if rna:
        digits[:min(9, len(digits))] = 0
    codes = synth._seq_codes(digits, k)
    dw = np.full(kc, S // kc)
    dw[: S - dw.sum()] += 1
    idx = np.repeat(codes, dw)
    sig = mean_code[idx] + 1.2 * sd_code[idx] * rng.standard_normal(len(idx))
    return synth.SynthRead(np.ascontiguousarray(sig, dtype=np.float64), "".join(synth.BASES[d] for d in digits))


@pytest.mark.parametrize("pore", ["rna002", "dna_r10_400bps"])
def test_row_loop_tails_at_block_and_parity_edges(models, pore):
    """The row loops run in blocks of 64 rows and two rows per iteration (ping-pong state): every combination of an
    odd / even number of rows and of a last block of 1, 2, 63, 64 rows, for align (both posterior layouts share the
    loop) and train, against the oracle."""
    path = model_for(models, pore)
    pid, rna, k = synth.PORES[pore]
    mf, sf = synth.read_model_file(path)[1:]
    mean_code, sd_code = synth.code_order_table(mf, sf, k, rna)
    rng = np.random.default_rng(64)
    reads = []
    for S in (2, 3, 4, 5, 62, 63, 64, 65, 66, 67, 126, 127, 128, 129, 130, 191, 192, 193, 194, 255, 256, 257, 258, 320, 321):
        for kc in sorted({1, max(1, S // 7), max(1, S // 2)}):
            reads.append(_read_with_exactly(rng, S, kc, k, mean_code, sd_code, rna))
    al = Aligner(path, pore, device=0)
    orc = Oracle(path, pid)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    tr = al.train_batch([r.signal for r in reads], [r.sequence for r in reads])
    checked = 0
    for i, r in enumerate(reads):
        try:
            want = orc.align(r.signal, r.sequence, True)
        except RuntimeError as e:
            assert res.error(i) == str(e), (i, len(r.signal), len(r.sequence))
            continue
        got = res.read(i)
        tag = (i, len(r.signal), len(r.sequence))
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"]), tag
        assert np.array_equal(got["signal_positions"], want["signal_positions"]), tag
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT, tag
        assert abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"])), tag
        try:
            wt = orc.train(r.signal, r.sequence)
        except RuntimeError as e:
            assert tr.error(i) == str(e), tag
            continue
        assert tr.status[i] == 0, (tag, tr.error(i))
        assert abs(tr.Z[i] - wt["Z"]) <= 1e-9 * max(1.0, abs(wt["Z"])), tag
        assert abs(tr.transitions[3 * i] - wt["m1"]) <= 1e-9 and abs(tr.transitions[3 * i + 2] - wt["e2"]) <= 1e-9, tag
        code, m, sdev = tr.sparse(i)
        touched = np.nonzero(wt["weight"] > 0)[0]
        assert np.array_equal(code, touched), tag
        assert np.abs(m - wt["mean"][touched]).max() <= 1e-9 and np.abs(sdev - wt["stdev"][touched]).max() <= 1e-7, tag
        checked += 1
    assert checked >= 60


def test_train_batch_against_oracle_and_pooled_stats(models, al9):
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(5, 6, "rna004", mean, sd, (120, 500))
    res = al9.train_batch([r.signal for r in reads], [r.sequence for r in reads], pooled=True)
    orc = Oracle(models["syn9"], 1)
    K = al9.num_kmers
    pooled = np.zeros(3 * K)
    for i, r in enumerate(reads):
        want = orc.train(r.signal, r.sequence)
        assert abs(res.Z[i] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))
        assert abs(res.transitions[3 * i] - want["m1"]) <= 1e-9 and abs(res.transitions[3 * i + 2] - want["e2"]) <= 1e-9
        code, m, s = res.sparse(i)
        touched = np.nonzero(want["weight"] > 0)[0]
        assert np.array_equal(code, touched)
        assert np.abs(m - want["mean"][touched]).max() <= 1e-9
        assert np.abs(s - want["stdev"][touched]).max() <= 1e-7
        pooled[:K] += want["weight"]
        pooled[K:2 * K] += want["sum"]
        pooled[2 * K:] += want["sumsq"]
    assert np.allclose(res.pooled, pooled, rtol=1e-9, atol=1e-9)
    # the device-resident pooled statistics (what a multi-GPU job all-reduces) agree with the host sum.
    # Read back with the HIP runtime this process already loaded (no torch here: a PyTorch wheel
    # bundles its own libamdhip64 and must be imported BEFORE dynamont_amd if both are used).
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")
    with al9.batch([r.signal for r in reads], [r.sequence for r in reads]) as b:
        b.train()
        ptr, cnt = b.device_pooled()
        dev = np.empty(cnt)
        rc = hip.hipMemcpy(ctypes.c_void_p(dev.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(cnt * 8), 2)  # D2H
        assert rc == 0
    assert cnt == 3 * K and np.allclose(dev, pooled, rtol=1e-9, atol=1e-9)


def test_traceback_segments_spanning_many_64_row_blocks(models):
    """The traceback walks 64 rows at a time and jumps a segment per step: exercise dwell times from the
    minimum (2 samples) to stalls of several hundred samples, turning points on block boundaries,
    and M cells that fall into the next block."""
    pore = "dna_r9"
    path = model_for(models, pore)
    _, mean, sd = synth.read_model_file(path)
    mean_c, sd_c = synth.code_order_table(mean, sd, 5, False)
    rng = np.random.default_rng(2024)
    sigs, seqs = [], []
    for rep in range(10):
        nb = int(rng.integers(20, 160))
        digits = rng.integers(0, 4, size=nb)
        codes = synth._seq_codes(digits, 5)
        dwell = rng.choice([2, 2, 3, 5, 9, 63, 64, 65, 127, 128, 129, 300, 700], size=len(codes))
        if rep == 0:
            dwell[:] = 64      # every turning point on a block boundary
        if rep == 1:
            dwell[:] = 2       # S == 2*Kc
        idx = np.repeat(codes, dwell)
        sigs.append(mean_c[idx] + 0.9 * sd_c[idx] * rng.standard_normal(len(idx)))
        seqs.append("".join("ACGT"[d] for d in digits))
    al = Aligner(path, pore, device=0)
    orc = Oracle(path, synth.PORES[pore][0])
    res = al.align_batch(sigs, seqs, True)
    for i in range(len(sigs)):
        want = orc.align(sigs[i], seqs[i], True)
        got = res.read(i)
        assert np.array_equal(got["signal_positions"], want["signal_positions"])
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"])
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT
        assert abs(got["Z"] - want["Z"]) <= 1e-9 * max(1.0, abs(want["Z"]))


def test_empty_and_all_failed_batches(al5):
    """Ragged/empty inputs: an empty batch and a batch in which every read is rejected on the host
    never reach a kernel and still answer per read."""
    res = al5.align_batch([], [], True)
    assert res.n == 0 and len(res.Z) == 0
    res = al5.align_batch([np.zeros(0), np.zeros(5), np.ones(40)], ["ACGTACGT", "ACG", "ACGTNACGTA"], True)
    assert [res.error(i) for i in range(3)] == ["Signal is empty", "Sequence shorter than model kmer size",
                                                 "Invalid nucleotide: N"]
    assert res.n_segments.sum() == 0
    tr = al5.train_batch([np.zeros(0)], ["ACGTACGT"])
    assert tr.error(0) == "Signal is empty" and tr.em_count[0] == 0
    # a Z mismatch is reported with the reference's text and does not disturb its neighbours
    good = np.random.default_rng(0).standard_normal(60)
    res = al5.align_batch([good, np.full(60, np.inf), good], ["ACGTACGTAC"] * 3, True)
    assert res.status[0] == 0 and res.status[2] == 0
    assert res.error(1) == "Alignment failed: alignment scores do not match"
    assert np.array_equal(res.read(0)["signal_positions"], res.read(2)["signal_positions"])
    one_inf = good.copy()
    one_inf[17] = -np.inf
    assert al5.align_batch([one_inf], ["ACGTACGTAC"], False).error(0) == "Alignment failed: alignment scores do not match"
    assert al5.train_batch([one_inf], ["ACGTACGTAC"]).error(0) == "Training failed: alignment scores do not match"


def test_many_small_reads_in_one_batch(models):
    """More reads than the chip has SIMDs (4 096 reads -> 1 024 workgroups per DP kernel, one block per
    read in the traceback): dispatch order, LPT sorting and the per-read offsets must not mix reads up."""
    pore = "dna_r9"
    path = model_for(models, pore)
    _, mean, sd = synth.read_model_file(path)
    reads = synth.make_reads(4096, 4096, pore, mean, sd, (8, 70), dwell=4.0)
    al = Aligner(path, pore, device=0)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    assert int((res.status == 0).sum()) >= 4000
    orc = Oracle(path, synth.PORES[pore][0])
    for i in list(range(0, 4096, 131)) + [4095]:
        r = reads[i]
        try:
            want = orc.align(r.signal, r.sequence, True)
        except RuntimeError as e:
            assert res.error(i) == str(e)
            continue
        got = res.read(i)
        assert np.array_equal(got["signal_positions"], want["signal_positions"])
        assert np.array_equal(got["sequence_positions"], want["sequence_positions"])
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT
    # empty batch and single read through the same entry point
    none = al.align_batch([], [], True)
    assert none.n == 0
    one = al.align_batch([reads[7].signal], [reads[7].sequence], True)
    assert np.array_equal(one.read(0)["signal_positions"], res.read(7)["signal_positions"])


def test_stalled_pore_long_segments(models):
    """A pore that sits on one k-mer for 20 000 samples (polyA / adapter stalls in real RNA reads): segments beyond
    256 rows take the radix-select median (k_median_long) instead of the quadratic rank count. Parity with the
    oracle on such reads, odd and even segment lengths, and the per-segment kernels must not blow up in time."""
    _, mean, sd = synth.read_model_file(models["syn9"])
    mean_c, sd_c = synth.code_order_table(mean, sd, 9, True)
    base = synth.make_reads(321, 4, "rna004", mean, sd, (300, 400))
    al = Aligner(models["syn9"], "rna004", device=0)
    orc = Oracle(models["syn9"], 1)

    def with_stall(r, where, length, seed):
        # the k-mer at `where` keeps emitting for `length` more samples
        codes = orc.kmers(r.sequence)
        rng = np.random.default_rng(seed)
        extra = mean_c[codes[where]] + 1.3 * sd_c[codes[where]] * rng.standard_normal(length)
        cut = int(len(r.signal) * where / len(codes))
        return synth.SynthRead(np.concatenate([r.signal[:cut], extra, r.signal[cut:]]), r.sequence)

    stalled = [with_stall(base[0], 100, 20000, 1), with_stall(base[1], 50, 20001, 2), with_stall(base[2], 200, 257, 3),
               with_stall(with_stall(base[3], 30, 3000, 4), 250, 5001, 5)]
    plain = [synth.SynthRead(np.concatenate([r.signal, r.signal[-20:]]), r.sequence) for r in base]
    times = {}
    for tag, reads in (("plain", plain), ("stalled", stalled)):
        with al.batch([r.signal for r in reads], [r.sequence for r in reads]) as b:
            b.align(True)
            res = b.fetch()
            tm = b.timing()
            times[tag] = tm["ms_total"] - tm["ms_dp"]      # k_median + k_median_long + k_final
        assert (res.status == 0).all()
        longest = 0
        for i, r in enumerate(reads):
            want, got = orc.align(r.signal, r.sequence, True), res.read(i)
            assert np.array_equal(got["signal_positions"], want["signal_positions"])
            assert np.abs(got["probabilities"] - want["probabilities"]).max() <= PROB_TIGHT
            longest = max(longest, int(np.diff(np.append(want["signal_positions"], len(r.signal))).max()))
        assert (longest > 19000) == (tag == "stalled")
    # 4e8 compares in one segment used to take ~100 ms here; now the stalled batch costs about what its rows cost
    assert times["stalled"] < 2.0 * times["plain"] + 1.0, times
    al.close()


def _orc_train(job):
    sig, seq = job
    t = _ORC.train(sig, seq, dense=False)
    return t["Z"], t["m1"], t["e2"], t["weight"], t["sum"], t["sumsq"]


def test_cfg5_share_full_size_train(models, al9):
    """BASELINE configs[4]'s per-GPU share: 1 024 RNA004 reads x ~20 k samples through train(). Size-independent
    properties on all reads (every lattice row carries posterior mass 1, every path has exactly one M cell per
    k-mer), full parity with the oracle on 8 reads, and the device-pooled statistics against the per-read ones."""
    import multiprocessing as mp
    _, mean, sd = synth.read_model_file(models["syn9"])
    cfg = synth.CONFIGS["cfg5"]
    reads = synth.make_reads(cfg["seed"], 1024, cfg["pore"], mean, sd, cfg["n_bases"])
    sig, so, sq, qo = synth.pack_reads(reads)
    t = al9.train_async(sig, so, sq, qo, pooled=True)
    res = t.wait()
    tm = t.timing()
    t.close()
    assert (res.status == 0).all() and tm["reads_ok"] == 1024 and tm["launches"] == 1
    K = al9.num_kmers
    k = al9.kmer_size
    for i, r in enumerate(reads):
        a, n = int(res.em_offsets[i]), int(res.em_count[i])
        # expected counts: sum over k-mers of w = S rows (one path cell per row); one E->M transition per k-mer; every
        # E cell is entered from M (once per k-mer) or from E: E->E = (S - Kc) - Kc
        assert abs(res.em_weight[a:a + n].sum() - len(r.signal)) <= 1e-6 * len(r.signal)
        kc = len(r.sequence) - k + 1
        assert abs(res.trans_counts[2 * i] - kc) <= 1e-6 * kc
        assert abs(res.trans_counts[2 * i + 1] - (len(r.signal) - 2 * kc)) <= 1e-6 * len(r.signal)
        assert np.all(res.em_stdev[a:a + n] > 0) and np.all(np.isfinite(res.em_mean[a:a + n]))
    assert abs(res.pooled[:K].sum() - float(so[-1])) <= 1e-6 * float(so[-1])
    picks = [0, 100, 333, 512, 700, 901, 1000, 1023]
    with mp.get_context("fork").Pool(8, initializer=_orc_init, initargs=(models["syn9"], 1)) as pool:
        want = pool.map(_orc_train, [(reads[i].signal, reads[i].sequence) for i in picks], chunksize=1)
    for i, (Z, m1, e2, w, s1, s2) in zip(picks, want):
        assert abs(res.Z[i] - Z) <= 1e-9 * abs(Z)
        assert abs(res.transitions[3 * i] - m1) <= 1e-9 and abs(res.transitions[3 * i + 2] - e2) <= 1e-9
        code, m, s = res.sparse(i)
        touched = np.nonzero(w > 0)[0]
        assert np.array_equal(code, touched)
        assert np.abs(m - s1[touched] / w[touched]).max() <= 1e-9
        a = int(res.em_offsets[i])
        # 1e-7, not 1e-9: the oracle (= the reference's arithmetic) adds ~20 k log-space terms of magnitude 4e4 per
        # path, each rounding the probability by 7e-12 -- its weights carry up to 6e-9 of noise, the posterior chain
        # 4e-11 (tests/extended_precision_train.py adjudicates in 80-bit; profiles/r03/train_precision.txt)
        assert np.allclose(res.em_weight[a:a + len(code)], w[touched], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("cus,frac", [(16, 0.5), (16, 0.12), (8, 0.3), (256, 0.2)])
def test_page_starved_queues_equal_the_unconstrained_run(models, al9, monkeypatch, cus, frac):
    """Few persistent workgroups (DYN_QUEUE_CUS) and a pool far too small for them: the planned queue, waves that
    change arenas on the device, surplus pages handed to waiting waves. Whatever the schedule, every read must come
    out exactly as in the unconstrained launch with the same posterior layout."""
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(9000 + cus, 260, "rna004", mean, sd, (100, 1500))
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    monkeypatch.setenv("DYN_FORCE_LAYOUT", "inplace")
    base = al9.align_batch(sigs, seqs, True)
    monkeypatch.setenv("DYN_QUEUE_CUS", str(cus))
    small = Aligner(models["syn9"], "rna004", device=0)
    monkeypatch.delenv("DYN_QUEUE_CUS")
    n_slots = min(len(reads), 4 * cus)
    lens = sorted((len(x) for x in sigs), reverse=True)
    wanted = sum(lens[:n_slots]) * (448 * 8 + 56)                      # in-place lattices of the reads in flight
    small.set_mem_budget(int(max(frac * wanted, 1.3 * lens[0] * (448 * 8 + 56))))
    with small.batch(sigs, seqs) as b:
        b.align(True)
        got = b.fetch()
        tm = b.timing()
    assert tm["launches"] == 1 and tm["lp_inplace"] == 1 and tm["n_waves"] == min(4 * cus, (len(reads) + 3) // 4 * 4)
    assert (got.status == 0).all() and (base.status == 0).all()
    for i in range(len(reads)):
        a, c = base.read(i), got.read(i)
        assert np.array_equal(a["signal_positions"], c["signal_positions"]) and a["Z"] == c["Z"]
        assert np.array_equal(a["probabilities"], c["probabilities"])
    # the training sweeps on the same starved pool: bit for bit the unconstrained statistics
    tb, ts = al9.train_batch(sigs, seqs), small.train_batch(sigs, seqs)
    assert (tb.status == 0).all() and (ts.status == 0).all()
    assert np.array_equal(tb.Z, ts.Z) and np.array_equal(tb.em_offsets, ts.em_offsets)
    assert np.array_equal(tb.em_code, ts.em_code) and np.array_equal(tb.em_weight, ts.em_weight)
    assert np.array_equal(tb.em_sum, ts.em_sum) and np.array_equal(tb.em_sumsq, ts.em_sumsq)
    small.close()


def test_g11_ntk_mode_answers_like_the_reference(models):
    """mode "resquiggle"/"ntk": the reference's NTKAligner fails every read that passes validation with one message
    and does not train (G11, generated from the compiled reference); these texts reach `.errors`, so they are output."""
    g = json.load(open(os.path.join(GOLDEN, "g11_ntk_messages.json")))
    for pore in ("rna002", "dna_r9", "rna004"):
        cases = [c for c in g["cases"] if c["pore"] == pore]
        al = Aligner(models[cases[0]["model"]], pore, mode="resquiggle", device=0)
        res = al.align_batch([np.array(c["signal"]) for c in cases], [c["sequence"] for c in cases], True)
        assert [res.error(i) for i in range(len(cases))] == [c["align"] for c in cases]
        assert int(res.n_segments.sum()) == 0
        for c in cases[:2]:
            with pytest.raises(RuntimeError) as e:
                al.align(np.array(c["signal"]), c["sequence"], True)
            assert str(e.value) == c["align"]
            with pytest.raises(RuntimeError) as e:
                al.train(np.array(c["signal"]), c["sequence"])
            assert str(e.value) == c["train"]
        al.close()



def test_band_above_447_short_reads_equal_the_oracle_and_long_reads_take_the_generic_kernel(models):
    """The reference takes any band (aligner.cpp:21). A handle created with band 1 000 computes every read whose half band
    min(band / 2, columns / 2) fits the register sweeps' 448 band slots -- here reads of 60 .. 446 k-mers, for the longer of
    which (402+ columns) the band is WIDER than at band 400 -- exactly as the oracle does at band 1 000, train() included. A read
    with more columns (half band 348 here) takes the generic kernel (wide_band.hip) IN THE SAME BATCH: the reference's own
    arithmetic in every cell, so its Z has the oracle's bits."""
    model = models["syn5"]
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(515, 40, "rna002", mean, sd, (64, 450))
    reads += synth.make_reads(516, 12, "rna002", mean, sd, (430, 450))     # 426 .. 446 k-mers: half band 213 .. 223
    long_read = synth.make_reads(517, 1, "rna002", mean, sd, 700)[0]
    al = Aligner(model, "rna002", band=1000, device=0)
    assert al.info.max_half_band == 2046
    orc = Oracle(model, synth.PORES["rna002"][0], 1000)
    sigs, seqs = [r.signal for r in reads] + [long_read.signal], [r.sequence for r in reads] + [long_read.sequence]
    res = al.align_batch(sigs, seqs, True)
    for i, r in enumerate(reads + [long_read]):
        assert res.status[i] == 0, (i, res.error(i))
        got, want = res.read(i), orc.align(r.signal, r.sequence, True)
        assert np.array_equal(got["signal_positions"], want["signal_positions"]) and np.array_equal(got["sequence_positions"], want["sequence_positions"]), i
        assert np.abs(got["probabilities"] - want["probabilities"]).max() <= 1e-6 and abs(got["Z"] - want["Z"]) <= 1e-9 * abs(want["Z"])
    assert res.read(len(reads))["Z"] == orc.align(long_read.signal, long_read.sequence, True)["Z"]   # the wide read: bit for bit
    # the wider band is a different computation for the reads above 401 columns: more lattice cells in the launch
    wide = [r for r in reads if len(r.sequence) - 4 + 1 > 401]
    cells = {}
    for band in (400, 1000):
        h = al if band == 1000 else Aligner(model, "rna002", band=400, device=0)
        with h.batch([r.signal for r in wide], [r.sequence for r in wide]) as b:
            b.align(True)
            cells[band] = b.timing()["cells"]
        if h is not al:
            h.close()
    assert len(wide) >= 12 and cells[1000] > cells[400]
    tr = al.train_batch(sigs, seqs)
    assert (tr.status == 0).all()
    al.close()


@pytest.mark.parametrize("pore,band", [("rna004", 600), ("rna004", 894), ("dna_r10_400bps", 1000), ("rna002", 4093)])
def test_any_band_long_reads_equal_the_oracle_bit_for_bit(models, pore, band):
    """Reads of 900 .. 2 600 bases under bands the register sweeps cannot hold (half band 300 .. 1 300): the generic kernel's
    borders, Z BITS (forward and backward are the reference's operation by operation), posteriors, the Z-only call and
    train() against the oracle at the same band -- among them reads that START with a structural tie (the polyA pad followed by
    A's), where the reference's choice rests on the last bits of its logPlus. Asynchronous tickets take the same path (no
    resident session for them)."""
    model = model_for(models, pore)
    _, mean, sd = synth.read_model_file(model)
    lengths = ((900, 2600) if "rna" in pore else (900, 1500)) if band < 4000 else (2500, 2600)   # (the oracle on the CPU is what takes the time)
    reads = synth.make_reads(5200 + band, (4 if "rna" in pore else 3) if band < 4000 else 1, pore, mean, sd, lengths)
    if "rna" in pore:
        reads += synth.make_reads(5300 + band, 2 if band < 4000 else 1, pore, mean, sd, (1000, 1200), polya=(20, 60))
    al = Aligner(model, pore, band=band, device=0)
    orc = Oracle(model, synth.PORES[pore][0], band)
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    res = al.align_batch(sigs, seqs, True)
    t = al.align_async(*synth.pack_reads(reads), True)
    res_async = t.wait()
    assert t.timing()["launches"] == 1
    zonly = al.align_batch(sigs, seqs, False)
    tr = al.train_batch(sigs, seqs)
    for i, r in enumerate(reads):
        want = orc.align(r.signal, r.sequence, True)
        for got_res in (res, res_async):
            assert got_res.status[i] == 0, (i, got_res.error(i))
            got = got_res.read(i)
            assert np.array_equal(got["signal_positions"], want["signal_positions"]), i
            assert np.array_equal(got["sequence_positions"], want["sequence_positions"]), i
            assert got["Z"] == want["Z"], (i, got["Z"], want["Z"])
            assert np.abs(got["probabilities"] - want["probabilities"]).max() <= 1e-6
        assert zonly.status[i] == 0 and zonly.Z[i] == want["Z"] and zonly.n_segments[i] == 0
        wt = orc.train(r.signal, r.sequence, dense=False)
        assert tr.status[i] == 0 and tr.Z[i] == wt["Z"]
        assert abs(tr.transitions[3 * i] - wt["m1"]) <= 1e-8 and abs(tr.transitions[3 * i + 2] - wt["e2"]) <= 1e-8
        a0, cnt = int(tr.em_offsets[i]), int(tr.em_count[i])
        codes = tr.em_code[a0:a0 + cnt]
        assert cnt > 100 and np.allclose(tr.em_weight[a0:a0 + cnt], wt["weight"][codes], rtol=1e-7, atol=1e-9)
        assert np.allclose(tr.em_sum[a0:a0 + cnt], wt["sum"][codes], rtol=1e-7, atol=1e-9)
    t.close()
    al.close()


def test_band_beyond_the_generic_kernels_rows_is_a_per_read_status(models):
    """4 096 band columns per row is what the generic kernel holds in LDS: a read whose half band exceeds 2 046 (band > 4 093
    AND more than 4 093 lattice columns) gets DYN_READ_BAND_TOO_WIDE and leaves the others alone."""
    model = models["syn5"]
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(5400, 3, "dna_r9", mean, sd, (300, 600))
    huge = synth.make_reads(5401, 1, "dna_r9", mean, sd, 4300, dwell=3.0)[0]
    al = Aligner(model, "dna_r9", band=5000, device=0)
    orc = Oracle(model, synth.PORES["dna_r9"][0], 5000)
    res = al.align_batch([r.signal for r in reads] + [huge.signal], [r.sequence for r in reads] + [huge.sequence], True)
    assert res.status[-1] == 11 and res.error(3) == "Band wider than this build's 4096 band columns for a read of this length"
    for i, r in enumerate(reads):
        want = orc.align(r.signal, r.sequence, True)
        assert res.status[i] == 0 and np.array_equal(res.read(i)["signal_positions"], want["signal_positions"])
    al.close()


def _g13_clustered9(mean5, sd5):
    """tests/golden/make_golden_g13.py, clustered9(): every 9-mer takes its central 5-mer's entry, moved by +-10^-(4 + code % 4)"""
    code = np.arange(4 ** 9, dtype=np.int64)
    central = (code // 16) % 1024
    sign = np.where((code * 2654435761 >> 7) & 1, 1.0, -1.0)
    return mean5[central] + sign * 10.0 ** -(4 + code % 4), sd5[central].copy()


@pytest.mark.parametrize("family", ["rna002_real", "rna002_trained", "clustered9", "dna_cfg3"])
def test_g13_decision_margin_families_in_the_default_mode(models, tmp_path, family):
    """G13 (tests/golden/make_golden_g13.py): the tables where distinct k-mers lie closest together -- the reference's real RNA002
    models (one with a stdev per k-mer), a 9-mer table with real-like clustering (256 nine-mers 1e-4 .. 1e-7 around each
    real 5-mer level), and configs[2]'s long DNA reads. The strict mode "ties" certifies a read only on IDENTICAL neighbouring
    parameters; everything else rests on the traceback's on-path margin (NT_aligner_api.cpp:445-448), which the fixture
    records per read from the compiled reference: its floor over all four families is 6e-7, seven orders above what the table
    softplus can move. Every read runs here in the DEFAULT mode and must come out on the reference's borders."""
    import json
    g = golden("g13_margin_families.npz")
    spec = json.loads(str(g["families"]))[family]
    pore = spec["pore"]
    k = synth.PORES[pore][2]
    if spec["table"] == "syn9":
        model = models["syn9"]
    else:
        if spec["table"] == "clustered9":
            mean, sd = _g13_clustered9(g["rna004_5_mean"], g["rna004_5_sd"])
        else:
            mean, sd = g[spec["table"] + "_mean"], g[spec["table"] + "_sd"]
        model = synth.write_model_values(str(tmp_path / f"{family}.model"), k, mean, sd)
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(spec["seed"], spec["reads"], pore, np.asarray(mean), np.asarray(sd), tuple(spec["bases"]))
    assert g[f"{family}_margin_distinct"].min() >= 1e-7   # what "far above 1e-13" means here; below 1e-9 the tie rule would need a net
    al = Aligner(model, pore, device=0)
    res = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    offs, want_sp, want_z = g[f"{family}_offsets"], g[f"{family}_signal_positions"], g[f"{family}_Z"]
    assert (res.status == 0).all()
    bad = []
    for i in range(len(reads)):
        got = res.read(i)
        if not np.array_equal(np.asarray(got["signal_positions"], dtype=np.int64), want_sp[offs[i]:offs[i + 1]].astype(np.int64)):
            bad.append(i)
        assert abs(got["Z"] - want_z[i]) <= 1e-12 * max(1.0, abs(want_z[i]))
    assert not bad, f"{family}: reads off the reference's borders: {bad[:10]}"
    al.close()
