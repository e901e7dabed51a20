"""GPU (-m gpu): the asynchronous pipeline (dyn_batch_align_async / _train_async / _wait) must
return exactly what the synchronous calls return -- several batches in flight, failed reads isolated
per read, pageable and page-locked caller memory -- and stay correct against the CPU oracle."""
import numpy as np
import pytest

from dynamont_amd import Aligner, synth
from dynamont_amd._dynamont import pinned_empty
from oracle.pyoracle import Oracle

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("native_lib", "oracle_built")]


def _batches(models, n_batches, n_reads, seed0=900, bases=(60, 400)):
    _, mean, sd = synth.read_model_file(models["syn9"])
    out = []
    for j in range(n_batches):
        reads = synth.make_reads(seed0 + j, n_reads + j, "rna004", mean, sd, bases)
        out.append((reads, synth.pack_reads(reads)))
    return out


def _same(a, b):
    assert np.array_equal(a.status, b.status)
    assert np.array_equal(a.Z, b.Z)
    assert np.array_equal(a.n_segments, b.n_segments)
    assert np.array_equal(a.seg_offsets, b.seg_offsets)
    for i in range(a.n):
        lo, hi = int(a.seg_offsets[i]), int(a.seg_offsets[i]) + int(a.n_segments[i])
        assert np.array_equal(a.signal_positions[lo:hi], b.signal_positions[lo:hi])
        assert np.array_equal(a.sequence_positions[lo:hi], b.sequence_positions[lo:hi])
        assert np.array_equal(a.probabilities[lo:hi], b.probabilities[lo:hi])
        assert np.array_equal(a.states[lo:hi], b.states[lo:hi])


def test_async_equals_sync_with_batches_in_flight(models):
    al = Aligner(models["syn9"], "rna004", device=0)
    data = _batches(models, 6, 24)
    want = []
    for reads, _ in data:
        want.append(al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True))
    tickets = [al.align_async(*packed, True) for _, packed in data]  # all six in flight at once
    for t, w in zip(tickets, want):
        _same(t.wait(), w)
        tm = t.timing()
        assert tm["reads_ok"] == int((w.status == 0).sum()) and tm["ms_total"] > 0
        ptr, cap, st = t.device_results()
        assert ptr and st and cap == w.seg_offsets[-1]
    for t in tickets:
        t.close()
    # result objects can be refilled by later batches
    t = al.align_async(*data[0][1], True, out=want[3])
    _same(t.wait(), al.align_batch([r.signal for r in data[0][0]], [r.sequence for r in data[0][0]], True))
    t.close()
    al.close()


def test_waiting_tickets_share_one_launch(models, monkeypatch):
    """Merged launches (async_engine.cpp): tickets that wait while the GPU is busy run as ONE read-queue launch. Eight
    tickets of 300 reads x ~4 k samples submitted at once: the first starts alone, later ones share launches (their
    dyn_timing.launch_share is < 1 and the shares add up to the number of launches), every ticket's results and its slice
    of the device rows equal the synchronous call's, a failed read stays with its own ticket, tickets of another kind
    (calc_probabilities = false) are not merged with them -- and DYN_NO_MERGE=1 gives every ticket its own launch."""
    import ctypes as C
    _, mean, sd = synth.read_model_file(models["syn9"])
    data = []
    for j in range(8):
        reads = synth.make_reads(1300 + j, 300, "rna004", mean, sd, (300, 500))
        if j == 5:
            reads[7] = synth.SynthRead(reads[7].signal, reads[7].sequence[:40] + "N" + reads[7].sequence[41:])
        data.append((reads, synth.pack_reads(reads)))
    al = Aligner(models["syn9"], "rna004", device=0)
    want = [al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True) for reads, _ in data]
    for attempt in range(3):   # (whether tickets meet in the queue is a matter of timing: 2 ms of lingering make it all but certain)
        tickets = [al.align_async(*packed, True) for _, packed in data]
        z_only = al.align_async(*data[0][1], False)
        shares = []
        for t, w in zip(tickets, want):
            _same(t.wait(), w)
            tm = t.timing()
            shares.append(tm["launch_share"])
            assert tm["reads_ok"] == int((w.status == 0).sum()) and tm["ms_dp"] > 0 and tm["launches"] == 1
            ptr, cap, st = t.device_results()
            assert cap == w.seg_offsets[-1]
            rows = np.empty(int(cap), dtype=[("signal_pos", "<u4"), ("sequence_pos", "<u4"), ("probability", "<f8")])
            hip = C.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(C.c_void_p(rows.ctypes.data), C.c_void_p(ptr), C.c_size_t(rows.nbytes), 2) == 0   # device -> host
            for i in range(w.n):
                lo, ns = int(w.seg_offsets[i]), int(w.n_segments[i])
                assert np.array_equal(rows["signal_pos"][lo:lo + ns], w.signal_positions[lo:lo + ns].astype(np.uint32))
        zr = z_only.wait()
        assert np.array_equal(zr.status, want[0].status) and z_only.timing()["launch_share"] == 1.0
        assert np.allclose(zr.Z, want[0].Z, rtol=1e-9)
        for t in tickets + [z_only]:
            t.close()
        merged = sum(1 for x in shares if x < 1.0)
        if merged >= 2:
            break
    # (how MANY tickets meet in the queue is scheduling; that waiting tickets DO share a launch -- at least one merged pair in
    # three attempts of eight tickets each -- is the property. On the GPU boxes: 6-7 of 8, first attempt.)
    assert merged >= 2, shares
    n_launches = sum(shares)
    assert abs(n_launches - round(n_launches)) < 1e-9 and round(n_launches) < len(shares)
    assert want[5].status[7] != 0 and sum(int((w.status != 0).sum()) for w in want) == 1
    al.close()
    monkeypatch.setenv("DYN_NO_MERGE", "1")
    al = Aligner(models["syn9"], "rna004", device=0)
    tickets = [al.align_async(*packed, True) for _, packed in data[:4]]
    for t, w in zip(tickets, want):
        _same(t.wait(), w)
        assert t.timing()["launch_share"] == 1.0
        t.close()
    al.close()


def test_async_isolates_failed_reads_and_matches_oracle(models):
    al = Aligner(models["syn9"], "rna004", device=0)
    orc = Oracle(models["syn9"], 1)
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(77, 10, "rna004", mean, sd, (80, 300))
    sigs = [r.signal for r in reads]
    seqs = [r.sequence for r in reads]
    seqs[2] = seqs[2][:40] + "N" + seqs[2][41:]           # Invalid nucleotide: N
    sigs[5] = sigs[5][:50]                                 # Signal too short compared to sequence
    seqs[7] = "ACGT"                                       # Sequence shorter than model kmer size
    sigs[8] = np.zeros(0)                                  # Signal is empty
    packed = synth.pack_reads([synth.SynthRead(np.asarray(s, dtype=np.float64), q) for s, q in zip(sigs, seqs)])
    t = al.align_async(*packed, True)
    res = t.wait()
    assert [res.error(i) for i in (2, 5, 7, 8)] == ["Invalid nucleotide: N", "Signal too short compared to sequence",
                                                    "Sequence shorter than model kmer size", "Signal is empty"]
    for i in (0, 1, 3, 4, 6, 9):
        got, ref = res.read(i), orc.align(sigs[i], seqs[i], True)
        assert np.array_equal(got["signal_positions"], ref["signal_positions"])
        assert np.array_equal(got["sequence_positions"], ref["sequence_positions"])
        assert np.abs(got["probabilities"] - ref["probabilities"]).max() <= 1e-6
        assert abs(got["Z"] - ref["Z"]) <= 1e-9 * abs(ref["Z"])
    t.close()
    # Z only
    t = al.align_async(*packed, False)
    rz = t.wait()
    assert np.array_equal(rz.status, res.status) and int(rz.n_segments.sum()) == 0
    ok = res.status == 0
    # (not bit-equal: Z-only jobs run the cheap arithmetic -- two-operation emission, degree-3 softplus polynomial)
    assert np.allclose(rz.Z[ok], res.Z[ok], rtol=1e-11, atol=0)
    t.close()
    al.close()


def test_async_pinned_inputs_and_unwaited_destroy(models):
    al = Aligner(models["syn9"], "rna004", device=0)
    (reads, (sig, sig_off, seqs, seq_off)), = _batches(models, 1, 32, seed0=31)
    want = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    ps = pinned_empty(sig.size, np.float64)
    ps[:] = sig
    t = al.align_async(ps, sig_off, seqs, seq_off, True)
    _same(t.wait(), want)
    t.close()
    t = al.align_async(ps, sig_off, seqs, seq_off, True)
    t.close()  # never waited for: destroy waits
    al.close()  # handle destruction drains the pipeline


def test_train_async_equals_train(models):
    al = Aligner(models["syn9"], "rna004", device=0)
    data = _batches(models, 3, 6, seed0=400, bases=(60, 200))
    for reads, packed in data:
        want = al.train_batch([r.signal for r in reads], [r.sequence for r in reads], pooled=True)
        t = al.train_async(*packed, pooled=True)
        got = t.wait()
        assert np.array_equal(got.status, want.status) and np.array_equal(got.Z, want.Z)
        assert np.array_equal(got.transitions, want.transitions)
        assert np.array_equal(got.em_count, want.em_count)
        n = int(got.em_offsets[-1])
        assert np.array_equal(got.em_code[:n], want.em_code[:n]) and np.array_equal(got.em_mean[:n], want.em_mean[:n])
        assert np.array_equal(got.pooled, want.pooled)
        ptr, cnt = t.device_pooled()
        assert ptr and cnt == 3 * al.num_kmers
        t.close()
        # statistics-only form: no per-read emission arrays, same Z / transition counts
        t = al.train_async(*packed, pooled=False, emissions=False)
        lean = t.wait()
        assert np.array_equal(lean.Z, want.Z) and np.array_equal(lean.trans_counts, want.trans_counts)
        t.close()
    al.close()


def test_multi_device_handle_equals_single_device(models):
    """dyn_multi_*: one handle, several devices (here the same GPU twice and three times): the batch is cut into
    contiguous ranges, each through its own pipeline, results written straight into one set of caller arrays --
    bitwise what a single handle returns, failed reads included, whatever the number of ranges."""
    from dynamont_amd import MultiAligner
    al = Aligner(models["syn9"], "rna004", device=0)
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(4242, 37, "rna004", mean, sd, (60, 500))
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    seqs[5] = "ACGT"                                  # Sequence shorter than model kmer size: no segment rows at all
    seqs[20] = seqs[20][:30] + "N" + seqs[20][31:]    # Invalid nucleotide
    want = al.align_batch(sigs, seqs, True)
    wt = al.train_batch(sigs, seqs, pooled=True)
    for devs in ([0, 0], [0, 0, 0], [0]):
        m = MultiAligner(models["syn9"], "rna004", devs)
        assert m.n_devices == len(devs)
        _same(m.align_batch(sigs, seqs, True), want)
        z = m.align_batch(sigs, seqs, False)
        assert np.array_equal(z.status, want.status) and int(z.n_segments.sum()) == 0
        tr = m.train_batch(sigs, seqs, pooled=True)
        assert np.array_equal(tr.status, wt.status) and np.array_equal(tr.Z, wt.Z)
        assert np.array_equal(tr.transitions, wt.transitions) and np.array_equal(tr.em_count, wt.em_count)
        assert np.array_equal(tr.em_offsets, wt.em_offsets)
        for i in range(len(reads)):
            a, n = int(tr.em_offsets[i]), int(tr.em_count[i])
            assert np.array_equal(tr.em_code[a:a + n], wt.em_code[a:a + n]) and np.array_equal(tr.em_mean[a:a + n], wt.em_mean[a:a + n])
        assert np.allclose(tr.pooled, wt.pooled, rtol=1e-13, atol=1e-13)   # summed per range, then over ranges
        m.close()
    with pytest.raises(ValueError, match="Unknown pore type"):
        MultiAligner(models["syn9"], "nope", [0])
    al.close()


def test_a_new_aligner_takes_over_the_parked_pool_of_its_predecessor(models):
    """The reference's training loop builds an Aligner per batch. A destroyed handle parks its lattice pool; the next
    handle on the device takes it over (no second multi-second allocation), gives the same results, still honours its
    own memory budget, and release_cached_memory() lets go of what is parked."""
    import time
    import dynamont_amd
    from dynamont_amd import synth
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(4711, 512, "rna004", mean, sd, (1500, 2000))
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    dynamont_amd.release_cached_memory()

    def once(budget=None):
        al = Aligner(models["syn9"], "rna004", device=0)
        if budget:
            al.set_mem_budget(budget)
        t0 = time.time()
        with al.batch(sigs, seqs) as b:
            b.train()
            res, tm = b.fetch_train(), b.timing()
        dt = time.time() - t0
        al.close()
        return res, tm, dt

    r1, t1, d1 = once()
    r2, t2, d2 = once()
    assert t2["pool_pages"] == t1["pool_pages"]
    assert np.array_equal(r1.Z, r2.Z) and np.array_equal(r1.em_weight, r2.em_weight)
    assert d2 < 0.6 * d1 + 0.2, (d1, d2)          # the second aligner did not allocate its pool again
    # a handle with a small budget does not use more of a large parked pool than its budget allows
    small_budget = 480 * 256 * 448 * 8
    r3, t3, _ = once(small_budget)
    assert t3["pool_pages"] < t1["pool_pages"] and t3["pool_pages"] * t3["page_rows"] * 448 * 8 <= small_budget, (t1, t3)
    assert np.array_equal(r1.Z, r3.Z) and np.array_equal(r1.em_weight, r3.em_weight)
    dynamont_amd.release_cached_memory()


def test_set_model_equals_an_aligner_built_from_the_written_file(models, tmp_path):
    """The training loop keeps one aligner and replaces its table (dyn_aligner_set_model) where the reference builds a new
    Aligner from the model file it has just written: same table, same results, bit for bit."""
    from dynamont_amd import synth
    from dynamont_amd.segmentation.utils import read_kmer_model, write_kmer_model_arrays
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(99, 24, "rna004", mean, sd, (200, 600))
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    al = Aligner(models["syn9"], "rna004", device=0)
    m0, s0 = al.model_table()
    rng = np.random.default_rng(3)
    m1, s1 = m0 + 0.05 * rng.standard_normal(len(m0)), s0 * np.exp(0.2 * rng.standard_normal(len(s0)))
    # the file the training loop would write: row r of the model file holds the k-mer with code code_of_row[r]
    names = synth.kmer_strings(9)
    code_of_row = synth.code_order_table(np.arange(len(names), dtype=np.float64), np.zeros(len(names)), 9, True)[0]
    # code_order_table maps file-order values into code order: out[code] = in[row]; invert it
    row_of_code = code_of_row.astype(np.int64)
    code_of_row = np.empty_like(row_of_code)
    code_of_row[row_of_code] = np.arange(len(names))
    path = str(tmp_path / "next.model")
    write_kmer_model_arrays(path, "".join(names).encode(), 9, m1[code_of_row], s1[code_of_row])
    fresh = Aligner(path, "rna004", device=0)
    al.set_model(m1, s1)
    fm, fs = fresh.model_table()
    am, asd = al.model_table()
    assert np.array_equal(fm, am) and np.array_equal(fs, asd)
    a, b = al.align_batch(sigs, seqs, True), fresh.align_batch(sigs, seqs, True)
    assert np.array_equal(a.Z, b.Z) and np.array_equal(a.signal_positions, b.signal_positions)
    assert np.array_equal(a.probabilities, b.probabilities)
    ta, tb = al.train_batch(sigs, seqs), fresh.train_batch(sigs, seqs)
    assert np.array_equal(ta.Z, tb.Z) and np.array_equal(ta.em_weight, tb.em_weight) and np.array_equal(ta.em_mean, tb.em_mean)
    al.close()
    fresh.close()
