"""GPU (-m gpu): the RESIDENT read queue (k_session; include/dynamont_mi.h, dyn_session_stats). Asynchronous
align(calc_probabilities=True) tickets of >= 512 reads are published into one launch of resident waves instead of getting
a launch each; every result must be the one-launch-per-batch result BIT FOR BIT (Z bits, integer columns, probabilities),
whatever else goes through the handle meanwhile."""
import numpy as np
import pytest

from dynamont_amd import Aligner, synth

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("native_lib", "oracle_built")]


def _same(a, b):
    assert np.array_equal(a.status, b.status)
    assert np.array_equal(a.Z.view(np.uint64), b.Z.view(np.uint64))
    assert np.array_equal(a.n_segments, b.n_segments)
    assert np.array_equal(a.seg_offsets, b.seg_offsets)
    for i in range(a.n):
        lo, hi = int(a.seg_offsets[i]), int(a.seg_offsets[i]) + int(a.n_segments[i])
        assert np.array_equal(a.signal_positions[lo:hi], b.signal_positions[lo:hi])
        assert np.array_equal(a.sequence_positions[lo:hi], b.sequence_positions[lo:hi])
        assert np.array_equal(a.probabilities[lo:hi].view(np.uint64), b.probabilities[lo:hi].view(np.uint64))


def _data(models, n_batches, n_reads, seed0, bases=(200, 420)):
    _, mean, sd = synth.read_model_file(models["syn9"])
    out = []
    for j in range(n_batches):
        reads = synth.make_reads(seed0 + j, n_reads, "rna004", mean, sd, bases)
        if j == 2:  # a read that fails in sequenceToKmers stays with its own ticket
            reads[11] = synth.SynthRead(reads[11].signal, reads[11].sequence[:40] + "N" + reads[11].sequence[41:])
        out.append((reads, synth.pack_reads(reads)))
    return out


def test_resident_queue_equals_one_launch_per_batch(models):
    al = Aligner(models["syn9"], "rna004", device=0)
    data = _data(models, 6, 640, 4100)
    want = [al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True) for reads, _ in data]
    before = al.session_stats()
    assert before["sessions"] == 0  # the synchronous calls are one launch per batch
    for rnd in range(2):  # the second round opens a new session behind the first
        tickets = [al.align_async(*packed, True) for _, packed in data]
        for t, w in zip(tickets, want):
            _same(t.wait(), w)
            tm = t.timing()
            assert tm["launches"] == 0 and tm["launch_share"] == 0.0  # no launch of its own
            assert tm["reads_ok"] == int((w.status == 0).sum()) and tm["ms_dp"] > 0 and tm["ms_total"] > tm["ms_dp"]
            assert tm["reads_strict"] > 0 and tm["cert_rows"] > 0  # rna004: pad + A reads run the certified sweeps
            ptr, cap, st = t.device_results()
            assert ptr and st and cap == w.seg_offsets[-1]
        for t in tickets:
            t.close()
        s = al.session_stats()
        assert s["aborted"] == 0
        assert s["sessions"] >= rnd + 1 and s["tickets"] == 6 * (rnd + 1)
        assert s["reads"] == (rnd + 1) * sum(int((w.status == 0).sum()) for w in want)
        assert s["ms"] > 0 and 0.0 < s["wave_occupancy"] <= 1.0
    al.close()


def test_other_jobs_between_resident_tickets(models):
    """Z-only tickets, training, small batches and the synchronous calls use the lattice pool as one launch per batch: the
    session is closed and waited for in front of them, and opened again behind them."""
    al = Aligner(models["syn9"], "rna004", device=0)
    data = _data(models, 4, 600, 4300)
    small = _data(models, 1, 24, 4400, bases=(60, 200))[0]
    want = [al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True) for reads, _ in data]
    want_small = al.align_batch([r.signal for r in small[0]], [r.sequence for r in small[0]], True)
    want_z = al.align_batch([r.signal for r in data[1][0]], [r.sequence for r in data[1][0]], False)
    want_tr = al.train_batch([r.signal for r in small[0]], [r.sequence for r in small[0]])
    t0 = al.align_async(*data[0][1], True)
    t1 = al.align_async(*data[1][1], True)
    tz = al.align_async(*data[1][1], False)        # Z only: a classic launch
    ts = al.align_async(*small[1], True)           # 24 reads join the session that is open, or run alone
    t2 = al.align_async(*data[2][1], True)
    ttr = al.train_async(*small[1])
    t3 = al.align_async(*data[3][1], True)
    for t, w in ((t0, want[0]), (t1, want[1]), (t2, want[2]), (t3, want[3]), (ts, want_small)):
        _same(t.wait(), w)
    rz = tz.wait()
    assert np.array_equal(rz.status, want_z.status) and np.array_equal(rz.Z, want_z.Z)
    rt = ttr.wait()
    assert np.array_equal(rt.status, want_tr.status) and np.allclose(rt.Z, want_tr.Z, rtol=1e-12)
    # a synchronous call while nothing is in flight, then the plain arithmetic: another kernel variant, another session
    _same(al.align_batch([r.signal for r in data[0][0]], [r.sequence for r in data[0][0]], True), want[0])
    for t in (t0, t1, t2, t3, tz, ts, ttr):
        t.close()
    al.set_strict("off")
    plain = [al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True) for reads, _ in data[:2]]
    tickets = [al.align_async(*packed, True) for _, packed in data[:2]]
    for t, w in zip(tickets, plain):
        _same(t.wait(), w)
        assert t.timing()["reads_strict"] == 0
        t.close()
    s = al.session_stats()
    assert s["aborted"] == 0 and s["sessions"] >= 2
    al.close()


def test_no_session_environment_switch(models, monkeypatch):
    monkeypatch.setenv("DYN_NO_SESSION", "1")
    al = Aligner(models["syn9"], "rna004", device=0)
    data = _data(models, 3, 600, 4500)
    want = [al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True) for reads, _ in data]
    tickets = [al.align_async(*packed, True) for _, packed in data]
    for t, w in zip(tickets, want):
        _same(t.wait(), w)
        assert t.timing()["launches"] == 1
        t.close()
    assert al.session_stats()["sessions"] == 0
    al.close()


def test_recycled_tickets_of_equal_size_with_long_copies(models):
    """Round 5 regression: a ticket's completion counter lives in a recycled device buffer and is cleared on the copy-in
    stream BEHIND the ticket's copies; the host's second opinion read it before that -- and found the full count of the
    buffer's last ticket (the same number of reads), so the per-segment kernels ran on a ticket whose reads had not started
    (a GPU memory fault with 4 096-read batches in the plain arithmetic, where the host is quickest). Every ticket must
    be the one-launch-per-batch result, however long its copies take."""
    _, mean, sd = synth.read_model_file(models["syn9"])
    reads = synth.make_reads(4700, 4096, "rna004", mean, sd, 2000)
    packed = synth.pack_reads(reads)
    al = Aligner(models["syn9"], "rna004", device=0)
    al.set_strict("off")  # no tie search on the host: the publish follows the copies at once
    want = al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True)
    for rnd in range(3):  # from the second round on every buffer is a recycled one
        tickets = [al.align_async(*packed, True) for _ in range(4)]
        for t in tickets:
            _same(t.wait(), want)
            assert t.timing()["launches"] == 0
            t.close()
    assert al.session_stats()["aborted"] == 0
    al.close()


def test_a_ticket_published_to_waves_that_have_left_is_published_again(models, monkeypatch):
    """The idle watchdog (DYN_SESSION_IDLE_S, 20 s by default; 5 ms here): resident waves that find nothing to do while the
    host still holds the session open raise the abort word and leave -- a front stage that takes longer than that (here:
    staging 4 096 reads behind a ticket of 600 tiny ones) then publishes its ticket to waves that are gone. The back thread
    sees the abort word, waits for the kernel's end and publishes the ticket again; the results are the synchronous call's."""
    monkeypatch.setenv("DYN_SESSION_IDLE_S", "0.005")
    _, mean, sd = synth.read_model_file(models["syn9"])
    tiny = synth.make_reads(4800, 599, "rna004", mean, sd, (30, 40))
    # one long read among them: the session's arenas are sized by it, so the big ticket JOINS this session (a session whose
    # arenas are too small is closed in front of the ticket, and nothing is lost); the wave that takes it is busy for tens of ms
    # while every other wave idles into the watchdog
    tiny += synth.make_reads(4802, 1, "rna004", mean, sd, 2600)
    big = synth.make_reads(4801, 4096, "rna004", mean, sd, 2000)
    al = Aligner(models["syn9"], "rna004", device=0)
    want_tiny = al.align_batch([r.signal for r in tiny], [r.sequence for r in tiny], True)
    want_big = al.align_batch([r.signal for r in big], [r.sequence for r in big], True)
    p_tiny, p_big = synth.pack_reads(tiny), synth.pack_reads(big)
    for rnd in range(2):
        t0 = al.align_async(*p_tiny, True)   # opens a session; its reads are done within a millisecond
        t1 = al.align_async(*p_big, True)    # in flight (the session stays open) but tens of ms away from its publish
        _same(t0.wait(), want_tiny)
        _same(t1.wait(), want_big)
        assert t1.timing()["launches"] == 0
        t0.close(), t1.close()
    s = al.session_stats()
    assert s["aborted"] >= 1 and s["republished"] >= 1, s
    assert s["reads"] >= 2 * (600 + 4096)  # (the reads of a ticket that was published twice count twice)
    al.close()


@pytest.mark.parametrize("layout", ["inplace", "separate"])
def test_page_starved_tickets_share_the_pool_of_a_paged_session(models, monkeypatch, layout):
    """A memory budget too small for an arena per resident wave (reads of 100 k samples; here: dyn_aligner_set_mem_budget):
    the session shares the pool's pages through the free list as the one-launch-per-batch kernel does -- waves keep what they
    hold, take more only while holding none, idle waves hand back what somebody waits for -- with the posteriors in place or
    separate. Results: the one-launch-per-batch results bit for bit, over tickets of different read lengths."""
    monkeypatch.setenv("DYN_FORCE_LAYOUT", layout)
    al = Aligner(models["syn9"], "rna004", device=0)
    al.set_mem_budget(3 << 30)   # 1 024 waves x ~30 MB of arena do not fit; the longest read needs ~30 MB
    data = _data(models, 5, 640, 4900, bases=(100, 420)) + _data(models, 1, 700, 4990, bases=(30, 120))
    want = [al.align_batch([r.signal for r in reads], [r.sequence for r in reads], True) for reads, _ in data]
    for rnd in range(2):
        tickets = [al.align_async(*packed, True) for _, packed in data]
        for t, w in zip(tickets, want):
            _same(t.wait(), w)
            tm = t.timing()
            assert tm["launches"] == 0 and tm["lp_inplace"] == (1 if layout == "inplace" else 0)
            assert tm["pool_pages"] < tm["n_waves"] * 8   # nowhere near an arena per wave
            t.close()
    s = al.session_stats()
    assert s["aborted"] == 0 and s["sessions"] >= 1 and s["tickets"] == 12
    # the waves' own account of their idle share (dyn_aligner_session_idle_split): the parts fit inside the wholes
    assert 0 < s["wave_cycles_pages"] <= s["wave_cycles_idle"]
    assert s["wave_cycles_before_first_read_pages"] <= s["wave_cycles_before_first_read"] <= s["wave_cycles_idle"]
    assert s["wave_cycles_before_first_read_pages"] <= s["wave_cycles_pages"]
    not_busy_not_idle = s["wave_cycles_life"] - s["wave_cycles_busy"] - s["wave_cycles_idle"]
    assert 0 < s["wave_cycles_last_turn"] <= not_busy_not_idle
    assert s["wave_cycles_longest_last_turn"] * s["waves"] >= s["wave_cycles_last_turn"]   # (summed over the sessions alike)
    al.close()


def test_getters_of_a_completed_ticket_do_not_wait_for_the_session(models):
    """dyn_batch_fetch / dyn_batch_signals copied with a null-stream hipMemcpy, which waits for the resident
    session kernel that LATER tickets keep open -- a caller fetching from ticket k with k+1.. in flight was held until the
    pipeline ran dry. The getters copy on the handle's own non-blocking stream now: fetching from the first ticket returns
    while the others are still being worked on, the rows are the ticket's, and the session is neither aborted nor drained
    (all tickets in ONE session)."""
    import time

    from dynamont_amd._dynamont import AlignBatchResult
    _, mean, sd = synth.read_model_file(models["syn9"])
    big = synth.make_reads(4801, 1024, "rna004", mean, sd, 2000)
    p_big = synth.pack_reads(big)
    al = Aligner(models["syn9"], "rna004", device=0)
    want = al.align_batch([r.signal for r in big], [r.sequence for r in big], True)
    al.align_async(*p_big, True).close()            # warm: pool, buffers
    s0 = al.session_stats()
    first = al.align_async(*p_big, True)
    rest = [al.align_async(*p_big, True) for _ in range(8)]   # ~0.3 s of GPU work behind the first ticket (same arena size: one session)
    first.wait()
    t0 = time.perf_counter()
    again = AlignBatchResult(want.n, int(want.seg_offsets[-1]))
    import ctypes
    rc = al._L.dyn_batch_fetch(first._h, ctypes.byref(again._c))
    t_fetch = time.perf_counter() - t0
    assert rc == 0
    t0 = time.perf_counter()
    for t in rest:
        t.wait()
    t_rest = time.perf_counter() - t0
    _same(again, want)
    _same(first.result, want)
    assert t_fetch < t_rest, (t_fetch, t_rest)     # (a null-stream copy returned when the LAST ticket's session ended: t_rest ~ 0)
    for t in [first] + rest:
        t.close()
    s1 = al.session_stats()
    assert s1["aborted"] == s0["aborted"] == 0 and s1["sessions"] - s0["sessions"] == 1 and s1["tickets"] - s0["tickets"] == 9
    al.close()


def test_successor_of_a_handle_that_parked_one_huge_array(models):
    """A handle that served page-starved batches in place (reads of 100 k samples: ONE array of most of the HBM) parks it; its
    successor needs three arrays (an arena per wave, separate posteriors) and takes the huge one over as its first -- the other
    two then do not fit. ensure_pool releases what the handle holds of the pool and allocates the three at their own sizes
    (bench.py met this between its cfg3 record and the CLI's handle: HIP out of memory)."""
    import dynamont_amd
    dynamont_amd.release_cached_memory()
    _, mean, sd = synth.read_model_file(models["syn9"])
    long_reads = synth.make_reads(4900, 1100, "dna_r10_400bps", mean, sd, (7000, 8000))   # ~90 k-100 k samples each
    al = Aligner(models["syn9"], "dna_r10_400bps", device=0)
    t = al.align_async(*synth.pack_reads(long_reads), True)
    r = t.wait()
    assert (r.status == 0).all()
    tm = t.timing()
    t.close()
    big_pool_gb = tm["pool_pages"] * tm["page_rows"] * 448 * 8 / 1e9
    al.close()                                    # parks the pool
    assert tm["lp_inplace"] == 1 and big_pool_gb > 180, (tm["lp_inplace"], big_pool_gb)
    reads = synth.make_reads(4901, 1024, "rna004", mean, sd, 2000)
    al2 = Aligner(models["syn9"], "rna004", device=0)
    want = al2.align_batch([x.signal for x in reads[:64]], [x.sequence for x in reads[:64]], True)
    t2 = al2.align_async(*synth.pack_reads(reads), True)
    got = t2.wait()
    assert (got.status == 0).all() and t2.timing()["launches"] == 0   # the resident queue: an arena per wave, three arrays
    for i in range(64):
        lo, hi = int(want.seg_offsets[i]), int(want.seg_offsets[i]) + int(want.n_segments[i])
        glo = int(got.seg_offsets[i])
        assert np.array_equal(got.signal_positions[glo:glo + hi - lo], want.signal_positions[lo:hi])
    t2.close()
    al2.close()
    dynamont_amd.release_cached_memory()
