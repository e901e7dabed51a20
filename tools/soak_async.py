#!/usr/bin/env python
"""GPU box: a few hundred tickets of mixed kind through ONE handle's asynchronous pipeline -- align with and without
probabilities, train (with pooled statistics), batches of 1 .. 400 reads of very different length with failed reads among
them, a random number of tickets in flight (so that some merge into one launch and some run alone) -- and every ticket's
result compared, bit for bit, with the synchronous call of a second handle on the same input. What `pytest -m gpu` checks
once per feature meets here in random order on recycled buffers.

    python tools/soak_async.py [tickets] [seed] [resident|watchdog|paged]

`resident` (round 5): batches of up to 1 100 reads as well, so that tickets open, join, outgrow and close sessions of the
RESIDENT read queue (sessions alternate with one-launch-per-batch jobs -- Z-only, training -- on the same lattice pool) -- the
results must still be the synchronous calls' bit for bit.
`paged`: `resident` with a memory budget of 2 GiB on both handles: no arena per wave fits, the sessions share the pool's pages
through the free list (posteriors in place), page-starved one-launch-per-batch jobs in between.
`watchdog`: the same with the idle watchdog at 3 ms (DYN_SESSION_IDLE_S): sessions abort under the tickets all the time, tickets
are published to waves that have left and published again (dyn_session_stats.republished) -- and nothing may change.
"""
import os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import numpy as np
from dynamont_amd import Aligner, synth

n_tickets = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WATCHDOG = len(sys.argv) > 3 and sys.argv[3] == "watchdog"
PAGED = len(sys.argv) > 3 and sys.argv[3] == "paged"
RESIDENT = len(sys.argv) > 3 and sys.argv[3] in ("resident", "watchdog", "paged")
if WATCHDOG:
    os.environ["DYN_SESSION_IDLE_S"] = "0.003"
d = tempfile.mkdtemp(prefix="dyn_soak_async_")
model = synth.write_model(os.path.join(d, "m9.model"), 9, seed=7, stdev=0.15)
_, mean, sd = synth.read_model_file(model)

pool = []
for k in range(24):
    n = int(rng.choice([3, 64, 400, 530, 700, 1100] if RESIDENT else [1, 3, 17, 64, 200, 400]))
    hi = int(rng.choice([120, 400, 900] if RESIDENT else [120, 600, 2500]))
    reads = synth.make_reads(5000 + k, n, "rna004", mean, sd, (30, hi), polya=(20, 100) if k % 3 == 0 else None)
    sigs, seqs = [r.signal for r in reads], [r.sequence for r in reads]
    for i in rng.choice(n, size=max(1, n // 12), replace=False):   # damage some reads
        kind = int(rng.integers(0, 4))
        if kind == 0 and len(seqs[i]) > 12:
            p = int(rng.integers(0, len(seqs[i]))); seqs[i] = seqs[i][:p] + "N" + seqs[i][p + 1:]
        elif kind == 1:
            sigs[i] = sigs[i][:max(0, len(seqs[i]) - 20)]
        elif kind == 2:
            seqs[i] = seqs[i][:int(rng.integers(1, 9))]
        else:
            sigs[i] = np.zeros(0)
    packed = synth.pack_reads([synth.SynthRead(np.asarray(s, dtype=np.float64), q) for s, q in zip(sigs, seqs)])
    pool.append((sigs, seqs, packed))

ref = Aligner(model, "rna004", device=0)
al = Aligner(model, "rna004", device=0)
if PAGED:
    ref.set_mem_budget(2 << 30)
    al.set_mem_budget(2 << 30)
want = {}
def reference(k, kind):
    if (k, kind) not in want:
        sigs, seqs, _ = pool[k]
        want[(k, kind)] = ref.train_batch(sigs, seqs, pooled=True) if kind == "train" else ref.align_batch(sigs, seqs, kind == "align")
    return want[(k, kind)]

ALIGN = ("Z", "status", "bad_char", "n_segments", "seg_offsets", "sequence_positions", "signal_positions", "probabilities")
TRAIN = ("Z", "status", "bad_char", "transitions", "em_offsets", "em_count", "em_code", "em_mean", "em_stdev", "em_weight", "em_sum", "em_sumsq",
         "trans_counts", "pooled")
def same(got, exp, kind):
    for f in (TRAIN if kind == "train" else ALIGN):
        a, b = getattr(got, f), getattr(exp, f)
        if f in ("sequence_positions", "signal_positions", "probabilities"):
            m = int(exp.seg_offsets[-1]); a, b = a[:m], b[:m]
        if f in ("em_code", "em_mean", "em_stdev", "em_weight", "em_sum", "em_sumsq"):
            m = int(exp.em_offsets[-1]); a, b = a[:m], b[:m]
        if PAGED and f == "probabilities":
            # a paged session keeps the posterior layout of the ticket that opened it, a launch chooses per batch: the in-place
            # layout rounds the M posterior to float where the separate one keeps it in double -- equal to 1e-6, not bit for bit
            if np.abs(np.asarray(a) - np.asarray(b)).max(initial=0.0) > 1e-6:
                return f
            continue
        if not np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)):
            return f
    return None

t0 = time.time()
inflight, done, merged, resident = [], 0, 0, 0
depth = int(rng.integers(1, 13))
for t in range(n_tickets):
    if t % 25 == 0:
        depth = int(rng.integers(1, 13))
    k = int(rng.integers(0, len(pool)))
    kind = str(rng.choice(["align", "align", "align", "z", "train"]))
    _, _, packed = pool[k]
    tk = al.train_async(*packed, pooled=True) if kind == "train" else al.align_async(*packed, kind == "align")
    inflight.append((tk, k, kind))
    while len(inflight) > depth or (t == n_tickets - 1 and inflight):
        tk, k, kind = inflight.pop(0)
        got = tk.wait()
        tm = tk.timing()
        resident += tm["launches"] == 0 and tm["reads_ok"] > 0
        merged += 0.0 < tm["launch_share"] < 1.0
        bad = same(got, reference(k, kind), kind)
        assert bad is None, f"ticket {done}: batch {k} ({len(pool[k][1])} reads), {kind}: field {bad} differs from the synchronous call"
        tk.close()
        done += 1
ss = al.session_stats()
assert WATCHDOG or ss["aborted"] == 0, ss
if WATCHDOG:
    print(f"watchdog at 3 ms: {ss['aborted']} sessions aborted, {ss['republished']} tickets published again")
print(f"soak done: {done} tickets ({merged} of them shared a launch, {resident} ran in the resident read queue: {ss['sessions']} sessions, "
      f"{ss['tickets']} tickets, wave occupancy {ss['wave_occupancy']:.3f}) in {time.time() - t0:.0f} s, every result bit-identical to the synchronous call"
      + (" (posteriors: to 1e-6 where the posterior layouts differ)" if PAGED else ""))
al.close(); ref.close()
