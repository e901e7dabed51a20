"""CPU: the harness rows P1-P5 -- counterparts of the reference's
segmentation/{utils,segment,train}.py -- against (a) the golden bytes produced by the
reference's own functions (tests/golden/g6_harness.npz) and (b) the known-answer vectors of the
reference's own tests, restated here as data (tests/test_segment.py:179-200,154-157,
tests/test_utils.py:7-128, tests/test_managed_list.py of the reference)."""
import io
import os
import queue
import sys

import numpy as np
import pytest

from conftest import golden, model_for
from dynamont_amd import synth, zstd_io
from dynamont_amd.segmentation import segment as seg
from dynamont_amd.segmentation import train as trn
from dynamont_amd.segmentation import utils as U
from oracle.pyoracle import Oracle

pytestmark = pytest.mark.usefixtures("native_lib")   # BAM basecalls are read by the library (built on demand; no compute call)


def hampel_loop(x, W=3, ns=3.0):
    """Explicit sliding-window restatement (the shape of the reference test's inline checker)."""
    x = np.asarray(x, dtype=np.float64)
    out = x.copy()
    if x.size <= W:
        return out
    n = x.size - W - (W % 2 == 0)
    for i in range(n):
        w = x[i:i + W]
        med = np.median(w)
        mad = np.median(np.abs(w - med))
        c = W // 2 + i
        if abs(x[c] - med) > ns * 1.4826 * mad:
            out[c] = med
    return out


def test_hampel_known_answers_from_reference_tests():
    a = np.array([1, 1, 1, 10, 1, 1, 1], dtype=float)
    U.hampel(a)
    assert np.array_equal(a, np.ones(7))                      # reference tests/test_utils.py:7-14
    b = np.array([1, 1, 50, 1, 1, 1, 75, 1], dtype=float)     # reference tests/test_segment.py:179-200
    want = hampel_loop(b)
    U.hampel(b)
    assert np.array_equal(b, want)
    assert b[-2] == 75.0  # index len-2 is never filtered


def test_hampel_matches_reference_outputs():
    g = golden("g6_harness.npz")
    for i in range(int(g["n_hampel"])):
        x = g[f"h{i}_in"].copy()
        U.hampel(x, int(g[f"h{i}_W"]), float(g[f"h{i}_ns"]))
        assert np.array_equal(x, g[f"h{i}_out"]), i
        assert np.array_equal(hampel_loop(g[f"h{i}_in"], int(g[f"h{i}_W"]), float(g[f"h{i}_ns"])), g[f"h{i}_out"])


def test_segmentation_to_string_bytes_match_reference(models, oracle_built):
    g = golden("g6_harness.npz")
    for i in range(int(g["n_csv"])):
        p = f"csv{i}_"
        pore = str(g[p + "pore"])
        pid, rna, k = synth.PORES[pore]
        seq = str(g[p + "sequence"])
        res = Oracle(model_for(models, pore), pid).align(g[p + "signal"], seq, True)
        res["polishes"] = [""] * len(res["states"])
        start = int(g[p + "start"])
        got = U.segmentation_to_string(res, f"read-{i}", f"sig-{i}", start, len(g[p + "signal"]) + start, seq, k, rna)
        assert got == g[p + "bytes"].tobytes()
        first = got.split(b"\n")[0].split(b",")
        assert first[2] == str(start).encode() and first[7] == b"M" and first[9] == b"NA"


def test_model_io_and_small_helpers(tmp_path):
    m = {"AAAAA": (0.5, 0.25), "AAAAC": (1.25, 0.125)}
    f = tmp_path / "m.model"
    U.write_kmer_model(str(f), m)
    assert f.read_text() == "kmer\tlevel_mean\tlevel_stdv\nAAAAA\t0.5\t0.25\nAAAAC\t1.25\t0.125\n"
    assert U.read_kmer_model(str(f)) == m
    assert U.cnt_nts("AACGTT") == {"A": 2, "C": 1, "G": 1, "T": 2}
    assert U.cnt_nts_ratios("AACG") == {"A": 0.5, "C": 0.25, "G": 0.25, "T": 0.0}
    e = U.SegmentationError("r1")
    assert str(e) == "No segmentation calculated for r1" and e.read == "r1"
    assert U.get_model("rna004").endswith("models/rna/rna004/rna004_9mer.model")
    assert U.get_model("/x/custom.model").endswith("/x/custom.model")
    assert U.kmer_of_code(1, 5, False) == "AAAAC" and U.kmer_of_code(1, 5, True) == "CAAAA"


def test_managed_list_semantics():
    ml = trn.ManagedList([1.0], max_size=3)
    for v in (2.0, 3.0, 4.0):
        ml.add(v)
    assert ml.get_list() == [2.0, 3.0, 4.0] and ml.mean() == 3.0 and ml.median() == 3.0
    assert repr(ml) == "ManagedList([2.0, 3.0, 4.0])"
    empty = trn.ManagedList([])
    assert empty.mean() is None and empty.median() is None
    tab = trn.ManagedTable(np.array([1.0, 2.0]), np.array([0.1, 0.2]), max_size=2)
    tab.add(np.array([3.0, 2.0]), np.array([0.3, 0.2]))
    m, s = tab.mean()
    assert np.allclose(m, [2.0, 2.0]) and np.allclose(s, [0.2, 0.2])
    tab.add(np.array([5.0, 2.0]), np.array([0.5, 0.2]))   # evicts the initial entry
    assert np.allclose(tab.mean()[0], [4.0, 2.0])


def test_cli_surfaces():
    a = seg.parse(["-r", "raw", "-b", "x.bam", "-o", "out", "--mode", "basic", "-p", "rna004"])
    assert a.processes > 0 and a.qscore == 0.0 and a.model_path is None and a.mode == "basic"
    with pytest.raises(SystemExit):
        seg.parse(["-r", "raw", "-b", "x.bam", "-o", "out", "--mode", "fancy", "-p", "rna004"])
    t = trn.parse(["-r", "raw", "-b", "x.bam", "-o", "out", "-p", "rna002"])
    assert t.batch_size == 24 and t.epochs == 1 and t.qscore == 10.0 and t.max_batches is None


class FakeReader:
    def __init__(self, name, fail_close=False):
        self.name, self.closed, self.fail_close = name, False, fail_close

    def close(self):
        self.closed = True
        if self.fail_close:
            raise RuntimeError("close failed")


def test_raw_cache_lru(monkeypatch):
    opened = []

    def fake_open(path):
        r = FakeReader(path, fail_close=(path == "b"))
        opened.append(r)
        return r

    monkeypatch.setattr(seg, "open_pod5", fake_open)
    monkeypatch.setattr(seg, "RAW_CACHE", None)
    monkeypatch.setattr(seg, "RAW_CACHE_SIZE", 2)
    ra = seg.get_raw("a")
    assert seg.get_raw("a") is ra and len(opened) == 1           # hit
    rb = seg.get_raw("b")
    seg.get_raw("a")                                              # move-to-end: b is now oldest
    seg.get_raw("c")                                              # evicts b; its close error is swallowed
    assert rb.closed and not ra.closed and list(seg.RAW_CACHE) == ["a", "c"]
    seg.close_raw_cache()
    assert ra.closed and len(seg.RAW_CACHE) == 0


def test_listener_protocol_and_zstd(tmp_path):
    out = tmp_path / "results.csv.zst"
    q = queue.Queue()
    for item in (b"r1,s1,0,5,2,A,ACG,M,0.500000,NA\n", "error: native, Signal is empty\tT: 0\tN: 9\tRid: r2\tSid: s2",
                 b"r3,s3,1,2,3,C,CCC,M,1.000000,NA\n", "kill"):
        q.put(item)
    seg.listener(q, str(out))
    data = zstd_io.decompress(out.read_bytes())
    assert data == (b"readid,signalid,start,end,basepos,base,motif,state,posterior_probability,polish\n"
                    b"r1,s1,0,5,2,A,ACG,M,0.500000,NA\nr3,s3,1,2,3,C,CCC,M,1.000000,NA\n")
    assert (tmp_path / "results.errors").read_text() == "error: native, Signal is empty\tT: 0\tN: 9\tRid: r2\tSid: s2\n"
    assert out.read_bytes()[:4] == b"\x28\xb5\x2f\xfd"  # zstd frame magic


def test_zstd_roundtrip_large():
    rng = np.random.default_rng(0)
    payload = b"".join(b"%d,%f\n" % (i, x) for i, x in enumerate(rng.standard_normal(20000)))
    buf = io.BytesIO()
    with zstd_io.ZstdWriter(buf, level=3) as w:
        for i in range(0, len(payload), 7777):
            w.write(payload[i:i + 7777])
    assert zstd_io.decompress(buf.getvalue()) == payload and len(buf.getvalue()) < len(payload)


@pytest.mark.parametrize("threads,chunk", [(1, 4 << 20), (3, 5000), (8, 1 << 16)])
def test_zstd_parallel_frames_roundtrip(threads, chunk):
    """The CLI writer: chunks compressed concurrently, written in order as independent frames."""
    rng = np.random.default_rng(1)
    payload = b"".join(b"r%d,%d,%f\n" % (i, i * 7, x) for i, x in enumerate(rng.standard_normal(30000)))
    buf = io.BytesIO()
    with zstd_io.ParallelZstdWriter(buf, level=3, threads=threads, chunk_bytes=chunk) as w:
        w.write(b"")                                  # ignored
        w.write(payload[:10])                         # small writes are coalesced
        for i in range(10, len(payload), 123457):
            w.write(memoryview(payload)[i:i + 123457])
    out = buf.getvalue()
    assert out[:4] == b"\x28\xb5\x2f\xfd" and zstd_io.decompress(out) == payload and len(out) < len(payload)
    empty = io.BytesIO()
    zstd_io.open_writer(empty).close()                # no rows at all: still one valid (empty) frame
    assert empty.getvalue()[:4] == b"\x28\xb5\x2f\xfd" and zstd_io.decompress(empty.getvalue()) == b""


def _one_shot_decompress(data: bytes, size: int) -> bytes:
    """libzstd's single-call decoder (a different code path from the streaming one zstd_io.decompress uses); it
    decodes exactly ONE frame and fails on anything malformed in it."""
    import ctypes as C
    L = zstd_io._libzstd()
    L.ZSTD_decompress.restype = C.c_size_t
    L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    dst = C.create_string_buffer(max(size, 1))
    rc = L.ZSTD_decompress(dst, size, data, len(data))
    assert not L.ZSTD_isError(rc), L.ZSTD_getErrorName(rc)
    return dst.raw[:rc]


@pytest.mark.parametrize("kind", ["csv", "repeats", "constant", "random", "tiny", "empty"])
@pytest.mark.parametrize("threads,chunk", [(1, 4 << 20), (4, 70001), (8, 1 << 17)])
def test_zstd_single_frame_jobs(kind, threads, chunk):
    """The default CLI writer: jobs compressed concurrently by contexts of their own and stitched into ONE frame
    (job 0's header, rep-codes invalidated at every later job's start, an empty last block). Payloads chosen so that
    the stitched blocks are compressed (with and without repeat offsets crossing a job boundary), RLE and raw."""
    rng = np.random.default_rng(3)
    payload = {
        "csv": lambda: b"".join(b"r%d,%d,%f\n" % (i, i * 7, x) for i, x in enumerate(rng.standard_normal(40000))),
        "repeats": lambda: (b"read-0001,12345,12399,77,M,0.999871\n" * 7 + b"read-0001,12399,12405,78,M,0.5\n") * 6000,
        "constant": lambda: b"\0" * 900000,
        "random": lambda: rng.integers(0, 256, 700000, dtype=np.uint8).tobytes(),
        "tiny": lambda: b"x",
        "empty": lambda: b"",
    }[kind]()
    buf = io.BytesIO()
    with zstd_io.ParallelZstdWriter(buf, level=3, threads=threads, chunk_bytes=chunk, single_frame=True) as w:
        for i in range(0, len(payload), 99991):
            w.write(payload[i:i + 99991])
    out = buf.getvalue()
    assert zstd_io.count_frames(out) == 1
    assert zstd_io.decompress(out) == payload
    assert _one_shot_decompress(out, len(payload)) == payload
    if kind in ("csv", "repeats", "constant"):
        assert len(out) < len(payload) // 2


def test_cli_writer_is_one_frame_by_default(tmp_path):
    """python-zstandard's default readers stop after the first frame, so the CLI must write ONE frame
    like the reference (segment.py:74-79) unless the user opts into parallel frames."""
    rng = np.random.default_rng(2)
    payload = b"".join(b"r%d,%d,%f\n" % (i, i * 7, x) for i, x in enumerate(rng.standard_normal(600000)))
    assert len(payload) > 3 * (4 << 20)               # several chunks of the parallel writer
    one, many = io.BytesIO(), io.BytesIO()
    with zstd_io.open_writer(one) as w:
        for i in range(0, len(payload), 1 << 20):
            w.write(payload[i:i + (1 << 20)])
    with zstd_io.open_writer(many, parallel_frames=True) as w:
        w.write(payload)
    assert zstd_io.count_frames(one.getvalue()) == 1
    assert zstd_io.count_frames(many.getvalue()) > 1
    assert zstd_io.decompress(one.getvalue()) == payload and zstd_io.decompress(many.getvalue()) == payload
    # the listener (what dynamont-resquiggle runs) uses the default
    out = tmp_path / "big.csv.zst"
    q = queue.Queue()
    q.put(payload)
    q.put("kill")
    seg.listener(q, str(out))
    assert zstd_io.count_frames(out.read_bytes()) == 1


def test_cli_output_reads_with_zstandard_defaults(tmp_path):
    zstandard = pytest.importorskip("zstandard")      # not in the build image; runs wherever it is installed
    payload = b"".join(b"r%d,%d\n" % (i, i * 7) for i in range(1500000))
    out = tmp_path / "big.csv.zst"
    q = queue.Queue()
    q.put(payload)
    q.put("kill")
    seg.listener(q, str(out))
    with zstandard.ZstdDecompressor().stream_reader(io.BytesIO(out.read_bytes())) as r:
        assert r.read() == seg.CSV_HEADER + payload


@pytest.mark.parametrize("pore", ["rna004", "dna_r10_400bps"])
def test_job_generation_and_preprocessing(models, tmp_path, pore):
    _, mean, sd = synth.read_model_file(model_for(models, pore))
    reads = synth.make_reads(31, 5, pore, mean, sd, (40, 90))
    raw, bam, expected = synth.write_dataset(str(tmp_path), "ds", reads, pore, seed=3)
    jobs = list(seg.generate_jobs(str(tmp_path), bam, 0))
    assert len(jobs) == 5 and all(j[0] == raw for j in jobs)
    assert all(j[3] == 37 for j in jobs)                       # start = sp + ts
    rna = synth.PORES[pore][1]
    for job, (x, seq) in zip(jobs, expected):
        signal, read = seg.prepare_job(job, rna)
        assert signal.dtype == np.float64 and len(signal) == job[4] - job[3]
        assert np.array_equal(signal, hampel_loop(x))          # -= shift, /= scale, Hampel(3, 3 sigma)
        assert read == seq
        if rna:
            assert read.startswith("AAAAAAAAA")
    # quality filter
    qs = [float(l.split("\t")[2]) for l in open(bam).read().splitlines()[1:]]
    kept = list(seg.generate_jobs(str(tmp_path), bam, 12.0))
    assert len(kept) == sum(q >= 12.0 for q in qs)
    seg.close_raw_cache()


def _batch_result_from_dicts(results):
    """Pack per-read result dicts (oracle output) into the columnar AlignBatchResult of the C ABI."""
    from dynamont_amd._dynamont import AlignBatchResult
    n = len(results)
    cap = sum(len(r["sequence_positions"]) for r in results if r is not None)
    out = AlignBatchResult(n, max(cap, 1))
    off = 0
    for i, r in enumerate(results):
        out.seg_offsets[i] = off
        if r is None:
            out.status[i] = 4
            continue
        k = len(r["sequence_positions"])
        out.n_segments[i] = k
        out.sequence_positions[off:off + k] = r["sequence_positions"]
        out.signal_positions[off:off + k] = r["signal_positions"]
        out.probabilities[off:off + k] = r["probabilities"]
        out.states[off:off + k] = ord("M")
        out.Z[i] = r["Z"]
        off += k
    out.seg_offsets[n] = off
    return out


def test_native_csv_formatter_matches_reference_bytes(models, oracle_built, native_lib):
    """dyn_format_csv (C++) == the reference's segmentation_to_string bytes (golden G6), incl. a failed read."""
    from dynamont_amd import Aligner
    from dynamont_amd._dynamont import format_csv
    g = golden("g6_harness.npz")
    for i in range(int(g["n_csv"])):
        p = f"csv{i}_"
        pore = str(g[p + "pore"])
        pid, rna, k = synth.PORES[pore]
        seq = str(g[p + "sequence"])
        res = Oracle(model_for(models, pore), pid).align(g[p + "signal"], seq, True)
        start = int(g[p + "start"])
        al = Aligner(model_for(models, pore), pore, device="host")
        batch = _batch_result_from_dicts([res, None, res])
        buf, begin, end = format_csv(al, batch, [seq, "ACGT", seq], [f"read-{i}", "bad", f"read-{i}"],
                                     [f"sig-{i}", "bad", f"sig-{i}"], [start, 0, start],
                                     [len(g[p + "signal"]) + start, 0, len(g[p + "signal"]) + start], threads=3)
        want = g[p + "bytes"].tobytes()
        assert buf[int(begin[0]):int(end[0])].tobytes() == want and buf[int(begin[2]):int(end[2])].tobytes() == want
        assert begin[1] == end[1]


def test_native_probability_formatting_equals_python(models, native_lib):
    """f"{p:.6f}" for awkward doubles: ties, values next to ties, tiny, exactly 1, slightly above 1."""
    from dynamont_amd import Aligner
    from dynamont_amd._dynamont import format_csv
    rng = np.random.default_rng(12)
    base = np.concatenate([rng.uniform(0, 1, 20000), 10.0 ** rng.uniform(-12, 0, 5000),
                           (rng.integers(0, 10 ** 6, 5000) + 0.5) / 1e6,                      # decimal ties
                           np.nextafter((rng.integers(0, 10 ** 6, 3000) + 0.5) / 1e6, 0),
                           np.nextafter((rng.integers(0, 10 ** 6, 3000) + 0.5) / 1e6, 1),
                           [0.0, 1.0, 1.0000000000004, 0.9999995, 0.99999949999, 5e-7, 4.999999e-7, 1e-300, 0.5, 0.125]])
    n = len(base)
    al = Aligner(models["syn5"], "dna_r9", device="host")
    seq = "ACGTACGTAC" * ((n + 20) // 10)
    res = dict(Z=0.0, sequence_positions=np.arange(n, dtype=np.uint64) + 2, signal_positions=np.arange(n, dtype=np.uint64) * 3,
               probabilities=base)
    buf, begin, end = format_csv(al, _batch_result_from_dicts([res]), [seq], ["r"], ["s"], [7], [3 * n + 7], threads=1)
    data = buf[int(begin[0]):int(end[0])].tobytes()
    got = [line.split(",")[8] for line in data.decode().splitlines()]
    assert got == [f"{float(p):.6f}" for p in base]
    res["states"] = ["M"] * n
    res["polishes"] = [""] * n
    assert data == U.segmentation_to_string(res, "r", "s", 7, 3 * n + 7, seq, 5, False)


def test_sam_and_bam_readers_round_trip(tmp_path):
    """The vendor-free SAM/BAM parsers deliver name, sequence and the eight tags segment.py reads."""
    from dynamont_amd import bam_io
    from dynamont_amd.pod5_io import iter_basecalls
    rng = np.random.default_rng(1)
    recs = []
    for i in range(300):   # > one 64 KB BGZF block
        seq = "".join(rng.choice(list("ACGTN"), size=int(rng.integers(1, 700))))
        tags = {"qs": float(np.float32(rng.uniform(5, 30))), "ns": int(rng.integers(1000, 10 ** 6)), "ts": int(rng.integers(0, 500)),
                "fn": f"file_{i % 3}.pod5", "sm": float(np.float32(rng.uniform(50, 120))), "sd": float(np.float32(rng.uniform(5, 30)))}
        if i % 4 == 0:
            tags["pi"] = f"parent-{i}"
            tags["sp"] = int(rng.integers(0, 10 ** 5))
        recs.append((f"read-{i:04d}", seq, tags))
    bam, sam = str(tmp_path / "x.bam"), str(tmp_path / "x.sam")
    bam_io.write_bam(bam, recs)
    bam_io.write_sam(sam, recs)
    import gzip
    assert gzip.open(bam).read(4) == b"BAM\x01"
    for path in (bam, sam):
        got = list(iter_basecalls(path))
        assert len(got) == len(recs)
        for r, (name, seq, tags) in zip(got, recs):
            assert r.query_name == name and r.query_sequence == seq
            assert r.has_tag("pi") == ("pi" in tags) and r.has_tag("f5") is False
            for k, v in tags.items():
                assert r.get_tag(k) == v, (path, k)
    # a SAM `f` field written with more digits than a float32 holds still reads as the float32 the BAM carries
    sam2 = str(tmp_path / "y.sam")
    with open(sam2, "w") as w:
        w.write("@HD\tVN:1.6\tSO:unknown\nr\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\t*\tsm:f:91.23456789012345\tsd:f:15.1\tqs:f:12\tns:i:9\tts:i:1\tfn:Z:a.pod5\n")
    (rec,) = list(iter_basecalls(sam2))
    assert rec.get_tag("sm") == float(np.float32(91.23456789012345)) and rec.get_tag("sd") == float(np.float32(15.1))
    # the BAM reader streams: the first record is out before the rest of the file has been touched
    it = bam_io.iter_bam(bam)
    assert next(it).query_name == "read-0000"
    it.close()
    # and the job generator consumes them exactly like the TSV container
    jobs = list(seg.generate_jobs("/data", bam, 10.0))
    assert len(jobs) == sum(t["qs"] >= 10.0 for _, _, t in recs)
    name, seq, tags = next(x for x in recs if x[2]["qs"] >= 10.0)
    j = jobs[0]
    assert j[0] == "/data/" + tags["fn"] and j[5] == seq and j[6] == name
    assert j[7] == tags.get("pi", name) and j[3] == tags.get("sp", 0) + tags["ts"] and j[4] == tags.get("sp", 0) + tags["ns"]


def test_native_csv_sink_frames_header_and_error_lines(native_lib, tmp_path):
    """csv_sink.cpp without a GPU: an empty run is one zstd frame that holds the CSV header (segment.py:80); error lines
    handed in by the caller land in `.errors`, one per line (the batches themselves are covered on the GPU box,
    tests/test_gpu_harness.py compares the CLI's bytes with the oracle pipeline's)."""
    import ctypes as C
    from dynamont_amd import zstd_io
    out, errs = str(tmp_path / "o.csv.zst"), str(tmp_path / "o.errors")
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    assert native_lib.dyn_csv_sink_open(out.encode(), errs.encode(), 3, 4, C.byref(h), err, 512) == 0, err.value
    assert native_lib.dyn_csv_sink_error_line(h, b"error: worker, boom\tN: 12\tRid: r1\tSid: s1") == 0
    assert native_lib.dyn_csv_sink_error_line(h, b"error: worker, bang\tN: 7\tRid: r2\tSid: s2") == 0
    csv, zst, nerr = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert native_lib.dyn_csv_sink_close(h, C.byref(csv), C.byref(zst), C.byref(nerr), err, 512) == 0, err.value
    data = open(out, "rb").read()
    assert zstd_io.count_frames(data) == 1 and len(data) == zst.value
    assert zstd_io.decompress(data) == b"readid,signalid,start,end,basepos,base,motif,state,posterior_probability,polish\n"
    assert csv.value == 80 and nerr.value == 2
    assert open(errs).read().splitlines() == ["error: worker, boom\tN: 12\tRid: r1\tSid: s1", "error: worker, bang\tN: 7\tRid: r2\tSid: s2"]


def test_console_scripts_carry_the_reference_names_and_resolve(capsys):
    """pyproject.toml declares `dynamont-resquiggle` / `dynamont-train` like /root/reference/pyproject.toml:46-48 does; the
    entry points import and answer --help (no GPU, no library call)."""
    import importlib

    import pytest
    import tomli
    from conftest import ROOT
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as f:
        scripts = tomli.load(f)["project"]["scripts"]
    assert set(scripts) == {"dynamont-resquiggle", "dynamont-train"}
    for name, target in scripts.items():
        module, func = target.split(":")
        main = getattr(importlib.import_module(module), func)
        assert callable(main)
        with pytest.raises(SystemExit) as e:
            main(["--help"])
        assert e.value.code == 0
        text = capsys.readouterr().out
        assert "--raw" in text and "--basecalls" in text and "--pore" in text


def test_bench_contract_without_a_gpu():
    """bench.py: the driver's flags parse, the N = 1 default is BASELINE configs[1], every workload it names exists in the
    generator, and without a GPU it refuses loudly instead of measuring something else (no CPU path to fall back to)."""
    import importlib.util
    import subprocess

    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"]
        a = bench.parse()
    finally:
        sys.argv = old
    assert (a.gpus, a.steps, a.warmup, a.mode, a.strict) == (1, 20, 5, "align", "ties") and a.workload is None
    for name, (cfgname, _per_batch, _n) in bench.WORKLOADS.items():
        assert cfgname in synth.CONFIGS, name
    assert synth.CONFIGS["cfg2"] == dict(pore="rna004", n_reads=1024, n_bases=2000, seed=2)
    assert bench.KBWD_BYTES_PER_CELL + bench.KFWD_BYTES_PER_CELL == 20.125 and bench.HBM_PEAK_GBPS == 8000.0
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-e2e", "--reads", "4", "--batches", "1"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout) and not [l for l in r.stdout.splitlines() if l.startswith("{")]
