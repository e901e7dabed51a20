"""File-format parity of the vendor-free readers (reference: src/dynamont/pod5_io.py:1-16,
src/dynamont/segmentation/segment.py:222-256).

What CAN be pinned without ONT's `pod5` package and htslib/pysam (neither is in the image, the reference's own tests mock
the reader and hold no file):
  * BAM: a file assembled HERE byte by byte from the SAM specification (section 4.2: BGZF blocks, the alignment record
    layout, 4-bit sequence packing, every tag type incl. `B` arrays, integer tags in the smallest type that fits -- which
    is how htslib, and therefore dorado, stores them) is read by ``bam_io.iter_bam`` and by a second, independent decoder
    written in this file straight from the same specification; both must agree field by field with what went in. The
    file is NOT produced by ``bam_io.write_bam``.
What needs the official tools (the `-m gpu` probes at the bottom): when `pod5` / `pysam` can be imported on the GPU box a
file written by the official writer is read with the vendor-free reader and the other way round; when they cannot, the
tests SKIP with that reason, so that the round's GPU test record shows that the formats remain unpinned against an
officially written file.
"""
import gzip
import os
import struct
import uuid
import zlib

import numpy as np
import pytest

from dynamont_amd import bam_io

pytestmark = pytest.mark.usefixtures("native_lib")   # BAM basecalls are read by the library (built on demand; no compute call)

SEQ_CODE = "=ACMGRSVTWYHKDBN"


# ---- assembling a BAM from the specification ------------------------------------------------------------------------
def spec_bgzf_block(payload: bytes) -> bytes:
    """SAM spec 4.1: a gzip member with FLG.FEXTRA, XLEN = 6 and the 'BC' subfield carrying BSIZE = total size - 1."""
    comp = zlib.compressobj(9, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    total = 12 + 6 + len(cdata) + 8
    head = bytes([31, 139, 8, 4]) + struct.pack("<I", 0) + bytes([0, 255]) + struct.pack("<H", 6)
    extra = b"BC" + struct.pack("<HH", 2, total - 1)
    return head + extra + cdata + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF)


def spec_int_tag(tag: str, v: int) -> bytes:
    """htslib's bam_aux_update_int / sam_parse: the smallest of c C s S i I that holds the value."""
    for code, fmt, lo, hi in (("C", "<B", 0, 255), ("c", "<b", -128, 127), ("S", "<H", 0, 65535), ("s", "<h", -32768, 32767),
                              ("I", "<I", 0, 2 ** 32 - 1), ("i", "<i", -2 ** 31, 2 ** 31 - 1)):
        if lo <= v <= hi:
            return tag.encode() + code.encode() + struct.pack(fmt, v)
    raise ValueError(v)


def spec_record(name, seq, qual, flag, cigar, tags: bytes, ref_id=-1, pos=-1) -> bytes:
    """SAM spec 4.2, one alignment: block_size | refID pos l_read_name mapq bin n_cigar_op flag l_seq next_refID
    next_pos tlen | read_name\\0 | cigar | seq (4 bit, high nibble first) | qual | tags"""
    nm = name.encode() + b"\0"
    codes = [SEQ_CODE.index(c) for c in seq]
    if len(codes) % 2:
        codes.append(0)
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    cig = b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(op)) for n, op in cigar)
    body = (struct.pack("<iiBBHHHiiii", ref_id, pos, len(nm), 60 if cigar else 0, 4680, len(cigar), flag, len(seq), -1, -1, 0)
            + nm + cig + packed + bytes(qual) + tags)
    return struct.pack("<i", len(body)) + body


def spec_bam(records, refs=()) -> bytes:
    text = b"@HD\tVN:1.6\tSO:unknown\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (n.encode(), ln) for n, ln in refs)
    out = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for n, ln in refs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    return out + b"".join(records)


# ---- a second decoder, straight from the specification (no code shared with bam_io) ------------------------------------
def spec_decode(raw: bytes):
    """BGZF by BSIZE (not by gzip's member detection), then the record layout. Returns [(name, seq, flag, n_cigar, tags)]."""
    data, p = b"", 0
    while p < len(raw):
        assert raw[p:p + 4] == bytes([31, 139, 8, 4])
        xlen, = struct.unpack_from("<H", raw, p + 10)
        q, bsize = p + 12, None
        while q < p + 12 + xlen:
            si1, si2, slen = raw[q], raw[q + 1], struct.unpack_from("<H", raw, q + 2)[0]
            if (si1, si2) == (66, 67):
                bsize, = struct.unpack_from("<H", raw, q + 4)
            q += 4 + slen
        assert bsize is not None
        cdata = raw[p + 12 + xlen:p + bsize + 1 - 8]
        block = zlib.decompress(cdata, -15)
        crc, isize = struct.unpack_from("<II", raw, p + bsize + 1 - 8)
        assert zlib.crc32(block) & 0xFFFFFFFF == crc and len(block) == isize
        data += block
        p += bsize + 1
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p)
    p += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, p)
        p += 4 + l_name + 4
    out = []
    size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while p < len(data):
        block_size, = struct.unpack_from("<i", data, p)
        rec = data[p + 4:p + 4 + block_size]
        p += 4 + block_size
        _, _, l_name, _, _, n_cig, flag, l_seq, _, _, _ = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        q = 32
        name = rec[q:q + l_name - 1].decode()
        q += l_name + 4 * n_cig
        nib = []
        for byte in rec[q:q + (l_seq + 1) // 2]:
            nib += [byte >> 4, byte & 15]
        seq = "".join(SEQ_CODE[x] for x in nib[:l_seq])
        q += (l_seq + 1) // 2 + l_seq
        tags = {}
        while q < len(rec):
            tag, typ = rec[q:q + 2].decode(), chr(rec[q + 2])
            q += 3
            if typ == "A":
                tags[tag] = chr(rec[q]); q += 1
            elif typ in fmt:
                tags[tag], = struct.unpack_from(fmt[typ], rec, q); q += size[typ]
            elif typ in "ZH":
                e = rec.index(b"\0", q); tags[tag] = rec[q:e].decode(); q = e + 1
            elif typ == "B":
                sub = chr(rec[q]); cnt, = struct.unpack_from("<i", rec, q + 1)
                tags[tag] = list(struct.unpack_from("<%d%s" % (cnt, fmt[sub][1]), rec, q + 5)); q += 5 + cnt * size[sub]
            else:
                raise AssertionError(typ)
        out.append((name, seq, flag, n_cig, tags))
    return out


def _records():
    rng = np.random.default_rng(12)
    recs, want = [], []
    for i in range(40):
        n = int(rng.integers(1, 3000))
        seq = "".join(rng.choice(list("ACGTN"), size=n))
        name = str(uuid.UUID(int=int(rng.integers(1, 2 ** 62))))
        mv = rng.integers(0, 2, size=int(rng.integers(1, 4000))).astype(np.int8)
        sm, sd, qs = np.float32(rng.uniform(50, 120)), np.float32(rng.uniform(5, 30)), np.float32(rng.uniform(5, 30))
        ns, ts = int(rng.integers(1000, 10 ** 6)), int(rng.integers(0, 500))
        tags = b"qsf" + struct.pack("<f", qs) + spec_int_tag("ns", ns) + spec_int_tag("ts", ts)
        tags += b"mvBc" + struct.pack("<i", len(mv)) + mv.tobytes()
        tags += b"smf" + struct.pack("<f", sm) + b"sdf" + struct.pack("<f", sd) + b"fnZ" + b"file_%d.pod5\0" % (i % 3)
        tags += b"stZ2024-01-01T00:00:00.000+00:00\0" + b"RGZrun_model\0" + b"xaAQ" + b"xhH1AE301\0"
        tags += b"xbBS" + struct.pack("<i", 3) + struct.pack("<3H", 1, 2, 65535) + b"xfBf" + struct.pack("<i", 2) + struct.pack("<2f", 0.5, -1.25)
        w = {"qs": float(qs), "ns": ns, "ts": ts, "sm": float(sm), "sd": float(sd), "fn": "file_%d.pod5" % (i % 3)}
        if i % 4 == 0:   # a split read: parent id and its offset into the parent's signal
            sp = int(rng.integers(0, 10 ** 5))
            tags += b"piZ" + name[::-1].encode() + b"\0" + spec_int_tag("sp", sp)
            w.update(pi=name[::-1], sp=sp)
        aligned = i % 5 == 0   # an aligned record: a reference, CIGAR operations the reader has to step over
        recs.append(spec_record(name, seq, rng.integers(0, 50, size=n).astype(np.uint8), 0 if aligned else 4,
                                [(n // 2, "M"), (n - n // 2, "S")] if aligned else [], tags, 0 if aligned else -1, 100 if aligned else -1))
        want.append((name, seq, w))
    return recs, want


def _blocks(data: bytes, cuts):
    """BGZF blocks cut at arbitrary byte positions (records straddle block boundaries in real files) + the EOF block"""
    edges = [0] + sorted(cuts) + [len(data)]
    return b"".join(spec_bgzf_block(data[a:b]) for a, b in zip(edges[:-1], edges[1:]) if b > a) + spec_bgzf_block(b"")


def test_bam_from_the_specification_is_read_by_both_decoders(tmp_path):
    recs, want = _records()
    data = spec_bam(recs, refs=[("chr1", 1000000)])
    rng = np.random.default_rng(3)
    raw = _blocks(data, [int(x) for x in rng.integers(1, len(data), size=9)] + [7, 8, 9])  # incl. cuts inside the header
    path = str(tmp_path / "spec.bam")
    open(path, "wb").write(raw)
    assert gzip.open(path).read() == data            # a BGZF file is a valid multi-member gzip file
    second = spec_decode(raw)
    ours = list(bam_io.iter_bam(path))
    assert len(ours) == len(second) == len(want)
    for rec, (name2, seq2, flag2, ncig2, tags2), (name, seq, w) in zip(ours, second, want):
        assert rec.query_name == name2 == name and rec.query_sequence == seq2 == seq
        for k in ("qs", "ns", "ts", "sm", "sd", "fn", "pi", "sp"):
            assert rec.has_tag(k) == (k in w) == (k in tags2), k
            if k in w:
                assert rec.get_tag(k) == w[k] == tags2[k], (name, k)   # floats: the float32 value, exactly
        assert not rec.has_tag("f5")
        # the tags nobody on this path reads are stepped over correctly by both (a misparsed B array derails the rest)
        assert list(rec.get_tag("mv")) == tags2["mv"] and list(rec.get_tag("xb")) == [1, 2, 65535] == tags2["xb"]
        assert list(rec.get_tag("xf")) == [0.5, -1.25] == tags2["xf"] and rec.get_tag("xa") == "Q" == tags2["xa"]
        assert rec.get_tag("xh") == "1AE301" == tags2["xh"]


def test_job_generator_on_the_specification_bam(tmp_path):
    """segment.py:222-256 on that file: start = sp + ts, end = sp + ns, signal id = pi or the read's own name, file = fn"""
    from dynamont_amd.segmentation import segment as seg
    recs, want = _records()
    path = str(tmp_path / "spec.bam")
    open(path, "wb").write(_blocks(spec_bam(recs), [5000, 70000]))
    jobs = list(seg.generate_jobs("/data", path, 10.0))
    keep = [(n, s, w) for n, s, w in want if w["qs"] >= 10.0]
    assert len(jobs) == len(keep) > 10
    for j, (name, seq, w) in zip(jobs, keep):
        assert j[0] == "/data/" + w["fn"] and j[1] == w["sm"] and j[2] == w["sd"]
        assert j[3] == w.get("sp", 0) + w["ts"] and j[4] == w.get("sp", 0) + w["ns"]
        assert j[5] == seq and j[6] == name and j[7] == w.get("pi", name)


def test_write_bam_output_is_read_by_the_specification_decoder(tmp_path):
    """the other direction: what bam_io.write_bam writes (synthetic datasets, bench.py's e2e_cli input) is a BAM by the
    specification's own reading"""
    recs = [("read-%d" % i, "ACGTN" * (i + 1) + "A" * (i % 2), {"qs": 12.5, "ns": 70000 + i, "ts": 3, "fn": "x.pod5", "sm": 90.0, "sd": 15.0})
            for i in range(30)]
    path = str(tmp_path / "w.bam")
    bam_io.write_bam(path, recs)
    got = spec_decode(open(path, "rb").read())
    assert [(n, s) for n, s, *_ in got] == [(n, s) for n, s, _ in recs]
    assert all(g[4] == t and g[2] == 4 and g[3] == 0 for g, (_, _, t) in zip(got, recs))


# ---- official tools, when the box has them -------------------------------------------------------------------------------
def _reads_for_official_tools():
    rng = np.random.default_rng(77)
    ids = [str(uuid.UUID(int=int(v))) for v in rng.integers(1, 2 ** 62, 5)]
    adcs = [rng.normal(500, 60, n).astype(np.int16) for n in (10, 4000, 123457, 3, 250000)]
    return ids, adcs, rng.uniform(-300, -200, 5).astype(np.float32), rng.uniform(0.1, 0.2, 5).astype(np.float32)


@pytest.mark.gpu
def test_official_pod5_package_round_trip(tmp_path):
    """ONT's `pod5` writes, pod5_native reads -- and the reverse: int16 signal, calibration, read ids identical."""
    pod5 = pytest.importorskip("pod5", reason="ONT's `pod5` package is not installed on this box: the vendor-free POD5 reader/writer "
                               "stays unpinned against an officially written file (N1 partial)")
    import datetime
    from dynamont_amd import pod5_native as P
    ids, adcs, offs, scales = _reads_for_official_tools()
    path = str(tmp_path / "official.pod5")
    run = pod5.RunInfo(acquisition_id="a", acquisition_start_time=datetime.datetime(2024, 1, 1, tzinfo=datetime.timezone.utc), adc_max=4095,
                       adc_min=-4096, context_tags={}, device_type="promethion", flow_cell_id="f", flow_cell_product_code="p",
                       protocol_name="n", protocol_run_id="r", protocol_start_time=datetime.datetime(2024, 1, 1, tzinfo=datetime.timezone.utc),
                       sample_id="s", sample_rate=4000, sequencing_kit="k", sequencer_position="x", sequencer_position_type="t",
                       software="dynamont_amd tests", system_name="sys", system_type="st", tracking_id={})
    with pod5.Writer(path) as w:
        for i, (rid, adc, o, s) in enumerate(zip(ids, adcs, offs, scales)):
            w.add_read(pod5.Read(read_id=uuid.UUID(rid), pore=pod5.Pore(channel=1 + i, well=1, pore_type="r10"),
                                 calibration=pod5.Calibration(offset=float(o), scale=float(s)), read_number=i, start_sample=0,
                                 median_before=0.0, end_reason=pod5.EndReason(pod5.EndReasonEnum.SIGNAL_POSITIVE, False),
                                 run_info=run, signal=adc))
    f = P.Pod5File(path)
    for rid, adc, o, s in zip(ids, adcs, offs, scales):
        got, go, gs = f.signal_adc(rid)
        assert np.array_equal(got, adc) and go == np.float32(o) and gs == np.float32(s)
        assert np.array_equal(f.signal(rid, True), (adc.astype(np.float32) + np.float32(o)) * np.float32(s))
    f.close()
    ours = str(tmp_path / "ours.pod5")
    P.write_pod5(ours, ids, adcs, offs, scales)
    with pod5.Reader(ours) as r:
        seen = {str(rec.read_id): rec for rec in r.reads()}
        for rid, adc, o, s in zip(ids, adcs, offs, scales):
            assert np.array_equal(seen[rid].signal, adc)
            assert np.float32(seen[rid].calibration.offset) == np.float32(o) and np.float32(seen[rid].calibration.scale) == np.float32(s)


@pytest.mark.gpu
def test_official_pysam_round_trip(tmp_path):
    """pysam (htslib) writes an unaligned BAM with dorado's tags, bam_io reads it -- and the reverse."""
    pysam = pytest.importorskip("pysam", reason="pysam / htslib is not installed on this box: the vendor-free BAM reader is pinned by the "
                                "specification-built file only (N1 partial)")
    rng = np.random.default_rng(5)
    recs = []
    for i in range(50):
        seq = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 2500))))
        tags = {"qs": float(np.float32(rng.uniform(5, 30))), "ns": int(rng.integers(1000, 10 ** 6)), "ts": int(rng.integers(0, 500)),
                "fn": f"f{i % 2}.pod5", "sm": float(np.float32(rng.uniform(50, 120))), "sd": float(np.float32(rng.uniform(5, 30)))}
        if i % 3 == 0:
            tags.update(pi=f"parent-{i}", sp=int(rng.integers(0, 10 ** 5)))
        recs.append((f"read-{i}", seq, tags))
    path = str(tmp_path / "official.bam")
    header = pysam.AlignmentHeader.from_dict({"HD": {"VN": "1.6", "SO": "unknown"}})
    with pysam.AlignmentFile(path, "wb", header=header) as f:
        for name, seq, tags in recs:
            a = pysam.AlignedSegment(header)
            a.query_name, a.query_sequence, a.flag = name, seq, 4
            for k, v in tags.items():
                a.set_tag(k, v, "f" if isinstance(v, float) else ("i" if isinstance(v, int) else "Z"))
            a.set_tag("mv", rng.integers(0, 2, size=100).astype(np.int8).tolist())
            f.write(a)
    got = list(bam_io.iter_bam(path))
    assert len(got) == len(recs)
    for r, (name, seq, tags) in zip(got, recs):
        assert r.query_name == name and r.query_sequence == seq
        for k, v in tags.items():
            assert r.get_tag(k) == v, (name, k)
    ours = str(tmp_path / "ours.bam")
    bam_io.write_bam(ours, recs)
    with pysam.AlignmentFile(ours, "rb", check_sq=False) as f:
        for rec, (name, seq, tags) in zip(f.fetch(until_eof=True), recs):
            assert rec.query_name == name and rec.query_sequence == seq
            for k, v in tags.items():
                assert rec.get_tag(k) == v, (name, k)
