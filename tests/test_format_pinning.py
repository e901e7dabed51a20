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


# ---- a POD5 file assembled from the specification ------------------------------------------------------------------------
# The same idea as the BAM above: nothing of dynamont_amd writes this file. Arrow IPC tables come from pyarrow called
# directly with the specification's schemas (extension types as their storage type + the ARROW:extension:* field metadata an
# IPC writer emits for them), the flatbuffers footer from a small generic builder written here from the flatbuffers encoding
# rules -- with a DIFFERENT physical layout than pod5_native.build_footer (vtables behind their tables, hence negative
# soffsets; the `format` field left at its default and therefore absent from the vtable; strings after the vector) -- and the
# VBZ chunks from a second StreamVByte-16 encoder (a plain loop over samples) around libzstd called through ctypes.
def _spec_svb16_zigzag_delta(samples) -> bytes:
    """pod5 SPECIFICATION.md, "VBZ": deltas of consecutive int16 samples (the first against 0), zigzag coded, then
    StreamVByte with 16-bit values: one KEY BIT per value (bit i % 8 of key byte i // 8; 0 = one data byte, 1 = two,
    little-endian), all key bytes first."""
    keys = bytearray((len(samples) + 7) // 8)
    data = bytearray()
    prev = 0
    for i, x in enumerate(int(v) for v in samples):
        d = (x - prev + 32768) % 65536 - 32768            # int16 wrap-around of the difference
        prev = x
        z = ((d << 1) ^ (d >> 15)) & 0xFFFF                # zigzag
        if z < 256:
            data.append(z)
        else:
            keys[i // 8] |= 1 << (i % 8)
            data += bytes((z & 255, z >> 8))
    return bytes(keys) + bytes(data)


def _spec_zstd(raw: bytes) -> bytes:
    import ctypes as C
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    out = C.create_string_buffer(z.ZSTD_compressBound(len(raw)))
    n = z.ZSTD_compress(out, len(out), raw, len(raw), 1)
    assert not z.ZSTD_isError(n)
    return out.raw[:n]


class _SpecFlatbuffer:
    """Just enough of the flatbuffers wire format (flatbuffers "Internals": little-endian; a uoffset32 points FORWARD from
    its own position; a table starts with an soffset32 = table position - vtable position; a vtable = u16 vtable bytes, u16
    table bytes, then one u16 per field = offset inside the table or 0 for a field left at its default)."""

    def __init__(self):
        self.b = bytearray(4)   # root uoffset
        self.fix = []           # (position of a uoffset, key of its target)
        self.at = {}            # key -> position

    def align(self, n):
        while len(self.b) % n:
            self.b.append(0)

    def table(self, key, fields):
        """fields: list of None (default: absent) | ('ref', target key) | (struct format, value). vtable BEHIND the table."""
        self.align(8)
        start = len(self.b)
        self.at[key] = start
        self.b += b"\0\0\0\0"           # soffset, patched once the vtable's place is known
        offs = []
        for f in fields:
            if f is None:
                offs.append(0)
                continue
            size = 4 if f[0] == "ref" else struct.calcsize(f[0])
            while (len(self.b) - start) % size:
                self.b.append(0)
            offs.append(len(self.b) - start)
            if f[0] == "ref":
                self.fix.append((len(self.b), f[1]))
                self.b += b"\0\0\0\0"
            else:
                self.b += struct.pack(f[0], f[1])
        size = len(self.b) - start
        self.align(2)
        vt = len(self.b)
        self.b += struct.pack("<HH", 4 + 2 * len(offs), size) + b"".join(struct.pack("<H", o) for o in offs)
        struct.pack_into("<i", self.b, start, start - vt)   # negative: the vtable lies behind

    def vector_of_refs(self, key, targets):
        self.align(4)
        self.at[key] = len(self.b)
        self.b += struct.pack("<I", len(targets))
        for t in targets:
            self.fix.append((len(self.b), t))
            self.b += b"\0\0\0\0"

    def string(self, key, text):
        self.align(4)
        self.at[key] = len(self.b)
        raw = text.encode()
        self.b += struct.pack("<I", len(raw)) + raw + b"\0"

    def finish(self, root) -> bytes:
        struct.pack_into("<I", self.b, 0, self.at[root])
        for pos, key in self.fix:
            assert self.at[key] > pos, "uoffsets point forward"
            struct.pack_into("<I", self.b, pos, self.at[key] - pos)
        return bytes(self.b)


def _spec_pod5(path, reads, chunk):
    """reads: (uuid, int16 samples, calibration offset, calibration scale). Table version 3 of pod5's SPECIFICATION.md."""
    import pyarrow as pa
    ext = lambda name: {b"ARROW:extension:name": name, b"ARROW:extension:metadata": b""}  # noqa: E731
    meta = {b"MINKNOW:pod5_version": b"0.3.2", b"MINKNOW:software": b"assembled from the specification",
            b"MINKNOW:file_identifier": b"cbf91180-0684-4a39-bf56-41eaf437de9e"}

    def ipc(schema, columns):
        sink = pa.BufferOutputStream()
        with pa.ipc.new_file(sink, schema.with_metadata(meta)) as w:
            w.write_batch(pa.record_batch(columns, schema=schema.with_metadata(meta)))
        return sink.getvalue().to_pybytes()

    # signal table: one row per chunk of at most `chunk` samples
    sig_ids, blobs, counts, rows_of = [], [], [], []
    for rid, adc, _, _ in reads:
        mine = []
        for a in range(0, max(1, len(adc)), chunk):
            part = adc[a:a + chunk]
            mine.append(len(blobs))
            sig_ids.append(rid.bytes)
            blobs.append(_spec_zstd(_spec_svb16_zigzag_delta(part)))
            counts.append(len(part))
        rows_of.append(mine)
    uuid_t = pa.binary(16)
    signal_schema = pa.schema([pa.field("read_id", uuid_t, metadata=ext(b"minknow.uuid")),
                               pa.field("signal", pa.large_binary(), metadata=ext(b"minknow.vbz")), pa.field("samples", pa.uint32())])
    signal_tab = ipc(signal_schema, [pa.array(sig_ids, uuid_t), pa.array(blobs, pa.large_binary()), pa.array(counts, pa.uint32())])
    dict_t = pa.dictionary(pa.int16(), pa.utf8())
    run_schema = pa.schema([pa.field("acquisition_id", pa.utf8()), pa.field("acquisition_start_time", pa.timestamp("ms", "UTC")),
                            pa.field("adc_max", pa.int16()), pa.field("adc_min", pa.int16()),
                            pa.field("context_tags", pa.map_(pa.utf8(), pa.utf8())), pa.field("device_type", pa.utf8()),
                            pa.field("flow_cell_id", pa.utf8()), pa.field("flow_cell_product_code", pa.utf8()), pa.field("protocol_name", pa.utf8()),
                            pa.field("protocol_run_id", pa.utf8()), pa.field("protocol_start_time", pa.timestamp("ms", "UTC")),
                            pa.field("sample_id", pa.utf8()), pa.field("sample_rate", pa.uint16()), pa.field("sequencing_kit", pa.utf8()),
                            pa.field("sequencer_position", pa.utf8()), pa.field("sequencer_position_type", pa.utf8()), pa.field("software", pa.utf8()),
                            pa.field("system_name", pa.utf8()), pa.field("system_type", pa.utf8()), pa.field("tracking_id", pa.map_(pa.utf8(), pa.utf8()))])
    one = lambda v, t: pa.array([v], t)  # noqa: E731
    run_tab = ipc(run_schema, [one("acq-1", pa.utf8()), one(1700000000000, pa.timestamp("ms", "UTC")), one(4095, pa.int16()), one(-4096, pa.int16()),
                               one([("k", "v")], pa.map_(pa.utf8(), pa.utf8())), one("promethion", pa.utf8()), one("FC1", pa.utf8()), one("FLO", pa.utf8()),
                               one("proto", pa.utf8()), one("run-1", pa.utf8()), one(1700000000000, pa.timestamp("ms", "UTC")), one("s", pa.utf8()),
                               one(4000, pa.uint16()), one("kit", pa.utf8()), one("1A", pa.utf8()), one("p2", pa.utf8()), one("sw", pa.utf8()),
                               one("sys", pa.utf8()), one("st", pa.utf8()), one([], pa.map_(pa.utf8(), pa.utf8()))])
    n = len(reads)
    reads_schema = pa.schema([pa.field("read_id", uuid_t, metadata=ext(b"minknow.uuid")), pa.field("signal", pa.list_(pa.uint64())),
                              pa.field("read_number", pa.uint32()), pa.field("start", pa.uint64()), pa.field("median_before", pa.float32()),
                              pa.field("num_minknow_events", pa.uint64()), pa.field("tracked_scaling_scale", pa.float32()),
                              pa.field("tracked_scaling_shift", pa.float32()), pa.field("predicted_scaling_scale", pa.float32()),
                              pa.field("predicted_scaling_shift", pa.float32()), pa.field("num_reads_since_mux_change", pa.uint32()),
                              pa.field("time_since_mux_change", pa.float32()), pa.field("num_samples", pa.uint64()), pa.field("channel", pa.uint16()),
                              pa.field("well", pa.uint8()), pa.field("pore_type", dict_t), pa.field("calibration_offset", pa.float32()),
                              pa.field("calibration_scale", pa.float32()), pa.field("end_reason", dict_t), pa.field("end_reason_forced", pa.bool_()),
                              pa.field("run_info", dict_t)])
    f32 = lambda vals: pa.array(np.asarray(vals, dtype=np.float32), pa.float32())  # noqa: E731
    dic = lambda word: pa.DictionaryArray.from_arrays(pa.array([0] * n, pa.int16()), pa.array([word], pa.utf8()))  # noqa: E731
    reads_tab = ipc(reads_schema, [pa.array([r[0].bytes for r in reads], uuid_t), pa.array(rows_of, pa.list_(pa.uint64())),
                                   pa.array(range(n), pa.uint32()), pa.array([0] * n, pa.uint64()), f32([200.0] * n), pa.array([0] * n, pa.uint64()),
                                   f32([1.0] * n), f32([0.0] * n), f32([1.0] * n), f32([0.0] * n), pa.array([0] * n, pa.uint32()), f32([0.0] * n),
                                   pa.array([len(r[1]) for r in reads], pa.uint64()), pa.array(range(1, n + 1), pa.uint16()), pa.array([1] * n, pa.uint8()),
                                   dic("not_set"), f32([r[2] for r in reads]), f32([r[3] for r in reads]), dic("signal_positive"),
                                   pa.array([False] * n, pa.bool_()), dic("acq-1")])
    marker = uuid.UUID("5b6e9c2a-2b0c-4d8c-9d4e-0123456789ab").bytes
    sig8 = b"\x8bPOD\r\n\x1a\n"
    body = bytearray(sig8 + marker)
    where = {}
    for kind, blob in ((1, signal_tab), (4, run_tab), (0, reads_tab)):   # content_type: 1 signal, 4 run info, 0 reads
        where[kind] = (len(body), len(blob))
        body += blob
        body += b"\0" * (-len(body) % 8)
        body += marker
    fb = _SpecFlatbuffer()
    fb.table("footer", [("ref", "id"), ("ref", "sw"), ("ref", "ver"), ("ref", "contents")])
    fb.vector_of_refs("contents", ["e1", "e4", "e0"])
    for kind in (1, 4, 0):   # EmbeddedFile { offset:int64; length:int64; format:short = FeatherV2 (default, absent); content_type:short }
        fb.table(f"e{kind}", [("<q", where[kind][0]), ("<q", where[kind][1]), None, ("<h", kind)])
    fb.string("ver", "0.3.2")
    fb.string("id", "cbf91180-0684-4a39-bf56-41eaf437de9e")
    fb.string("sw", "assembled from the specification")
    footer = fb.finish("footer")
    body += b"FOOTER\0\0" + footer + b"\0" * (-len(footer) % 8)
    body += struct.pack("<q", len(footer) + (-len(footer) % 8)) + marker + sig8
    open(path, "wb").write(bytes(body))


def test_pod5_from_the_specification_is_read_by_the_reader_and_the_native_decoder(tmp_path, native_lib):
    """N1: the container framing, the footer and the VBZ stage are ours alone (Arrow IPC is read by Apache Arrow, zstd frames by
    libzstd) -- pinned here against a file no code of dynamont_amd wrote. Samples cover every svb16 case: deltas of one and
    of two bytes, the int16 extremes (wrap-around differences), a chunk whose sample count is not a multiple of 8, a
    one-sample read, a read of several chunks."""
    import ctypes as C
    from dynamont_amd import pod5_native as P
    rng = np.random.default_rng(123)
    walk = np.clip(np.cumsum(rng.integers(-300, 301, size=25013)) + 500, -32768, 32767).astype(np.int16)
    edge = np.array([0, 32767, -32768, -1, 1, 127, 128, -128, -129, 255, 256, 32767, 32767, -32768], dtype=np.int16)
    reads = [(uuid.UUID(int=int(rng.integers(1, 2 ** 62))), adc, np.float32(o), np.float32(sc))
             for adc, o, sc in ((walk, -243.0, 0.1462), (edge, 3.0, 0.25), (np.array([-7], dtype=np.int16), 0.0, 1.0),
                                (rng.integers(-32768, 32768, size=4099).astype(np.int16), -10.5, 0.2))]
    path = str(tmp_path / "spec.pod5")
    _spec_pod5(path, reads, chunk=10240)
    f = P.Pod5File(path)
    assert f.footer["software"] == "assembled from the specification" and f.footer["pod5_version"] == "0.3.2"
    assert sorted(c["content_type"] for c in f.footer["contents"]) == [0, 1, 4] and all(c["format"] == 0 for c in f.footer["contents"])
    assert sorted(f.read_ids) == sorted(str(r[0]) for r in reads)
    for rid, adc, o, sc in reads:
        got, go, gs = f.signal_adc(str(rid))
        assert got.dtype == np.int16 and np.array_equal(got, adc) and np.float32(go) == o and np.float32(gs) == sc
        assert np.array_equal(f.signal(str(rid), True), (adc.astype(np.float32) + o) * sc)   # pod5_io.py:6-16: signal_pa
    # the library's own decoder (vbz_decode.cpp, what dyn_batch_align_vbz_async runs) on the chunks as they lie in the file
    ids16 = np.frombuffer(b"".join(r[0].bytes for r in reads), dtype=np.uint8).reshape(-1, 16)
    found, ptrs, nbytes, samples, read_off, cal_o, cal_s = f.signal_chunks_batch(ids16)
    assert found.all() and [int(x) for x in np.diff(read_off.astype(np.int64))] == [3, 1, 1, 1]
    err = C.create_string_buffer(256)
    for i, (rid, adc, o, sc) in enumerate(reads):
        parts = []
        for c in range(int(read_off[i]), int(read_off[i + 1])):
            out = np.full(int(samples[c]) + 1, 21845, dtype=np.int16)
            rc = native_lib.dyn_vbz_decode(C.c_void_p(int(ptrs[c])), int(nbytes[c]), int(samples[c]), out.ctypes.data_as(C.c_void_p), err, 256)
            assert rc == 0, err.value
            assert out[-1] == 21845
            parts.append(out[:-1])
        assert np.array_equal(np.concatenate(parts), adc) and cal_o[i] == o and cal_s[i] == sc
    f.close()


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
