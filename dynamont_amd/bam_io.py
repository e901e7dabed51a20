"""Vendor-free readers for basecalls in SAM text and BAM (BGZF) form -- SURVEY.md §8f N1.

The reference reads basecalls through pysam (segment.py:222-256, train.py:121-153) and only needs
the read name, the sequence and eight optional tags: qs, pi, ns, ts, sp, fn|f5, sm, sd. pysam/htslib
are absent from the ROCm image, so the two container formats are parsed directly from the SAM
specification (BAM: BGZF = concatenated gzip members; records little-endian). Unaligned dorado output
(uBAM) is the expected input; alignment fields are skipped. ``write_bam``/``write_sam`` exist for
tests and for producing synthetic datasets; they emit only what the readers consume.
"""
from __future__ import annotations

import binascii
import gzip
import struct
import zlib

import numpy as np

from dynamont_amd.pod5_io import BasecallRecord

_SEQ_DECODE = "=ACMGRSVTWYHKDBN"
_HEX_TO_BASE = bytes.maketrans(b"0123456789abcdef", _SEQ_DECODE.encode())
_TAG_SIZE = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
_TAG_NP = {"c": "<i1", "C": "<u1", "s": "<i2", "S": "<u2", "i": "<i4", "I": "<u4", "f": "<f4"}
_TAG_FMT = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}


def _parse_sam_tag(field: str):
    tag, typ, val = field.split(":", 2)
    if typ == "i":
        return tag, int(val)
    if typ == "f":
        # a SAM `f` tag is a single-precision value (SAM spec 1.5): pysam and the BAM form of the same record yield
        # the float32-rounded number, so must this (sm/sd feed the normalisation: a double here would change the
        # low-order bits of every sample compared with the .bam of the same run)
        return tag, float(np.float32(val))
    return tag, val  # Z, A, H, B kept as text


def iter_sam(path: str):
    """Records of a SAM text file (header lines start with '@')."""
    with open(path) as f:
        for line in f:
            if not line or line[0] == "@":
                continue
            p = line.rstrip("\n").split("\t")
            if len(p) < 11:
                continue
            tags = dict(_parse_sam_tag(x) for x in p[11:] if x.count(":") >= 2)
            yield BasecallRecord(p[0], p[9], tags)


class _Stream:
    """Bounded-memory reader over the inflated BGZF stream (a series of gzip members, which Python's gzip module
    reads as one stream): a multi-GB dorado BAM is never held in memory, and the first record is available after
    the first chunk. The stream is pulled in 1 MB pieces and records are cut out of that buffer: two gzip reads per
    record (its length, its body) cost more than parsing the record."""

    CHUNK = 1 << 20

    def __init__(self, path: str):
        self.f = gzip.open(path, "rb")
        self.buf = b""
        self.pos = 0

    def take(self, n: int) -> bytes:
        if self.pos + n > len(self.buf):
            rest = self.buf[self.pos:]
            parts = [rest]
            have = len(rest)
            while have < n:
                more = self.f.read(max(self.CHUNK, n - have))
                if not more:
                    break
                parts.append(more)
                have += len(more)
            self.buf = b"".join(parts)
            self.pos = 0
        out = self.buf[self.pos:self.pos + n]
        self.pos += len(out)
        return out

    def close(self):
        self.f.close()


def iter_bam(path: str):
    """Records of a BAM file (SAM spec §4): name, sequence and tags of every alignment record, streamed."""
    st = _Stream(path)
    try:
        head = st.take(8)
        if head[:4] != b"BAM\x01":
            raise ValueError(f"{path}: not a BAM file")
        l_text, = struct.unpack_from("<i", head, 4)
        st.take(l_text)
        n_ref, = struct.unpack("<i", st.take(4))
        for _ in range(n_ref):
            l_name, = struct.unpack("<i", st.take(4))
            st.take(l_name + 4)
        while True:
            hdr = st.take(4)
            if len(hdr) < 4:
                break
            block_size, = struct.unpack("<i", hdr)
            data = st.take(block_size)
            if len(data) < block_size:
                raise ValueError(f"{path}: truncated BAM record")
            yield _parse_bam_record(path, data)
    finally:
        st.close()


def _parse_bam_record(path: str, data: bytes):
    pos, end = 0, len(data)
    (_ref, _p, l_read_name, _mapq, _bin, n_cigar, _flag, l_seq, _nref, _npos, _tlen) = struct.unpack_from("<iiBBHHHiiii", data, pos)
    q = pos + 32
    name = data[q:q + l_read_name - 1].decode()
    q += l_read_name + 4 * n_cigar
    nb = (l_seq + 1) // 2
    # two bases per byte, high nibble first: hexlify spells the nibbles out as hex digits, translate maps those to bases
    # (two C calls per record; the NumPy table lookup this replaces cost 3 us of call overhead for a 2 000-base read)
    seq = binascii.hexlify(data[q:q + nb]).translate(_HEX_TO_BASE)[:l_seq].decode("ascii")
    q += nb + l_seq
    tags = {}
    while q < end:
        tag = data[q:q + 2].decode()
        typ = chr(data[q + 2])
        q += 3
        if typ in _TAG_FMT:
            tags[tag], = struct.unpack_from(_TAG_FMT[typ], data, q)
            q += _TAG_SIZE[typ]
        elif typ == "A":
            tags[tag] = chr(data[q])
            q += 1
        elif typ in "ZH":
            e = data.index(b"\0", q)
            tags[tag] = data[q:e].decode()
            q = e + 1
        elif typ == "B":
            sub = chr(data[q])
            cnt, = struct.unpack_from("<i", data, q + 1)
            # dorado's move table (mv:B:c) has thousands of entries per read and nobody on this path reads it: a NumPy
            # view instead of a Python list per element (pysam returns an array.array here)
            tags[tag] = np.frombuffer(data, dtype=_TAG_NP[sub], count=cnt, offset=q + 5)
            q += 5 + _TAG_SIZE[sub] * cnt
        else:
            raise ValueError(f"{path}: unknown BAM tag type {typ!r}")
    return BasecallRecord(name, seq, tags)


def _sam_tag_text(tag, value):
    if isinstance(value, bool):
        value = int(value)
    if isinstance(value, int):
        return f"{tag}:i:{value}"
    if isinstance(value, float):
        return f"{tag}:f:{value!r}"
    return f"{tag}:Z:{value}"


def write_sam(path: str, records) -> None:
    """records: iterable of (name, sequence, tags dict). Unaligned records (flag 4)."""
    with open(path, "w") as w:
        w.write("@HD\tVN:1.6\tSO:unknown\n")
        for name, seq, tags in records:
            fields = [name, "4", "*", "0", "0", "*", "*", "0", "0", seq, "*"] + [_sam_tag_text(k, v) for k, v in tags.items()]
            w.write("\t".join(fields) + "\n")


def _bgzf_block(payload: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(payload) + comp.flush()
    bsize = len(body) + 25
    header = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return header + body + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


def write_bam(path: str, records) -> None:
    """Minimal uBAM writer (tests / synthetic datasets): floats as 'f' (single precision, as dorado
    writes sm/sd/qs), ints as 'i', strings as 'Z'."""
    enc_lut = np.full(256, 15, dtype=np.uint8)  # anything outside the 16 IUPAC codes is stored as N
    for i, c in enumerate(_SEQ_DECODE):
        enc_lut[ord(c)] = i
    out = bytearray(b"BAM\x01")
    text = b"@HD\tVN:1.6\tSO:unknown\n"
    out += struct.pack("<i", len(text)) + text + struct.pack("<i", 0)
    for name, seq, tags in records:
        nm = name.encode() + b"\0"
        codes = enc_lut[np.frombuffer(seq.upper().encode("latin-1"), dtype=np.uint8)]
        if len(codes) % 2:
            codes = np.append(codes, np.uint8(0))
        packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()
        tagb = bytearray()
        for k, v in tags.items():
            if isinstance(v, int):
                tagb += k.encode() + b"i" + struct.pack("<i", v)
            elif isinstance(v, float):
                tagb += k.encode() + b"f" + struct.pack("<f", v)
            else:
                tagb += k.encode() + b"Z" + str(v).encode() + b"\0"
        body = struct.pack("<iiBBHHHiiii", -1, -1, len(nm), 0, 4680, 0, 4, len(seq), -1, -1, 0) + nm + packed + b"\xff" * len(seq) + bytes(tagb)
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as w:
        data = bytes(out)
        for i in range(0, len(data), 0xFF00):
            w.write(_bgzf_block(data[i:i + 0xFF00]))
        w.write(_bgzf_block(b""))  # EOF marker block


# ---------------------------------------------------------------------------------------------
# the native reader (csrc/bam_reader.cpp): jobs in columns, a batch per call
# ---------------------------------------------------------------------------------------------
class JobBatch:
    """The jobs of one batch as columns (dyn_job_batch, copied out of the reader): what generate_jobs yields read by read
    (segment.py:189-258), for `n` reads at once. ``names`` / ``sids`` are NUL-terminated strings back to back (uint8
    arrays) with ``*_off[i]`` = start of entry i; ``seqs`` the sequences in aligner orientation back to back (bytes) with
    ``seq_off``; ``uuid`` the signal ids as (n, 16) uint8 with ``uuid_ok``; ``shift``, ``scale`` (sm, sd), ``start``,
    ``end`` (sp + ts, sp + ns); ``files`` the distinct raw-file names of the batch and ``file_id`` each read's index; ``bases`` the stored
    sequence lengths."""

    __slots__ = ("n", "names", "name_off", "sids", "sid_off", "uuid", "uuid_ok", "seqs", "seq_off", "shift", "scale",
                 "start", "end", "files", "file_id", "bases")

    def name(self, i: int) -> str:
        return self.names[int(self.name_off[i]):int(self.name_off[i + 1]) - 1].tobytes().decode()

    def sid(self, i: int) -> str:
        return self.sids[int(self.sid_off[i]):int(self.sid_off[i + 1]) - 1].tobytes().decode()

    def read(self, i: int) -> str:
        return self.seqs[int(self.seq_off[i]):int(self.seq_off[i + 1])].decode("ascii")

    def take(self, keep: np.ndarray) -> "JobBatch":
        """the sub-batch of the reads ``keep`` (ascending indices); strings are re-packed"""
        keep = np.asarray(keep, dtype=np.int64)
        out = JobBatch()
        out.n = len(keep)

        def repack(buf, off):
            lens = (off[1:] - off[:-1])[keep]
            new_off = np.zeros(len(keep) + 1, dtype=np.uint64)
            np.cumsum(lens, out=new_off[1:])
            total = int(new_off[-1])
            idx = np.arange(total, dtype=np.int64) - np.repeat(new_off[:-1].astype(np.int64), lens.astype(np.int64)) \
                + np.repeat(off[:-1][keep].astype(np.int64), lens.astype(np.int64))
            return buf[idx], new_off

        out.names, out.name_off = repack(self.names, self.name_off)
        out.sids, out.sid_off = repack(self.sids, self.sid_off)
        seqs, out.seq_off = repack(np.frombuffer(self.seqs, dtype=np.uint8), self.seq_off)
        out.seqs = seqs.tobytes()
        out.uuid, out.uuid_ok = self.uuid[keep], self.uuid_ok[keep]
        out.shift, out.scale, out.start, out.end = self.shift[keep], self.scale[keep], self.start[keep], self.end[keep]
        out.files, out.file_id, out.bases = self.files, self.file_id[keep], self.bases[keep]
        return out


class NativeBamJobs:
    """dyn_bam_open / dyn_bam_next: the basecalls of a BAM file as JobBatch columns. ``rna``: sequences come reversed and
    with ``pad`` in front unless they start with it (segment.py:149-153). ``min_qual``, ``rank``, ``world`` as in
    generate_jobs / segment(): reads below the quality are skipped (``skipped``), of the others index % world == rank is
    kept. A tag the reference reads unconditionally and that is missing raises KeyError, like pysam's get_tag."""

    def __init__(self, path: str, rna: bool = False, pad: str = "", min_qual: float = 0.0, rank: int = 0, world: int = 1,
                 threads: int = 4):
        import ctypes as C
        from dynamont_amd import _native as N
        self._C, self._N, self._L = C, N, N.lib()
        self._flags = 1 if rna else 0
        self._q, self._rank, self._world = float(min_qual or 0.0), int(rank), int(world)
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = self._L.dyn_bam_open(path.encode(), int(threads), pad.encode(), C.byref(h), err, 1024)
        if rc != N.DYN_OK:
            raise ValueError(err.value.decode())
        self._h = h

    @property
    def skipped(self) -> int:
        return int(self._L.dyn_bam_skipped(self._h)) if self._h else 0

    def next(self, max_reads: int) -> JobBatch | None:
        C, N = self._C, self._N
        b = N.DynJobBatch()
        err = C.create_string_buffer(1024)
        rc = self._L.dyn_bam_next(self._h, int(max_reads), self._flags, self._q, self._rank, self._world, C.byref(b), err, 1024)
        if rc != N.DYN_OK:
            msg = err.value.decode()
            if msg.startswith("tag '"):
                raise KeyError(msg)
            raise ValueError(msg)
        n = int(b.n)
        if n == 0:
            return None

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            addr = ptr if isinstance(ptr, int) else C.addressof(ptr.contents)
            return np.ctypeslib.as_array((C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(addr)).view(dtype).copy()

        out = JobBatch()
        out.n = n
        out.names, out.name_off = arr(b.names, int(b.names_bytes), np.uint8), arr(b.name_off, n + 1, np.uint64)
        out.sids, out.sid_off = arr(b.signal_ids, int(b.signal_ids_bytes), np.uint8), arr(b.signal_id_off, n + 1, np.uint64)
        out.uuid = arr(b.signal_uuid, 16 * n, np.uint8).reshape(n, 16)
        out.uuid_ok = arr(b.signal_uuid_ok, n, np.uint8)
        out.seqs = C.string_at(b.seqs, int(b.seqs_bytes)) if b.seqs_bytes else b""
        out.seq_off = arr(b.seq_off, n + 1, np.uint64)
        out.shift, out.scale = arr(b.shift, n, np.float64), arr(b.scale, n, np.float64)
        out.start, out.end = arr(b.start, n, np.int64), arr(b.end, n, np.int64)
        files = C.string_at(b.files, int(b.files_bytes)) if b.files_bytes else b""
        out.files = [f.decode() for f in files.split(b"\0")[:int(b.n_files)]]
        out.file_id = arr(b.file_id, n, np.uint32)
        out.bases = arr(b.bases, n, np.uint32)
        return out

    def close(self) -> None:
        if self._h:
            self._L.dyn_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
