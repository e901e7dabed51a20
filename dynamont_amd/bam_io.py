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
