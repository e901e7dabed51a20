"""Vendor-free POD5 access: what ``open_pod5`` falls back to when the ``pod5`` package is absent
(the reference reads raw signal through ``pod5.Reader``, ``src/dynamont/pod5_io.py``).

Written from the published POD5 format specification (pod5-file-format ``docs/SPECIFICATION.md``,
table version 3) with what the ROCm image offers: ``pyarrow`` for the embedded Arrow IPC tables,
the system libzstd through ctypes, NumPy for the StreamVByte stage. NOT yet checked against a file
produced by the ``pod5`` package itself (neither the package nor a sample file is available in the
build image); ``write_pod5`` below follows the same text, so the round-trip tests pin the two halves
to each other, not to ONT's implementation. ``open_pod5`` prefers the real package whenever it imports.

Container layout (all integers little-endian)::

    signature  8B  8B 50 4F 44 0D 0A 1A 0A
    section marker (16-byte UUID, the same one after every section)
    embedded Arrow IPC file, padded to 8 bytes, section marker          (signal, run info, reads tables)
    ...
    "FOOTER\\0\\0", flatbuffers Footer padded to 8 bytes, int64 footer length, section marker, signature

    Footer       { file_identifier:string; software:string; pod5_version:string; contents:[EmbeddedFile] }
    EmbeddedFile { offset:int64; length:int64; format:short (0 = FeatherV2); content_type:short }
    content_type: 0 reads table, 1 signal table, 2 read-id index, 3 other index, 4 run-info table

    signal table: read_id fixed_size_binary[16] (minknow.uuid), signal large_binary (minknow.vbz) or
                  large_list<int16>, samples uint32
    reads table:  read_id, signal list<uint64> (rows of the signal table), calibration_offset float,
                  calibration_scale float, ... ; picoampere = (adc + calibration_offset) * calibration_scale

    VBZ: zstd( svb16( zigzag( delta(int16 samples) ) ) ); svb16 = ceil(n/8) key bytes (bit i of key byte
    i/8, LSB first: 0 -> one data byte, 1 -> two, little-endian) followed by the data bytes.
"""
from __future__ import annotations

import ctypes as C
import mmap
import struct
import uuid

import numpy as np

SIGNATURE = b"\x8bPOD\r\n\x1a\n"
FOOTER_MAGIC = b"FOOTER\x00\x00"
CT_READS, CT_SIGNAL, CT_READ_ID_INDEX, CT_OTHER_INDEX, CT_RUN_INFO = 0, 1, 2, 3, 4


# ---------------------------------------------------------------------------------------------
# StreamVByte-16 + zigzag + delta (numpy) and zstd (ctypes)
# ---------------------------------------------------------------------------------------------
def svb16_decode(buf: np.ndarray, count: int) -> np.ndarray:
    """uint8[...] -> int16[count]"""
    if count == 0:
        return np.zeros(0, dtype=np.int16)
    kb = (count + 7) // 8
    keys = np.unpackbits(buf[:kb], bitorder="little")[:count].astype(np.int64)
    sizes = keys + 1
    ends = np.cumsum(sizes)
    if kb + int(ends[-1]) > len(buf):
        raise ValueError("VBZ: truncated svb16 stream")
    starts = ends - sizes + kb
    data = np.concatenate([buf, np.zeros(1, dtype=np.uint8)])  # starts+1 of a final 1-byte value
    v = data[starts].astype(np.uint16) | (np.where(keys == 1, data[starts + 1], 0).astype(np.uint16) << 8)
    zz = (v >> 1) ^ (-(v & 1).astype(np.int16)).astype(np.uint16)  # zigzag^-1, modulo 2^16
    return np.cumsum(zz, dtype=np.uint16).view(np.int16)          # delta^-1, modulo 2^16


def svb16_encode(x: np.ndarray) -> np.ndarray:
    """int16[n] -> uint8[ceil(n/8) + data]"""
    x = np.ascontiguousarray(x, dtype=np.int16)
    n = len(x)
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    u = x.view(np.uint16)
    d = np.empty(n, dtype=np.uint16)
    d[0] = u[0]
    d[1:] = u[1:] - u[:-1]                                          # modulo 2^16
    s = d.view(np.int16)
    zz = ((s << 1) ^ (s >> 15)).view(np.uint16)
    keys = (zz > 255).astype(np.uint8)
    out_keys = np.packbits(keys, bitorder="little")
    sizes = keys.astype(np.int64) + 1
    ends = np.cumsum(sizes)
    data = np.zeros(int(ends[-1]), dtype=np.uint8)
    starts = ends - sizes
    data[starts] = (zz & 0xFF).astype(np.uint8)
    two = keys == 1
    data[starts[two] + 1] = (zz[two] >> 8).astype(np.uint8)
    return np.concatenate([out_keys, data])


def _zstd():
    from dynamont_amd.zstd_io import _libzstd
    L = _libzstd()
    L.ZSTD_compress.restype = C.c_size_t
    L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    L.ZSTD_compressBound.restype = C.c_size_t
    L.ZSTD_compressBound.argtypes = [C.c_size_t]
    L.ZSTD_decompress.restype = C.c_size_t
    L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_getFrameContentSize.restype = C.c_ulonglong
    L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
    return L


def vbz_decompress(blob: bytes, samples: int) -> np.ndarray:
    L = _zstd()
    n = int(L.ZSTD_getFrameContentSize(blob, len(blob)))
    if n >= (1 << 62):  # unknown / error sentinel: bound by the worst case of svb16
        n = (samples + 7) // 8 + 2 * samples
    dst = np.empty(max(n, 1), dtype=np.uint8)
    rc = L.ZSTD_decompress(dst.ctypes.data, len(dst), blob, len(blob))
    if L.ZSTD_isError(rc):
        raise ValueError("VBZ: zstd: " + L.ZSTD_getErrorName(rc).decode())
    return svb16_decode(dst[:rc], samples)


def vbz_compress(x: np.ndarray, level: int = 1) -> bytes:
    L = _zstd()
    enc = svb16_encode(x)
    cap = int(L.ZSTD_compressBound(len(enc)))
    dst = C.create_string_buffer(cap)
    rc = L.ZSTD_compress(dst, cap, enc.ctypes.data if len(enc) else None, len(enc), level)
    if L.ZSTD_isError(rc):
        raise ValueError("VBZ: zstd: " + L.ZSTD_getErrorName(rc).decode())
    return dst.raw[:rc]


# ---------------------------------------------------------------------------------------------
# the flatbuffers footer, by hand (two tables, no dependency)
# ---------------------------------------------------------------------------------------------
def _fb_field(buf, table: int, k: int) -> int:
    """Absolute position of field k of the table at `table`, 0 if absent."""
    vt = table - struct.unpack_from("<i", buf, table)[0]
    vt_size = struct.unpack_from("<H", buf, vt)[0]
    if 4 + 2 * k >= vt_size:
        return 0
    off = struct.unpack_from("<H", buf, vt + 4 + 2 * k)[0]
    return table + off if off else 0


def _fb_indirect(buf, pos: int) -> int:
    return pos + struct.unpack_from("<I", buf, pos)[0]


def _fb_string(buf, table: int, k: int) -> str:
    p = _fb_field(buf, table, k)
    if not p:
        return ""
    s = _fb_indirect(buf, p)
    n = struct.unpack_from("<I", buf, s)[0]
    return bytes(buf[s + 4:s + 4 + n]).decode()


def parse_footer(buf) -> dict:
    root = _fb_indirect(buf, 0)
    out = {"file_identifier": _fb_string(buf, root, 0), "software": _fb_string(buf, root, 1),
           "pod5_version": _fb_string(buf, root, 2), "contents": []}
    p = _fb_field(buf, root, 3)
    if p:
        vec = _fb_indirect(buf, p)
        n = struct.unpack_from("<I", buf, vec)[0]
        for i in range(n):
            t = _fb_indirect(buf, vec + 4 + 4 * i)
            def scalar(k, fmt, default=0, t=t):
                q = _fb_field(buf, t, k)
                return struct.unpack_from(fmt, buf, q)[0] if q else default
            out["contents"].append({"offset": scalar(0, "<q"), "length": scalar(1, "<q"),
                                    "format": scalar(2, "<h"), "content_type": scalar(3, "<h")})
    return out


def build_footer(file_identifier: str, software: str, pod5_version: str, contents: list[dict]) -> bytes:
    """Forward layout (every uoffset points ahead, vtables sit right before their tables)."""
    b = bytearray(4)  # root uoffset, patched below

    def align(n):
        while len(b) % n:
            b.append(0)

    def put_string(s: str) -> int:
        align(4)
        pos = len(b)
        raw = s.encode()
        b.extend(struct.pack("<I", len(raw)) + raw + b"\x00")
        return pos

    # Footer table: vtable (4 fields) then the table (soffset + 4 uoffsets)
    align(4)
    vt = len(b)
    b.extend(struct.pack("<HHHHHH", 12, 20, 4, 8, 12, 16))
    root = len(b)
    b.extend(struct.pack("<i", root - vt) + b"\x00" * 16)
    struct.pack_into("<I", b, 0, root)
    refs = [put_string(file_identifier), put_string(software), put_string(pod5_version)]
    align(4)
    vec = len(b)
    b.extend(struct.pack("<I", len(contents)) + b"\x00" * (4 * len(contents)))
    refs.append(vec)
    for k, target in enumerate(refs):
        struct.pack_into("<I", b, root + 4 + 4 * k, target - (root + 4 + 4 * k))
    for i, c in enumerate(contents):
        # 12-byte vtable, then the table = soffset(4) + pad(4) + offset(8) + length(8) + 2 shorts:
        # the table start must be a multiple of 8 for the int64 fields to be aligned in the buffer
        while (len(b) + 12) % 8:
            b.append(0)
        evt = len(b)
        b.extend(struct.pack("<HHHHHH", 12, 28, 8, 16, 24, 26))
        tab = len(b)
        b.extend(struct.pack("<i", tab - evt) + b"\x00" * 4 + struct.pack("<qqhh", c["offset"], c["length"],
                                                                         c.get("format", 0), c["content_type"]))
        struct.pack_into("<I", b, vec + 4 + 4 * i, tab - (vec + 4 + 4 * i))
    return bytes(b)


# ---------------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------------
_ARROW_NP = {"uint8": "<u1", "int8": "<i1", "uint16": "<u2", "int16": "<i2", "uint32": "<u4", "int32": "<i4", "uint64": "<u8",
             "int64": "<i8", "float": "<f4", "double": "<f8"}


def _primitive(arr) -> np.ndarray:
    """A fixed-width Arrow array as a NumPy view of its data buffer. (``to_numpy(zero_copy_only=False)`` imports pandas
    on its way -- 0.4 s of a run's start-up -- and these columns never hold nulls.)"""
    if arr.null_count:
        raise ValueError("POD5 column with nulls")
    dt = np.dtype(_ARROW_NP[str(arr.type)])  # (type.to_pandas_dtype() imports pandas too)
    return np.frombuffer(arr.buffers()[1], dtype=dt)[arr.offset:arr.offset + len(arr)]


class Pod5File:
    """``signal(read_id, calibrated)`` / ``close()`` over one .pod5 file (same surface as the
    synthetic container reader of ``pod5_io``)."""

    def __init__(self, path: str):
        import pyarrow as pa
        self.path = path
        self._fh = open(path, "rb")
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        mm = self._mm
        if len(mm) < 64 or mm[:8] != SIGNATURE or mm[-8:] != SIGNATURE:
            raise ValueError(f"{path}: not a POD5 file (signature)")
        self.section_marker = bytes(mm[8:24])
        flen_pos = len(mm) - 8 - 16 - 8
        flen = struct.unpack_from("<q", mm, flen_pos)[0]
        fstart = flen_pos - flen
        if flen <= 0 or fstart < 32 or mm[fstart - 8:fstart] != FOOTER_MAGIC:
            raise ValueError(f"{path}: POD5 footer not found")
        self.footer = parse_footer(memoryview(mm)[fstart:fstart + flen])
        self._tables = {}
        for c in self.footer["contents"]:
            if c["content_type"] in (CT_READS, CT_SIGNAL):
                view = memoryview(mm)[c["offset"]:c["offset"] + c["length"]]
                self._tables[c["content_type"]] = pa.ipc.open_file(pa.BufferReader(pa.py_buffer(view)))
        if CT_READS not in self._tables or CT_SIGNAL not in self._tables:
            raise ValueError(f"{path}: POD5 file without a reads or signal table")
        sig = self._tables[CT_SIGNAL]
        self._sig_rows = np.cumsum([0] + [sig.get_batch(i).num_rows for i in range(sig.num_record_batches)])
        self._sig_cache = (-1, None)
        self._sig_meta = {}
        self._index = None
        self.closed = False

    # -- reads table -------------------------------------------------------------------------
    def _build_index(self):
        """The reads table as arrays: ``_index`` maps the 16 id bytes to a row r; the read's signal rows are
        ``_sig_flat[_sig_first[r] : _sig_first[r] + _sig_count[r]]``, its calibration ``_cal_offset[r]``, ``_cal_scale[r]``
        (float32). Built once per file, without a Python object per signal row."""
        rd = self._tables[CT_READS]
        names = rd.schema.names
        if "calibration_offset" not in names or "calibration_scale" not in names:
            raise ValueError(f"{self.path}: reads table older than version 3 (no calibration_offset/scale columns)")
        keys, first, count, flat, cal_o, cal_s = [], [], [], [], [], []
        base = 0
        for bi in range(rd.num_record_batches):
            b = rd.get_batch(bi)
            ids = b.column(names.index("read_id"))
            ids = ids.storage if hasattr(ids, "storage") else ids
            raw = np.frombuffer(ids.buffers()[1], dtype=np.uint8)[ids.offset * 16:(ids.offset + len(ids)) * 16].tobytes()
            keys.extend(raw[i:i + 16] for i in range(0, len(raw), 16))
            sig = b.column(names.index("signal"))  # list<uint64>: offsets + flat values
            offs = np.frombuffer(sig.buffers()[1], dtype=np.int64 if str(sig.type).startswith("large_list") else np.int32)
            offs = offs[sig.offset:sig.offset + len(sig) + 1].astype(np.int64)
            vals = _primitive(sig.values).astype(np.int64)  # .values ignores the list's own offset: index with offs
            first.append(offs[:-1] + base)
            count.append(np.diff(offs))
            flat.append(vals)
            base += len(vals)
            cal_o.append(_primitive(b.column(names.index("calibration_offset"))).astype(np.float32))
            cal_s.append(_primitive(b.column(names.index("calibration_scale"))).astype(np.float32))
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)  # noqa: E731
        self._sig_first, self._sig_count, self._sig_flat = cat(first, np.int64), cat(count, np.int64), cat(flat, np.int64)
        self._cal_offset, self._cal_scale = cat(cal_o, np.float32), cat(cal_s, np.float32)
        self._index = dict(zip(keys, range(len(keys))))

    def _lookup(self, read_id: str):
        """(signal rows, calibration offset, calibration scale) of a read; KeyError when the file does not hold it"""
        if self._index is None:
            self._build_index()
        try:
            key = uuid.UUID(read_id).bytes
        except ValueError:
            raise KeyError(read_id) from None
        r = self._index.get(key)
        if r is None:
            raise KeyError(read_id)
        f = int(self._sig_first[r])
        return self._sig_flat[f:f + int(self._sig_count[r])], float(self._cal_offset[r]), float(self._cal_scale[r])

    @property
    def read_ids(self) -> list[str]:
        if self._index is None:
            self._build_index()
        return [str(uuid.UUID(bytes=k)) for k in self._index]

    # -- signal table ------------------------------------------------------------------------
    def _signal_row(self, row: int) -> np.ndarray:
        bi = int(np.searchsorted(self._sig_rows, row, side="right") - 1)
        if self._sig_cache[0] != bi:
            self._sig_cache = (bi, self._tables[CT_SIGNAL].get_batch(bi))
        b = self._sig_cache[1]
        i = row - int(self._sig_rows[bi])
        names = b.schema.names
        col = b.column(names.index("signal"))
        col = col.storage if hasattr(col, "storage") else col
        samples = b.column(names.index("samples"))[i].as_py()
        import pyarrow as pa
        if pa.types.is_large_binary(col.type) or pa.types.is_binary(col.type):
            return vbz_decompress(col[i].as_py(), samples)
        return np.asarray(col[i].as_py(), dtype=np.int16)  # uncompressed large_list<int16>

    def _batch_chunk_meta(self, bi: int):
        """(data address, offsets[int64], samples[uint32]) of signal-table batch bi when its signal column is VBZ
        (binary / large_binary), None when it holds uncompressed int16 lists. Zero copy: the Arrow buffers are views of
        the memory-mapped file, so the addresses stay valid until close()."""
        meta = self._sig_meta.get(bi)
        if meta is None:
            import pyarrow as pa
            b = self._tables[CT_SIGNAL].get_batch(bi)
            names = b.schema.names
            col = b.column(names.index("signal"))
            col = col.storage if hasattr(col, "storage") else col
            if pa.types.is_large_binary(col.type) or pa.types.is_binary(col.type):
                bufs = col.buffers()
                dt = np.int64 if pa.types.is_large_binary(col.type) else np.int32
                offs = np.frombuffer(bufs[1], dtype=dt)[col.offset:col.offset + len(col) + 1].astype(np.int64)
                samples = _primitive(b.column(names.index("samples"))).astype(np.uint32)
                meta = (int(bufs[2].address), offs, samples, b)  # b keeps the buffers alive
            else:
                meta = False
            self._sig_meta[bi] = meta
        return meta or None

    def signal_chunks(self, read_id: str):
        """The read's signal still compressed: (chunk addresses uint64[c], chunk bytes uint64[c], chunk samples
        uint32[c], calibration offset, calibration scale) for dyn_batch_align_vbz_async / dyn_vbz_decode, or None when
        the file stores uncompressed samples. Nothing is decoded or copied here."""
        rows, offset, scale = self._lookup(read_id)
        ptrs = np.empty(len(rows), dtype=np.uint64)
        nbytes = np.empty(len(rows), dtype=np.uint64)
        samples = np.empty(len(rows), dtype=np.uint32)
        for k, r in enumerate(rows):
            bi = int(np.searchsorted(self._sig_rows, r, side="right") - 1)
            meta = self._batch_chunk_meta(bi)
            if meta is None:
                return None
            base, offs, smp, _ = meta
            i = int(r) - int(self._sig_rows[bi])
            ptrs[k] = base + int(offs[i])
            nbytes[k] = int(offs[i + 1] - offs[i])
            samples[k] = smp[i]
        return ptrs, nbytes, samples, np.float32(offset), np.float32(scale)

    def _chunk_tables(self):
        """(address, bytes, samples) of EVERY row of the signal table, or None when a batch stores uncompressed samples"""
        t = getattr(self, "_chunk_tab", None)
        if t is None:
            ptrs, nbytes, samples = [], [], []
            for bi in range(len(self._sig_rows) - 1):
                meta = self._batch_chunk_meta(bi)
                if meta is None:
                    self._chunk_tab = t = False
                    return None
                base, offs, smp, _ = meta
                ptrs.append((offs[:-1] + base).astype(np.uint64))
                nbytes.append(np.diff(offs).astype(np.uint64))
                samples.append(smp)
            cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)  # noqa: E731
            self._chunk_tab = t = (cat(ptrs, np.uint64), cat(nbytes, np.uint64), cat(samples, np.uint32))
        return t or None

    def signal_chunks_batch(self, ids16: np.ndarray):
        """signal_chunks for a batch: ``ids16`` = (n, 16) uint8 read ids. Returns (found bool[n], ptrs uint64[c], nbytes
        uint64[c], samples uint32[c], read_off uint64[n + 1], cal_offset float32[n], cal_scale float32[n]) -- the flattened
        chunk tables dyn_batch_align_vbz_async takes; reads the file does not hold have ``found`` False and no chunks --
        or None when the file stores uncompressed samples."""
        if self._index is None:
            self._build_index()
        tab = self._chunk_tables()
        if tab is None:
            return None
        n = len(ids16)
        raw = np.ascontiguousarray(ids16, dtype=np.uint8).tobytes()
        get = self._index.get
        rows = np.fromiter((get(raw[i:i + 16], -1) for i in range(0, 16 * n, 16)), dtype=np.int64, count=n)
        found = rows >= 0
        r = np.where(found, rows, 0)
        cnt = np.where(found, self._sig_count[r], 0) if len(self._sig_count) else np.zeros(n, dtype=np.int64)
        read_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(cnt, out=read_off[1:])
        total = int(read_off[-1])
        flat = np.arange(total, dtype=np.int64) - np.repeat(read_off[:-1].astype(np.int64), cnt) + np.repeat(self._sig_first[r] if total else r, cnt)
        srow = self._sig_flat[flat]
        cal_o = np.where(found, self._cal_offset[r], np.float32(0)) if len(self._cal_offset) else np.zeros(n, dtype=np.float32)
        cal_s = np.where(found, self._cal_scale[r], np.float32(0)) if len(self._cal_scale) else np.zeros(n, dtype=np.float32)
        return found, tab[0][srow], tab[1][srow], tab[2][srow], read_off, cal_o.astype(np.float32), cal_s.astype(np.float32)

    def signal_adc(self, read_id: str):
        """(int16 ADC samples, calibration offset, calibration scale)"""
        rows, offset, scale = self._lookup(read_id)  # KeyError = read missing (pod5: missing_ok=False raises too)
        parts = [self._signal_row(int(r)) for r in rows]
        adc = np.concatenate(parts) if len(parts) != 1 else parts[0]
        return adc, np.float32(offset), np.float32(scale)

    def signal(self, read_id: str, calibrated: bool):
        adc, offset, scale = self.signal_adc(read_id)
        if calibrated:  # pod5 Calibration: picoampere = (adc + offset) * scale, float32
            return (adc.astype(np.float32) + offset) * scale
        return adc

    def close(self):
        if not self.closed:
            self._tables = {}
            self._sig_cache = (-1, None)
            try:
                self._mm.close()
            except BufferError:  # an Arrow buffer still references the map (e.g. the chunk tables that VbzSlice objects of
                pass             # batches in flight point into, via their `owner`): the GC releases it with the last of them
            self._fh.close()
            self.closed = True


# ---------------------------------------------------------------------------------------------
# writer (tests, synthetic datasets)
# ---------------------------------------------------------------------------------------------
def write_pod5(path: str, read_ids: list[str], adcs: list[np.ndarray], cal_offset, cal_scale,
               chunk_samples: int = 102400, batch_rows: int = 100, compress: bool = True) -> None:
    """Minimal table-version-3 file: signal table (VBZ or plain chunks of `chunk_samples`), a one-row
    run-info table, a reads table with the columns this package reads."""
    import pyarrow as pa

    def ext(name):
        return {b"ARROW:extension:name": name, b"ARROW:extension:metadata": b""}

    marker = uuid.uuid4().bytes
    sig_type = pa.large_binary() if compress else pa.large_list(pa.int16())
    sig_schema = pa.schema([pa.field("read_id", pa.binary(16), metadata=ext(b"minknow.uuid")),
                            pa.field("signal", sig_type, metadata=ext(b"minknow.vbz") if compress else None),
                            pa.field("samples", pa.uint32())],
                           metadata={b"MINKNOW:pod5_version": b"0.3.2", b"MINKNOW:software": b"dynamont_amd",
                                     b"MINKNOW:file_identifier": str(uuid.uuid4()).encode()})
    sig_ids, sig_vals, sig_n, read_rows = [], [], [], []
    packed = {}  # the same array object handed in again (a replicated synthetic dataset) is compressed once
    for rid, adc_in in zip(read_ids, adcs):
        adc = np.ascontiguousarray(adc_in, dtype=np.int16)
        rows = []
        for s in range(0, max(len(adc), 1), chunk_samples):
            part = adc[s:s + chunk_samples]
            rows.append(len(sig_ids))
            sig_ids.append(uuid.UUID(rid).bytes)
            if compress:
                key = (id(adc_in), s)
                if key not in packed:
                    packed[key] = vbz_compress(part)
                sig_vals.append(packed[key])
            else:
                sig_vals.append(part.tolist())
            sig_n.append(len(part))
        read_rows.append(rows)

    def ipc_bytes(schema, columns, n):
        sink = pa.BufferOutputStream()
        with pa.ipc.new_file(sink, schema) as w:
            for s in range(0, max(n, 1), batch_rows):
                arrays = [pa.array(c[s:s + batch_rows], type=f.type) for c, f in zip(columns, schema)]
                w.write_batch(pa.record_batch(arrays, schema=schema))
        return sink.getvalue().to_pybytes()

    signal_file = ipc_bytes(sig_schema, [sig_ids, sig_vals, sig_n], len(sig_ids))
    run_schema = pa.schema([pa.field("acquisition_id", pa.utf8()), pa.field("sample_rate", pa.uint16())])
    run_file = ipc_bytes(run_schema, [["synthetic"], [4000]], 1)
    reads_schema = pa.schema([pa.field("read_id", pa.binary(16), metadata=ext(b"minknow.uuid")),
                              pa.field("signal", pa.list_(pa.uint64())),
                              pa.field("read_number", pa.uint32()),
                              pa.field("num_samples", pa.uint64()),
                              pa.field("calibration_offset", pa.float32()),
                              pa.field("calibration_scale", pa.float32())],
                             metadata={b"MINKNOW:pod5_version": b"0.3.2"})
    reads_file = ipc_bytes(reads_schema, [[uuid.UUID(r).bytes for r in read_ids], read_rows,
                                          list(range(len(read_ids))), [len(a) for a in adcs],
                                          [float(v) for v in cal_offset], [float(v) for v in cal_scale]], len(read_ids))
    with open(path, "wb") as f:
        f.write(SIGNATURE + marker)
        contents = []
        for blob, ct in ((signal_file, CT_SIGNAL), (run_file, CT_RUN_INFO), (reads_file, CT_READS)):
            contents.append({"offset": f.tell(), "length": len(blob), "format": 0, "content_type": ct})
            f.write(blob + b"\x00" * (-len(blob) % 8) + marker)
        footer = build_footer(str(uuid.uuid4()), "dynamont_amd", "0.3.2", contents)
        footer += b"\x00" * (-len(footer) % 8)
        f.write(FOOTER_MAGIC + footer + struct.pack("<q", len(footer)) + marker + SIGNATURE)
