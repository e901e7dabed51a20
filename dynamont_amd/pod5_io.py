"""Raw-signal and basecall readers (counterpart of ``src/dynamont/pod5_io.py`` and of the pysam
loop in ``segment.py:189-258``).

``pod5`` / ``pysam`` are used when importable. They are not installed in the ROCm image and
there is no network, so ``.pod5`` files are otherwise read by the vendor-free
``dynamont_amd.pod5_native.Pod5File`` (pyarrow + libzstd + NumPy), and the same interface is also
served from a synthetic container that ``dynamont_amd.synth.write_dataset`` produces:

  raw:        ``<name>.dynraw.npz``  read_ids, offsets, adc (int16), cal_scale, cal_offset
              (pod5 convention: picoampere = (adc + offset) * scale)
  basecalls:  ``.bam`` / ``.sam`` through pysam, else through the vendor-free parsers of
              ``dynamont_amd.bam_io``; or ``<name>.dynbam.tsv``, one line per read with the BAM fields
              the reference reads: query_name, sequence, qs, pi, ns, ts, sp, fn, sm, sd ('*' = absent)
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import pod5 as _pod5
except Exception:  # noqa: BLE001
    _pod5 = None
try:  # pragma: no cover
    import pysam as _pysam
except Exception:  # noqa: BLE001
    _pysam = None


def _npz_member(path: str, z, name: str) -> np.ndarray:
    """One array of an .npz. A STORED (uncompressed) member is memory-mapped in place: np.load would copy it through
    zipfile and CRC-check every byte first (0.7 s for the 1.3 GB of samples of a 32 768-read container)."""
    import struct
    import zipfile
    try:
        with zipfile.ZipFile(path) as zf:
            info = zf.getinfo(name + ".npy")
            if info.compress_type != zipfile.ZIP_STORED:
                return z[name]
            with open(path, "rb") as f:
                f.seek(info.header_offset)
                hdr = f.read(30)
                n_name, n_extra = struct.unpack("<HH", hdr[26:30])
                f.seek(info.header_offset + 30 + n_name + n_extra)
                version = np.lib.format.read_magic(f)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if version == (1, 0)
                                         else np.lib.format.read_array_header_2_0(f))
                if fortran or dtype.hasobject:
                    return z[name]
                # a plain ndarray view: np.memmap's own __getitem__ / __array_finalize__ cost 4 us per slice
                return np.memmap(path, dtype=dtype, mode="r", offset=f.tell(), shape=shape).view(np.ndarray)
    except Exception:  # noqa: BLE001  (any surprise in the container layout: the ordinary loader)
        return z[name]


class SynthRawReader:
    def __init__(self, path: str):
        z = np.load(path, allow_pickle=False)
        self.path = path
        self._ids = {str(r): i for i, r in enumerate(z["read_ids"])}
        self._off = z["offsets"].tolist()
        self._adc = _npz_member(path, z, "adc")
        self._scale = z["cal_scale"].astype(np.float32)
        self._offset = z["cal_offset"].astype(np.float32)
        self.closed = False

    def close(self):
        self.closed = True

    def signal_adc(self, read_id: str):
        """(int16 ADC samples -- a view, no copy --, calibration offset, calibration scale)"""
        i = self._ids[read_id]  # KeyError = read missing (pod5: missing_ok=False raises too)
        return self._adc[self._off[i]:self._off[i + 1]], self._offset[i], self._scale[i]

    def signal(self, read_id: str, calibrated: bool):
        adc, offset, scale = self.signal_adc(read_id)
        if calibrated:
            return (adc.astype(np.float32) + offset) * scale
        return adc


def open_pod5(path: str):
    """pod5_io.py:3-4"""
    if path.endswith(".npz"):
        return SynthRawReader(path)
    if _pod5 is None:
        from dynamont_amd.pod5_native import Pod5File  # vendor-free reader (pyarrow + libzstd)
        return Pod5File(path)
    return _pod5.Reader(path)


def get_signal(reader, read_id: str, calibrated: bool = False):
    """pod5_io.py:6-16: ``signal_pa`` when calibrated, raw ADC ``signal`` otherwise."""
    if not hasattr(reader, "reads"):  # SynthRawReader / pod5_native.Pod5File
        return reader.signal(read_id, calibrated)
    record = next(reader.reads(selection=[read_id], missing_ok=False, preload={"samples"}))
    return record.signal_pa if calibrated else record.signal


def get_signal_adc(reader, read_id: str):
    """(int16 ADC samples, calibration offset, calibration scale) of a read: picoampere = (float32(adc) + offset) *
    scale is what ``signal_pa`` computes (pod5_io.py:6-16); handing the ADC counts on lets that happen on the device."""
    if hasattr(reader, "signal_adc"):  # SynthRawReader / pod5_native.Pod5File
        return reader.signal_adc(read_id)
    record = next(reader.reads(selection=[read_id], missing_ok=False, preload={"samples"}))  # pragma: no cover
    return record.signal, np.float32(record.calibration.offset), np.float32(record.calibration.scale)


class VbzSlice:
    """A read's [start:end) signal slice as POD5 chunks that are still compressed (VBZ): what ``prepare_job_raw`` hands on
    when the reader can point at the chunks in the memory-mapped file. ``len()`` = samples of the slice, clipped to the
    read like ``signal[start:end]`` clips; decoding happens in the library's helper threads
    (dyn_batch_align_vbz_async)."""
    dtype = np.dtype(np.int16)
    vbz = True

    def __init__(self, ptrs, nbytes, samples, start: int, end: int, owner=None):
        self.owner = owner  # the reader whose memory map the addresses point into: alive as long as this slice is
        total = int(samples.sum())
        s = min(max(int(start), 0), total)
        e = min(max(int(end), s), total)
        self.ptrs, self.nbytes, self.samples, self.start, self.length = ptrs, nbytes, samples, s, e - s

    def __len__(self):
        return self.length


def get_signal_chunks(reader, read_id: str):
    """(chunk addresses, chunk bytes, chunk samples, calibration offset, calibration scale) when the reader can hand
    out the read's signal without decoding it (pod5_native.Pod5File on a VBZ file), else None."""
    fn = getattr(reader, "signal_chunks", None)
    return fn(read_id) if fn is not None else None


class BasecallRecord:
    """The subset of pysam.AlignedSegment the reference touches (segment.py:222-245)."""

    def __init__(self, query_name, query_sequence, tags):
        self.query_name = query_name
        self.query_sequence = query_sequence
        self._tags = tags

    def has_tag(self, t):
        return t in self._tags

    def get_tag(self, t):
        return self._tags[t]  # KeyError like pysam


def iter_basecalls(path: str):
    """Yield records in file order: pysam for .bam/.sam, the TSV container otherwise."""
    if path.endswith((".bam", ".sam")):
        if _pysam is not None:
            with _pysam.AlignmentFile(path, "r" if path.endswith(".sam") else "rb", check_sq=False) as f:
                for rec in f.fetch(until_eof=True):
                    yield rec
            return
        from dynamont_amd import bam_io  # vendor-free parsers (pysam/htslib are not in the ROCm image)
        yield from (bam_io.iter_sam(path) if path.endswith(".sam") else bam_io.iter_bam(path))
        return
    with open(path) as f:
        cols = f.readline().rstrip("\n").split("\t")
        for line in f:
            if not line.strip():
                continue
            p = dict(zip(cols, line.rstrip("\n").split("\t")))
            tags = {}
            for k, conv in (("qs", float), ("pi", str), ("ns", int), ("ts", int), ("sp", int), ("fn", str),
                            ("sm", float), ("sd", float)):
                if p.get(k, "*") != "*":
                    tags[k] = conv(p[k])
            yield BasecallRecord(p["query_name"], p["sequence"], tags)
