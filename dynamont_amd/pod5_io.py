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


class SynthRawReader:
    def __init__(self, path: str):
        z = np.load(path, allow_pickle=False)
        self.path = path
        self._ids = {str(r): i for i, r in enumerate(z["read_ids"])}
        self._off = z["offsets"]
        self._adc = z["adc"]
        self._scale = z["cal_scale"]
        self._offset = z["cal_offset"]
        self.closed = False

    def close(self):
        self.closed = True

    def signal(self, read_id: str, calibrated: bool):
        i = self._ids[read_id]  # KeyError = read missing (pod5: missing_ok=False raises too)
        adc = self._adc[self._off[i]:self._off[i + 1]]
        if calibrated:
            return (adc.astype(np.float32) + np.float32(self._offset[i])) * np.float32(self._scale[i])
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
