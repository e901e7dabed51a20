"""Vendor-free POD5 reader/writer (dynamont_amd/pod5_native.py): StreamVByte/zigzag/delta stage,
hand-rolled flatbuffers footer, Arrow tables, and the pod5_io surface the CLIs use. Round-trip
tests: the `pod5` package is not available in the build image (see the module docstring)."""
import os
import struct
import uuid

import numpy as np
import pytest

from dynamont_amd import pod5_io, pod5_native as P, synth
from dynamont_amd.segmentation import segment as seg
from conftest import model_for

pytestmark = pytest.mark.usefixtures("native_lib")   # BAM basecalls are read by the library (built on demand; no compute call)


def test_svb16_known_bytes():
    # samples 5, 5, 300, 299, -1 -> deltas 5, 0, 295, -1, -300 -> zigzag 10, 0, 590, 1, 599
    x = np.array([5, 5, 300, 299, -1], dtype=np.int16)
    enc = P.svb16_encode(x)
    # one key byte: values 2 and 4 need two bytes -> bits 2 and 4 -> 0b00010100
    assert enc.tobytes() == bytes([0b00010100, 10, 0, 590 & 0xFF, 590 >> 8, 1, 599 & 0xFF, 599 >> 8])
    assert np.array_equal(P.svb16_decode(enc, 5), x)


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 4097, 70001])
def test_svb16_and_vbz_roundtrip(n):
    rng = np.random.default_rng(n)
    for x in (rng.integers(-32768, 32768, n).astype(np.int16),          # wrap-around deltas
              rng.normal(500, 30, n).astype(np.int16),                  # signal-like: mostly 1-byte codes
              np.full(n, -32768, dtype=np.int16)):
        assert np.array_equal(P.svb16_decode(P.svb16_encode(x), n), x)
        assert np.array_equal(P.vbz_decompress(P.vbz_compress(x), n), x)
    with pytest.raises(ValueError):
        P.svb16_decode(P.svb16_encode(np.arange(100, dtype=np.int16) * 300)[:-3], 100)


def test_footer_flatbuffer_roundtrip_and_alignment():
    contents = [{"offset": 24, "length": 1000, "format": 0, "content_type": P.CT_SIGNAL},
                {"offset": 2048, "length": 12345678901, "format": 0, "content_type": P.CT_READS},
                {"offset": 1 << 40, "length": 7, "format": 0, "content_type": P.CT_RUN_INFO}]
    for ident in ("", "a", "0123456789abcdef-file-id"):
        fb = P.build_footer(ident, "software x", "0.3.2", contents)
        got = P.parse_footer(memoryview(fb))
        assert got == {"file_identifier": ident, "software": "software x", "pod5_version": "0.3.2", "contents": contents}
        # every int64 field of an EmbeddedFile table is 8-byte aligned in the buffer
        for c in contents:
            pos = fb.index(struct.pack("<qq", c["offset"], c["length"]))
            assert pos % 8 == 0


@pytest.mark.parametrize("compress", [True, False])
def test_file_roundtrip(tmp_path, compress):
    rng = np.random.default_rng(4)
    ids = [str(uuid.UUID(int=int(v))) for v in rng.integers(1, 2 ** 62, 6)]
    adcs = [rng.integers(-500, 3000, n).astype(np.int16) for n in (10, 0, 5000, 123456, 3, 100000)]
    off, sc = rng.uniform(-300, -200, 6), rng.uniform(0.1, 0.2, 6)
    path = str(tmp_path / "t.pod5")
    P.write_pod5(path, ids, adcs, off, sc, chunk_samples=50000, batch_rows=3, compress=compress)
    raw = open(path, "rb").read()
    assert raw[:8] == P.SIGNATURE and raw[-8:] == P.SIGNATURE and raw.count(raw[8:24]) == 5  # section markers
    f = pod5_io.open_pod5(path)                          # no `pod5` package here -> the vendor-free reader
    assert isinstance(f, P.Pod5File) and sorted(f.read_ids) == sorted(ids)
    assert {c["content_type"] for c in f.footer["contents"]} == {P.CT_SIGNAL, P.CT_RUN_INFO, P.CT_READS}
    for i in (3, 0, 5, 1, 2, 4, 3):                      # random access across signal-table batches
        a = pod5_io.get_signal(f, ids[i], calibrated=False)
        assert a.dtype == np.int16 and np.array_equal(a, adcs[i])
        pa_ = pod5_io.get_signal(f, ids[i], calibrated=True)
        assert pa_.dtype == np.float32
        assert np.array_equal(pa_, (adcs[i].astype(np.float32) + np.float32(off[i])) * np.float32(sc[i]))
    with pytest.raises(KeyError):
        f.signal(str(uuid.uuid4()), False)
    with pytest.raises(KeyError):
        f.signal("not-a-uuid", False)
    f.close()
    bad = tmp_path / "bad.pod5"
    bad.write_bytes(raw[:-8] + b"XXXXXXXX")
    with pytest.raises(ValueError):
        P.Pod5File(str(bad))


def test_cli_jobs_from_pod5_equal_jobs_from_npz(models, tmp_path):
    """generate_jobs / prepare_job over a .pod5 container give the same samples as over the .npz one."""
    pore = "rna004"
    _, mean, sd = synth.read_model_file(model_for(models, pore))
    reads = synth.make_reads(31, 5, pore, mean, sd, (40, 90))
    _, bam_a, expected = synth.write_dataset(str(tmp_path / "a"), "ds", reads, pore, seed=3)
    raw_b, bam_b, _ = synth.write_dataset(str(tmp_path / "b"), "ds", reads, pore, seed=3, container="pod5")
    assert raw_b.endswith(".pod5") and os.path.getsize(raw_b) > 0
    ja, jb = list(seg.generate_jobs(str(tmp_path / "a"), bam_a, 0)), list(seg.generate_jobs(str(tmp_path / "b"), bam_b, 0))
    assert len(ja) == len(jb) == 5
    for a, b in zip(ja, jb):
        sa, ra = seg.prepare_job(a, True)
        sb, rb = seg.prepare_job(b, True)
        assert ra == rb and np.array_equal(sa, sb)
        (xa, _, ca), (xb, _, cb) = seg.prepare_job_raw(a, True), seg.prepare_job_raw(b, True)
        # .npz: a view of the int16 samples; .pod5: the still-compressed chunks (decoded by the library, see below)
        assert xa.dtype == xb.dtype == np.int16 and isinstance(xb, pod5_io.VbzSlice) and len(xa) == len(xb)
        assert ca is not None and ca == cb  # shift <= 400: the calibrated signal, calibration handed on for the device
    seg.close_raw_cache()


def test_native_vbz_decoder_equals_the_numpy_one(native_lib):
    """dyn_vbz_decode (what the asynchronous engine's helper threads run for .pod5 input) against the NumPy decoder of
    pod5_native on chunk sizes around the 8-value key groups, spikes that need two-byte deltas, and a truncated chunk."""
    import ctypes as C
    rng = np.random.default_rng(11)
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 100, 4096, 20011, 102400):
        x = rng.normal(500, 120, n).astype(np.int16)
        if n > 10:
            x[rng.integers(0, n, max(1, n // 50))] = rng.integers(-32768, 32767)
        blob = P.vbz_compress(x)
        out = np.empty(max(n, 1), dtype=np.int16)
        err = C.create_string_buffer(256)
        assert native_lib.dyn_vbz_decode(blob, len(blob), n, out.ctypes.data, err, 256) == 0, err.value
        assert np.array_equal(out[:n], x) and np.array_equal(P.vbz_decompress(blob, n), x)
    assert native_lib.dyn_vbz_decode(blob[:len(blob) // 2], len(blob) // 2, n, out.ctypes.data, err, 256) != 0
    assert err.value.startswith(b"VBZ:")


def test_signal_chunks_point_at_the_reads_compressed_signal(models, tmp_path, native_lib):
    """Pod5File.signal_chunks -> VbzSlice (prepare_job_raw on a .pod5 file): decoding the chunks natively and cutting
    [start:end) gives exactly what signal_adc()[start:end] gives; the calibration travels with it."""
    import ctypes as C
    pore = "rna004"
    _, mean, sd = synth.read_model_file(model_for(models, pore))
    reads = synth.make_reads(77, 6, pore, mean, sd, (100, 1500))
    _, bam, _ = synth.write_dataset(str(tmp_path / "p"), "ds", reads, pore, seed=4, container="pod5")
    for job in seg.generate_jobs(str(tmp_path / "p"), bam, 0):
        raw, _, cal = seg.prepare_job_raw(job, True)
        assert isinstance(raw, pod5_io.VbzSlice) and raw.owner is not None
        adc, off, sc = pod5_io.get_signal_adc(seg.get_raw(job[0]), job[7])
        parts = []
        for p, nb, sm in zip(raw.ptrs, raw.nbytes, raw.samples):
            buf = np.empty(int(sm), dtype=np.int16)
            err = C.create_string_buffer(256)
            assert native_lib.dyn_vbz_decode(C.c_void_p(int(p)), int(nb), int(sm), buf.ctypes.data, err, 256) == 0, err.value
            parts.append(buf)
        whole = np.concatenate(parts)
        assert np.array_equal(whole[raw.start:raw.start + len(raw)], adc[job[3]:job[4]]) and cal == (off, sc)
    seg.close_raw_cache()
