"""The native job reader (csrc/bam_reader.cpp, dyn_bam_*) and the column form of the workers' preparation
(segment.prepare_job_columns) against the read-by-read Python path they replace (reference:
src/dynamont/segmentation/segment.py:141-158 worker preparation, :189-258 generate_jobs). CPU only: no compute call."""
import gzip
import os
import struct
import uuid

import numpy as np
import pytest

from dynamont_amd import bam_io, synth
from dynamont_amd.segmentation import segment as seg

from test_format_pinning import _blocks, _records, spec_bam, spec_bgzf_block, spec_record  # the BAM assembled from the SAM specification

pytestmark = pytest.mark.usefixtures("native_lib")   # the reader lives in libdynamont_mi.so (built on demand; no compute call)


def python_jobs(data_path, bam, minq=0.0, rank=0, world=1):
    os.environ["DYN_PY_BAM"] = "1"
    try:
        return list(seg.generate_jobs(data_path, bam, minq, rank, world))
    finally:
        del os.environ["DYN_PY_BAM"]


def orient(read):
    read = read[::-1]
    return read if read.startswith(seg.POLYA) else seg.POLYA + read


@pytest.mark.parametrize("cuts", [list(range(50000, 300000, 50000)), [5000, 70000], list(range(777, 150000, 777))])
def test_native_reader_equals_the_python_parsers_on_the_specification_bam(tmp_path, cuts):
    """records straddling BGZF blocks of any size, aligned records with CIGARs, every tag type, split reads (pi + sp)"""
    recs, want = _records()
    path = str(tmp_path / "spec.bam")
    open(path, "wb").write(_blocks(spec_bam(recs, refs=[("chr1", 1000000)]), cuts))
    for minq in (0.0, 10.0):
        ref = python_jobs("/data", path, minq)
        for batch_reads in (1, 7, 4096):
            for rna in (False, True):
                got = []
                r = bam_io.NativeBamJobs(path, rna=rna, pad=seg.POLYA, min_qual=minq)
                while (jb := r.next(batch_reads)) is not None:
                    assert jb.n <= batch_reads
                    for i in range(jb.n):
                        got.append(("/data/" + jb.files[jb.file_id[i]], float(jb.shift[i]), float(jb.scale[i]), int(jb.start[i]), int(jb.end[i]),
                                    jb.read(i), jb.name(i), jb.sid(i)))
                        if jb.uuid_ok[i]:
                            assert jb.uuid[i].tobytes() == uuid.UUID(jb.sid(i)).bytes
                assert r.skipped == len(want) - len(ref)
                r.close()
                expect = [j[:5] + (orient(j[5]),) + j[6:] for j in ref] if rna else ref
                assert got == expect
        assert list(seg.generate_jobs("/data", path, minq)) == ref       # generate_jobs itself rides on the native reader


def test_rank_shares_partition_the_filtered_jobs(tmp_path):
    recs, want = _records()
    path = str(tmp_path / "spec.bam")
    open(path, "wb").write(_blocks(spec_bam(recs), [40000, 90000, 140000, 190000]))
    full = python_jobs("/data", path, 10.0)
    for world in (2, 3, 8):
        shares = [list(seg.generate_jobs("/data", path, 10.0, rank, world)) for rank in range(world)]
        assert [python_jobs("/data", path, 10.0, rank, world) for rank in range(world)] == shares
        for rank, share in enumerate(shares):
            assert share == full[rank::world]


def test_uuid_names_rna_pad_and_bases(tmp_path):
    ids = [str(uuid.UUID(int=(i + 1) * 0x1234567890abcdef1234567 % 2 ** 128)) for i in range(6)]
    seqs = ["ACGTTGCA" * 5, "TTTT" + "A" * 9, "A" * 9, "ACG", "", "GATTACA" + "A" * 9]
    recs = [(ids[i], s, {"qs": 20.0, "ns": 5000 + i, "ts": 10, "fn": "x.pod5", "sm": 90.0, "sd": 15.0}) for i, s in enumerate(seqs)]
    path = str(tmp_path / "u.bam")
    bam_io.write_bam(path, recs)
    jb = bam_io.NativeBamJobs(path, rna=True, pad=seg.POLYA).next(100)
    assert jb.n == 6 and jb.uuid_ok.all() and list(jb.bases) == [len(s) for s in seqs]
    for i, s in enumerate(seqs):
        assert jb.read(i) == orient(s) and jb.uuid[i].tobytes() == uuid.UUID(ids[i]).bytes
    sub = jb.take(np.array([1, 4, 5]))
    assert [sub.name(i) for i in range(3)] == [ids[1], ids[4], ids[5]] and [sub.read(i) for i in range(3)] == [orient(seqs[k]) for k in (1, 4, 5)]
    assert list(sub.bases) == [len(seqs[k]) for k in (1, 4, 5)] and np.array_equal(sub.uuid, jb.uuid[[1, 4, 5]])


def test_missing_tag_is_a_key_error_and_damage_is_reported(tmp_path):
    def rec(tags):
        return spec_record("r0", "ACGT", np.zeros(4, np.uint8), 4, [], tags)
    ftag = lambda t, v: t.encode() + b"f" + struct.pack("<f", v)  # noqa: E731
    full = ftag("qs", 12.0) + b"nsI" + struct.pack("<I", 100) + b"tsC\x05" + b"fnZx.pod5\0" + ftag("sm", 1.0) + ftag("sd", 2.0)
    for missing, tags in (("qs", full[7:]), ("sd", full[:-7]), ("ns", full[:7] + full[14:])):
        path = str(tmp_path / f"no_{missing}.bam")
        open(path, "wb").write(_blocks(spec_bam([rec(tags)]), []))
        with pytest.raises(KeyError, match=f"tag '{missing}' not present"):
            bam_io.NativeBamJobs(path).next(10)
        with pytest.raises(KeyError):
            python_jobs("/d", path)
    # f5 instead of fn (segment.py:243), sp absent = 0
    path = str(tmp_path / "f5.bam")
    open(path, "wb").write(_blocks(spec_bam([rec(full.replace(b"fnZ", b"f5Z"))]), []))
    jb = bam_io.NativeBamJobs(path).next(10)
    assert jb.files == ["x.pod5"] and jb.start[0] == 5 and jb.end[0] == 100
    good = _blocks(spec_bam([rec(full)] * 50), [300])
    for name, data, msg in (("crc", good[:40] + bytes([good[40] ^ 0x55]) + good[41:], "corrupt BGZF block|malformed|truncated"),
                            ("cut", good[:len(good) // 2], "truncated"),
                            ("gz", gzip.compress(b"BAM\1" + os.urandom(64)), "not a BGZF block"),
                            ("magic", spec_bgzf_block(b"BAX\1" + bytes(8)), "not a BAM file")):
        path = str(tmp_path / (name + ".bam"))
        open(path, "wb").write(data)
        with pytest.raises(ValueError, match=msg):
            r = bam_io.NativeBamJobs(path)
            while r.next(10) is not None:
                pass
    with pytest.raises(ValueError, match="cannot open"):
        bam_io.NativeBamJobs(str(tmp_path / "absent.bam"))


def test_column_preparation_equals_the_per_read_preparation(models, tmp_path):
    """prepare_job_columns over two .pod5 files in one batch, a read neither file holds and a read with a damaged id:
    chunk tables, slices, calibration and error lines are those of prepare_job_raw read by read"""
    from conftest import model_for
    _, mean, sd = synth.read_model_file(model_for(models, "rna004"))
    reads_a = synth.make_reads(41, 6, "rna004", mean, sd, (60, 120))
    reads_b = synth.make_reads(43, 5, "rna004", mean, sd, (60, 120))
    _, bam_a, _ = synth.write_dataset(str(tmp_path), "a", reads_a, "rna004", seed=3, container="pod5", basecalls="bam", pod5_chunk_samples=700)
    _, bam_b, _ = synth.write_dataset(str(tmp_path), "b", reads_b, "rna004", seed=4, container="pod5", basecalls="bam", pod5_chunk_samples=500)
    recs = []
    os.environ["DYN_PY_BAM"] = "1"
    try:
        from dynamont_amd.pod5_io import iter_basecalls
        ra, rb = list(iter_basecalls(bam_a)), list(iter_basecalls(bam_b))
    finally:
        del os.environ["DYN_PY_BAM"]
    for k in range(6):   # interleave the two files' reads
        recs.append(ra[k])
        if k < 5:
            recs.append(rb[k])
    as_tuple = lambda r, **kw: (kw.get("name", r.query_name), r.query_sequence, {**r._tags, **kw.get("tags", {})})  # noqa: E731
    mixed = [as_tuple(r) for r in recs]
    mixed.insert(3, as_tuple(ra[0], name=str(uuid.uuid4())))                    # a read no file holds
    mixed.insert(7, as_tuple(rb[1], name="not-a-uuid"))                          # an id uuid.UUID() refuses
    mixed.insert(9, as_tuple(ra[2], tags={"sm": 500.0}))                         # shift > 400: ADC counts, no calibration
    mixed.append(as_tuple(ra[1], tags={"ts": 10 ** 7}))                          # a slice behind the read's end: empty
    bam = str(tmp_path / "mixed.bam")
    bam_io.write_bam(bam, mixed)
    jb = bam_io.NativeBamJobs(bam, rna=True, pad=seg.POLYA).next(1000)
    assert jb.n == len(mixed) and len(jb.files) == 2
    errors = []
    jb2, chunks, raw_off, cal, owners = seg.prepare_job_columns(jb, str(tmp_path), errors.append)
    ptrs, nbytes, samples, read_off, skip = chunks
    ref_errors, k = [], 0
    for job in python_jobs(str(tmp_path), bam):
        try:
            raw, read, c = seg.prepare_job_raw(job, True)
        except Exception as error:  # noqa: BLE001
            ref_errors.append(f"error: worker, {error}\tN: {len(job[5])}\tRid: {job[6]}\tSid: {job[7]}")
            continue
        sl = slice(int(read_off[k]), int(read_off[k + 1]))
        assert np.array_equal(raw.ptrs, ptrs[sl]) and np.array_equal(raw.nbytes, nbytes[sl]) and np.array_equal(raw.samples, samples[sl])
        assert raw.start == skip[k] and len(raw) == raw_off[k + 1] - raw_off[k]
        assert jb2.read(k) == read and jb2.name(k) == job[6] and jb2.sid(k) == job[7] and jb2.shift[k] == job[1]
        assert (c is not None) == bool(cal[2][k]) == (job[1] <= 400)
        if c is not None:
            assert c == (cal[0][k], cal[1][k])
        k += 1
    assert k == jb2.n == len(mixed) - 2 and errors == ref_errors and len(errors) == 2
    assert raw_off[-1] - raw_off[-2] == 0 and (np.diff(read_off) >= 1).all()
    seg.close_raw_cache()
