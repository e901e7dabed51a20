"""GPU (-m gpu): the CLI counterparts end to end on the synthetic container --
dynamont-resquiggle (CSV bytes, .errors lines) and dynamont-train (model files, params.csv) --
against the same pipeline evaluated with the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import model_for
from dynamont_amd import synth, zstd_io
from dynamont_amd.segmentation import segment as seg
from dynamont_amd.segmentation import train as trn
from dynamont_amd.segmentation import utils as U
from oracle.pyoracle import Oracle

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("native_lib", "oracle_built")]


@pytest.mark.parametrize("pore", ["rna004", "dna_r10_260bps"])
def test_resquiggle_cli_end_to_end(models, tmp_path, pore):
    model = model_for(models, pore)
    pid, rna, k = synth.PORES[pore]
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(71, 9, pore, mean, sd, (60, 260))
    raw, bam, expected = synth.write_dataset(str(tmp_path / "in"), "ds", reads, pore, seed=9)
    # corrupt one basecall with an N and make one signal too short -> per-read error lines
    lines = open(bam).read().splitlines()
    f = lines[3].split("\t"); f[1] = f[1][:20] + "N" + f[1][21:]; lines[3] = "\t".join(f)
    f = lines[6].split("\t"); f[4] = str(int(f[5]) + 30); lines[6] = "\t".join(f)   # ns = ts + 30 samples
    open(bam, "w").write("\n".join(lines) + "\n")
    out = tmp_path / "out" / "res.csv"
    seg.main(["-r", str(tmp_path / "in"), "-b", bam, "-o", str(out), "--mode", "basic", "-p", pore,
              "--model_path", model, "--batch-reads", "4"])
    got = zstd_io.decompress(open(str(out) + ".zst", "rb").read())
    # device preprocessing (default) and host preprocessing produce the same bytes
    out2 = tmp_path / "out2" / "res.csv"
    seg.main(["-r", str(tmp_path / "in"), "-b", bam, "-o", str(out2), "--mode", "basic", "-p", pore,
              "--model_path", model, "--batch-reads", "5", "--host-preprocess"])
    assert zstd_io.decompress(open(str(out2) + ".zst", "rb").read()) == got
    assert open(str(tmp_path / "out2" / "res.errors")).read() == open(str(tmp_path / "out" / "res.errors")).read()
    orc = Oracle(model, pid)
    want = [seg.CSV_HEADER]
    errors = []
    for job in seg.generate_jobs(str(tmp_path / "in"), bam, 0):
        signal, read = seg.prepare_job(job, rna)
        _, _, _, start, _, _, readid, signalid = job
        try:
            res = orc.align(signal, read, True)
            res["polishes"] = [""] * len(res["states"])
            want.append(U.segmentation_to_string(res, readid, signalid, start, len(signal) + start, read, k, rna))
        except RuntimeError as e:
            errors.append(f"error: native, {e}\tT: {len(signal)}\tN: {len(read)}\tRid: {readid}\tSid: {signalid}")
    seg.close_raw_cache()
    want = b"".join(want)
    assert len(errors) == 2
    assert open(str(tmp_path / "out" / "res.errors")).read().splitlines() == errors
    # integer / text columns byte-identical; posterior column within 1e-4 (formatted %.6f)
    gl, wl = got.decode().splitlines(), want.decode().splitlines()
    assert len(gl) == len(wl) and gl[0] == wl[0]
    for a, b in zip(gl[1:], wl[1:]):
        fa, fb = a.split(","), b.split(",")
        assert fa[:8] == fb[:8] and fa[9] == fb[9]
        assert abs(float(fa[8]) - float(fb[8])) <= 1e-4
    assert sum(a == b for a, b in zip(gl, wl)) >= 0.99 * len(gl)


@pytest.mark.parametrize("aggregate", ["pooled", "window-mean"])
def test_train_cli(models, tmp_path, aggregate):
    pore = "rna002"
    # the reference skips per-read updates whose polyA k-mer mean is < 0.5 (train.py:198-199); real
    # models have it near 0.9, the seeded synthetic one does not, so pin it for this test
    kmers, mean, sd = synth.read_model_file(model_for(models, pore))
    mean[kmers.index("AAAAA")] = 1.2
    model = str(tmp_path / "polyA_ok.model")
    U.write_kmer_model(model, {k_: (float(m), float(s)) for k_, m, s in zip(kmers, mean, sd)})
    reads = synth.make_reads(72, 8, pore, mean, sd, (80, 200))
    raw, bam, expected = synth.write_dataset(str(tmp_path / "in"), "tr", reads, pore, seed=4)
    # qs filter (default 10) would drop some reads: raise all qualities
    lines = open(bam).read().splitlines()
    for i in range(1, len(lines)):
        f = lines[i].split("\t"); f[2] = "15.0"; lines[i] = "\t".join(f)
    open(bam, "w").write("\n".join(lines) + "\n")
    outdir = tmp_path / f"train_{aggregate}"
    trn.main(["-r", str(tmp_path / "in"), "-b", bam, "-o", str(outdir), "-p", pore, "--model_path", model,
              "--batch_size", "4", "--max_batches", "2", "--aggregate", aggregate, "--no-timestamp"])
    files = sorted(os.listdir(outdir))
    assert files == ["params.csv", "trained_0_0.model", "trained_0_1.model", "trained_0_2.model"]
    rows = open(outdir / "params.csv").read().splitlines()
    assert rows[0] == "epoch,batch,read,e1,m1,e2,Zchange"
    assert rows[1].startswith("0,1,4,") and rows[2].startswith("0,2,8,")
    assert all(len(r.split(",")) == 7 for r in rows[1:])
    assert float(rows[1].split(",")[-1]) > 0       # an EM step on its own batch raises the likelihood
    # host preprocessing (NumPy float32 arithmetic) gives the very same model files
    outdir2 = tmp_path / f"train_{aggregate}_host"
    trn.main(["-r", str(tmp_path / "in"), "-b", bam, "-o", str(outdir2), "-p", pore, "--model_path", model,
              "--batch_size", "4", "--max_batches", "2", "--aggregate", aggregate, "--no-timestamp", "--host-preprocess"])
    for f in ("trained_0_1.model", "trained_0_2.model"):
        assert open(outdir / f).read() == open(outdir2 / f).read()
    m0, m1 = U.read_kmer_model(str(outdir / "trained_0_0.model")), U.read_kmer_model(str(outdir / "trained_0_1.model"))
    assert list(m0) == list(m1) and m0 != m1
    if aggregate == "pooled":
        # first batch against the oracle's pooled statistics
        orc = Oracle(model, 0)
        K = orc.num_kmers
        w, s1, s2 = np.zeros(K), np.zeros(K), np.zeros(K)
        items = [it for it in trn.read_items(str(tmp_path / "in"), bam, pore, 10.0) if not isinstance(it, str)][:4]
        for signal, seq, _ in items:
            t = orc.train(np.asarray(signal, dtype=np.float64), seq, dense=False)
            w += t["weight"]; s1 += t["sum"]; s2 += t["sumsq"]
        hit = np.nonzero(w > 0)[0]
        for c in hit[:50]:
            name = U.kmer_of_code(int(c), 5, True)
            assert abs(m1[name][0] - s1[c] / w[c]) <= 1e-9
            var = max(s2[c] / w[c] - (s1[c] / w[c]) ** 2, 1e-12)
            assert abs(m1[name][1] - np.sqrt(var)) <= 1e-7


def test_resquiggle_cli_on_a_pod5_file_equals_the_npz_container(models, tmp_path):
    """The same reads from a .pod5 file (VBZ chunks handed to dyn_batch_align_vbz_async still compressed, decoded by the
    library's helper threads; UUID read ids) and from the .npz container: the rows must be byte-identical apart from the
    id columns, for reads that span several chunks too, and a read whose [start:end) slice is cut short still fails
    with the reference's message."""
    pore = "rna004"
    model = model_for(models, pore)
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(72, 11, pore, mean, sd, (60, 900))
    rows = {}
    for container in ("npz", "pod5"):
        d = tmp_path / container
        raw, bam, _ = synth.write_dataset(str(d / "in"), "ds", reads, pore, seed=9, container=container, pod5_chunk_samples=1000)
        lines = open(bam).read().splitlines()
        f = lines[5].split("\t"); f[4] = str(int(f[5]) + 30); lines[5] = "\t".join(f)   # ns = ts + 30 samples: too short
        open(bam, "w").write("\n".join(lines) + "\n")
        out = d / "out" / "res.csv"
        seg.main(["-r", str(d / "in"), "-b", bam, "-o", str(out), "--mode", "basic", "-p", pore, "--model_path", model,
                  "--batch-reads", "4"])
        body = zstd_io.decompress(open(str(out) + ".zst", "rb").read()).decode().splitlines()
        rows[container] = [",".join(r.split(",")[2:]) for r in body[1:]]
        errs = open(str(d / "out" / "res.errors")).read().splitlines()
        assert len(errs) == 1 and "Signal too short compared to sequence" in errs[0]
        seg.close_raw_cache()
    assert len(rows["pod5"]) > 1000 and rows["pod5"] == rows["npz"]


def test_resquiggle_cli_corrupt_pod5_chunk_fails_that_read_only(models, tmp_path):
    """A POD5 chunk that does not decode (here: its zstd frame header overwritten in the file) fails ITS read with the
    line the reference's worker writes for a read whose signal cannot be read (segment.py:178-187) -- the other reads of
    the batch come out as they do from the intact file. Basecalls from a real (unaligned) BAM."""
    from dynamont_amd.pod5_native import vbz_compress
    pore = "rna004"
    model = model_for(models, pore)
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(73, 9, pore, mean, sd, (60, 400))
    outs = {}
    for name in ("intact", "corrupt"):
        d = tmp_path / name
        raw, bam, _ = synth.write_dataset(str(d / "in"), "ds", reads, pore, seed=9, container="pod5", basecalls="bam")
        assert bam.endswith(".bam")
        if name == "corrupt":
            # read 4's only chunk, as write_dataset compressed it
            sm, sdv, scale, offset = 90.0, 15.0, 0.1755, -240.0
            rng = np.random.default_rng(9)
            blob = None
            for i, r in enumerate(reads):
                adc = np.rint((r.signal * sdv + sm) / scale - offset).astype(np.int16)
                pre = rng.integers(300, 900, size=37).astype(np.int16)
                rng.uniform(8.0, 20.0)
                if i == 4:
                    blob = vbz_compress(np.concatenate([pre, adc]))
            data = bytearray(open(raw, "rb").read())
            at = bytes(data).find(blob)
            assert at > 0 and bytes(data).find(blob, at + 1) < 0
            data[at:at + 4] = b"\xde\xad\xbe\xef"   # no longer a zstd frame
            open(raw, "wb").write(bytes(data))
        out = d / "out" / "res.csv"
        seg.main(["-r", str(d / "in"), "-b", bam, "-o", str(out), "--mode", "basic", "-p", pore, "--model_path", model,
                  "--batch-reads", "4"])
        body = zstd_io.decompress(open(str(out) + ".zst", "rb").read()).decode().splitlines()
        errs = open(str(d / "out" / "res.errors")).read().splitlines() if os.path.exists(str(d / "out" / "res.errors")) else []
        outs[name] = (body, errs)
    body_ok, errs_ok = outs["intact"]
    body_bad, errs_bad = outs["corrupt"]
    assert errs_ok == []
    assert len(errs_bad) == 1 and errs_bad[0].startswith("error: worker, Signal could not be decoded\tN: ")
    rid = errs_bad[0].split("\tRid: ")[1].split("\t")[0]
    assert [r for r in body_ok if not r.startswith(rid + ",")] == body_bad and len(body_bad) < len(body_ok)


def test_resquiggle_cli_column_front_end_equals_the_read_by_read_one(models, tmp_path, monkeypatch):
    """BAM basecalls + .pod5 files take the native job reader and prepare_job_columns (no Python object per read);
    DYN_PY_BAM=1 takes the Python BAM parser and prepare_job_raw read by read. Same out.csv.zst rows and the same .errors
    lines -- over two raw files whose reads interleave, a read no file holds, a read whose id is no UUID, a read normalised
    with shift > 400 (ADC counts, no calibration: a second submission of the batch) and a slice cut too short."""
    import uuid
    from dynamont_amd import bam_io
    from dynamont_amd.pod5_io import iter_basecalls
    pore = "rna004"
    model = model_for(models, pore)
    _, mean, sd = synth.read_model_file(model)
    d = tmp_path / "in"
    _, bam_a, _ = synth.write_dataset(str(d), "a", synth.make_reads(81, 14, pore, mean, sd, (60, 500)), pore, seed=3, container="pod5",
                                      basecalls="bam", pod5_chunk_samples=800)
    _, bam_b, _ = synth.write_dataset(str(d), "b", synth.make_reads(82, 13, pore, mean, sd, (60, 500)), pore, seed=4, container="pod5",
                                      basecalls="bam", pod5_chunk_samples=600)
    monkeypatch.setenv("DYN_PY_BAM", "1")
    ra, rb = list(iter_basecalls(bam_a)), list(iter_basecalls(bam_b))
    monkeypatch.delenv("DYN_PY_BAM")
    tup = lambda r, **kw: (kw.get("name", r.query_name), r.query_sequence, {**r._tags, **kw.get("tags", {})})  # noqa: E731
    recs = [tup(r) for pair in zip(ra, rb + [ra[0]]) for r in pair][:-1]
    recs.insert(4, tup(ra[0], name=str(uuid.uuid4())))
    recs.insert(11, tup(rb[1], name="not-a-uuid"))
    recs.insert(17, tup(ra[2], tags={"sm": 500.0, "sd": 90.0}))
    recs.insert(20, tup(rb[3], tags={"ns": rb[3].get_tag("ts") + 30}))
    bam = str(tmp_path / "mixed.bam")
    bam_io.write_bam(bam, recs)
    got = {}
    for name in ("columns", "per_read"):
        if name == "per_read":
            monkeypatch.setenv("DYN_PY_BAM", "1")
        out = tmp_path / name / "res.csv"
        seg.main(["-r", str(d), "-b", bam, "-o", str(out), "--mode", "basic", "-p", pore, "--model_path", model, "--batch-reads", "8"])
        got[name] = (zstd_io.decompress(open(str(out) + ".zst", "rb").read()), sorted(open(str(tmp_path / name / "res.errors")).read().splitlines()))
        seg.close_raw_cache()
    def blocks(blob):   # the rows of each read, in file order; the order of the reads depends on where the batches are cut
        out = []          # (a batch sends its reads without calibration first, and the per-read path fills a batch with
        for line in blob.splitlines()[1:]:   # 8 reads that could be read where the column path takes 8 records)
            rid = line.split(b",")[0]
            if not out or out[-1][0] != rid:
                out.append((rid, []))
            out[-1][1].append(line)
        return out
    ba, bb = blocks(got["columns"][0]), blocks(got["per_read"][0])
    assert len(ba) == len(bb) == len(recs) - 3 and sorted(ba) == sorted(bb) and sum(len(r) for _, r in ba) > 2000
    assert [b for b in ba if b[0] != ra[2].query_name.encode()] == [b for b in bb if b[0] != ra[2].query_name.encode()]   # file order otherwise
    assert got["columns"][1] == got["per_read"][1] and len(got["columns"][1]) == 3
    assert sum("error: worker" in l for l in got["columns"][1]) == 2 and sum("Signal too short" in l for l in got["columns"][1]) == 1


def test_resquiggle_cli_page_starved_batches_in_a_paged_session(models, tmp_path, monkeypatch):
    """`dynamont-resquiggle --mem-budget`: batches whose lattices do not fit an arena per resident wave (long DNA reads; here a
    small budget) run in a PAGED session of the resident read queue. The output must be the one-launch-per-batch output
    (DYN_NO_SESSION=1, the same in-place posterior layout) byte for byte."""
    pore = "dna_r10_400bps"
    model = model_for(models, pore)
    _, mean, sd = synth.read_model_file(model)
    reads = synth.make_reads(91, 1300, pore, mean, sd, (100, 700))
    raw, bam, _ = synth.write_dataset(str(tmp_path / "in"), "ds", reads, pore, seed=5, container="pod5", basecalls="bam")
    monkeypatch.setenv("DYN_FORCE_LAYOUT", "inplace")
    outs = {}
    for name, nosess in (("resident", None), ("launches", "1")):
        if nosess:
            monkeypatch.setenv("DYN_NO_SESSION", nosess)
        out = tmp_path / name / "res.csv"
        seg.main(["-r", str(tmp_path / "in"), "-b", bam, "-o", str(out), "--mode", "basic", "-p", pore, "--model_path", model,
                  "--batch-reads", "650", "--mem-budget", "2.0"])
        outs[name] = zstd_io.decompress(open(str(out) + ".zst", "rb").read())
        seg.close_raw_cache()
    assert outs["resident"].count(b"\n") > 100000
    assert outs["resident"] == outs["launches"]
