"""GPU (-m gpu): P1/P2 on the device (dyn_batch_create_raw) is bit-identical to the NumPy
preprocessing of the reference's workers (segment.py:146-153 float64 / train.py:163-170 float32),
whose `hampel` is itself pinned to the reference's outputs in tests/test_harness.py."""
import numpy as np
import pytest

from conftest import model_for
from dynamont_amd import Aligner, synth
from dynamont_amd.segmentation.utils import hampel

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("native_lib")]


def _numpy_prep(raw, shift, scale, W, ns, f32):
    x = np.array(raw, dtype=np.float32 if f32 else np.float64, copy=True)
    x -= shift
    x /= scale
    hampel(x, W, ns)
    return x.astype(np.float64)


@pytest.mark.parametrize("dtype,W,ns,f32", [(np.float32, 3, 3.0, False), (np.int16, 3, 3.0, False),
                                            (np.float32, 7, 5.0, True), (np.float64, 4, 3.0, False)])
def test_device_preprocessing_bit_identical(models, dtype, W, ns, f32):
    rng = np.random.default_rng(7)
    al = Aligner(models["syn5"], "dna_r9", device=0)
    raws, seqs, shifts, scales = [], [], [], []
    for size in (0, 1, W - 1, W, W + 1, W + 2, 57, 1000, 20011):
        pa = rng.normal(90.0, 15.0, size)
        if size > 20:
            pa[rng.integers(0, size, size=max(1, size // 20))] += rng.choice([-70.0, 55.0, 200.0])
            pa[5:9] = pa[5]                                        # flat run: MAD == 0
        raw = np.rint(pa * 5.7).astype(np.int16) if dtype == np.int16 else pa.astype(dtype)
        raws.append(raw)
        seqs.append("ACGTACGTAC")
        shifts.append(float(np.float32(rng.uniform(80, 100))) * (5.7 if dtype == np.int16 else 1.0))
        scales.append(float(np.float32(rng.uniform(10, 20))))
    with al.batch_raw(raws, seqs, shifts, scales, window=W, n_sigmas=ns, f32=f32) as b:
        got = b.signals()
    want = np.concatenate([_numpy_prep(r, sh, sc, W, ns, f32) for r, sh, sc in zip(raws, shifts, scales)])
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert np.any(np.concatenate([_numpy_prep(r, sh, sc, W, 1e9, f32) for r, sh, sc in zip(raws, shifts, scales)]) != want)


def test_raw_batch_alignment_equals_host_preprocessed(models):
    """align on a raw batch == align on the host-preprocessed signals (same kernels, same inputs)."""
    pore = "rna004"
    _, mean, sd = synth.read_model_file(model_for(models, pore))
    reads = synth.make_reads(81, 6, pore, mean, sd, (100, 400))
    al = Aligner(model_for(models, pore), pore, device=0)
    raws = [(r.signal * 15.0 + 90.0).astype(np.float32) for r in reads]
    shifts, scales = [90.0] * len(reads), [15.0] * len(reads)
    seqs = [r.sequence for r in reads]
    host = al.align_batch([_numpy_prep(x, 90.0, 15.0, 3, 3.0, False) for x in raws], seqs, True)
    with al.batch_raw(raws, seqs, shifts, scales) as b:
        b.align(True)
        dev = b.fetch()
    for i in range(len(reads)):
        a, c = host.read(i), dev.read(i)
        assert a["Z"] == c["Z"] and np.array_equal(a["signal_positions"], c["signal_positions"])
        assert np.array_equal(a["probabilities"], c["probabilities"])


def _pack(raws, seqs):
    off = np.zeros(len(raws) + 1, dtype=np.uint64)
    so = np.zeros(len(raws) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in raws], out=off[1:])
    np.cumsum([len(s) for s in seqs], out=so[1:])
    return np.concatenate(raws), off, "".join(seqs).encode(), so


@pytest.mark.parametrize("calibrated", [True, False])
def test_raw_async_equals_host_preprocessed(models, calibrated):
    """dyn_batch_align_raw_async (what dynamont-resquiggle drives): int16 ADC counts, with the pod5 calibration
    (picoampere formed on the device, raw_dtype 3) or without it (raw_dtype 1), three batches in flight -- every
    result bit-identical to the synchronous call on signals preprocessed with NumPy the way segment.py:141-153 does."""
    pore = "rna004"
    _, mean, sd = synth.read_model_file(model_for(models, pore))
    al = Aligner(model_for(models, pore), pore, device=0)
    rng = np.random.default_rng(5)
    tickets, wants = [], []
    for j in range(4):
        reads = synth.make_reads(500 + j, 9, pore, mean, sd, (60, 300))
        seqs = [r.sequence for r in reads]
        cal_off = rng.uniform(-260, -200, len(reads)).astype(np.float32)
        cal_sc = rng.uniform(0.15, 0.2, len(reads)).astype(np.float32)
        adcs = [np.rint((r.signal * 15.0 + 90.0) / float(sc) - float(of)).astype(np.int16) for r, of, sc in zip(reads, cal_off, cal_sc)]
        if calibrated:
            shifts, scales = rng.uniform(85, 95, len(reads)), rng.uniform(13, 17, len(reads))
            host = [_numpy_prep((a.astype(np.float32) + of) * sc, sh, sc2, 3, 3.0, False)
                    for a, of, sc, sh, sc2 in zip(adcs, cal_off, cal_sc, shifts, scales)]
        else:  # shift > 400: the reference takes the ADC counts themselves (segment.py:147)
            shifts, scales = rng.uniform(700, 800, len(reads)), rng.uniform(80, 100, len(reads))
            host = [_numpy_prep(a, sh, sc2, 3, 3.0, False) for a, sh, sc2 in zip(adcs, shifts, scales)]
        wants.append(al.align_batch(host, seqs, True))
        raw, off, sq, so = _pack(adcs, seqs)
        # odd batches: the slices as a list (DYN_RAW_SCATTERED: the library gathers them), even ones concatenated
        tickets.append(al.align_raw_async(adcs if j % 2 else raw, off, shifts, scales, sq, so,
                                          calibration=(cal_off, cal_sc) if calibrated else None))
    for t, want in zip(tickets, wants):
        got = t.wait()
        assert np.array_equal(got.status, want.status) and (got.status == 0).all()
        for i in range(got.n):
            a, c = want.read(i), got.read(i)
            assert a["Z"] == c["Z"] and np.array_equal(a["signal_positions"], c["signal_positions"])
            assert np.array_equal(a["probabilities"], c["probabilities"])
        t.close()


def test_train_raw_async_equals_host_preprocessed(models):
    """dyn_batch_train_raw_async: float32 arithmetic + Hampel(7, 5 sigma) as in train.py:163-170."""
    pore = "rna002"
    _, mean, sd = synth.read_model_file(model_for(models, pore))
    al = Aligner(model_for(models, pore), pore, device=0)
    reads = synth.make_reads(77, 5, pore, mean, sd, (60, 200))
    seqs = [r.sequence for r in reads]
    raws = [(r.signal * 15.0 + 90.0).astype(np.float32) for r in reads]
    host = al.train_batch([_numpy_prep(x, 90.0, 15.0, 7, 5.0, True) for x in raws], seqs)
    raw, off, sq, so = _pack(raws, seqs)
    t = al.train_raw_async(raw, off, [90.0] * len(reads), [15.0] * len(reads), sq, so)
    got = t.wait()
    assert np.array_equal(got.Z, host.Z) and np.array_equal(got.transitions, host.transitions)
    for i in range(len(reads)):
        for x, y in zip(got.sparse(i), host.sparse(i)):
            assert np.array_equal(x, y)
    t.close()

