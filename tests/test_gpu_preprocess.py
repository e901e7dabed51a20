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
