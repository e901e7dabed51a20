import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def models(tmp_path_factory):
    """Synthetic model files from the committed seeded generator (the same the goldens used)."""
    from dynamont_amd import synth
    d = tmp_path_factory.mktemp("models")
    m5 = synth.write_model(str(d / "syn5.model"), 5, seed=7, stdev=0.25)
    m9 = synth.write_model(str(d / "syn9.model"), 9, seed=7, stdev=0.15)
    return {"syn5": m5, "syn9": m9, "dir": str(d)}


@pytest.fixture(scope="session")
def native_lib():
    from dynamont_amd import _native
    if _native.needs_build():
        _native.build()
    return _native.lib()


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import pyoracle
    pyoracle.build("oracle")
    return True


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def model_for(models, pore):
    return models["syn5"] if pore in ("rna002", "dna_r9") else models["syn9"]


def assert_matches_golden(res: dict, g, p: str, prob_tol=1e-4, z_rel=1e-9):
    """integer columns bit-exact; posterior within north_star's 1e-4; Z to 1e-9 relative."""
    assert np.array_equal(np.asarray(res["sequence_positions"], dtype=np.uint64), g[p + "seqpos"].astype(np.uint64))
    assert np.array_equal(np.asarray(res["signal_positions"], dtype=np.uint64), g[p + "sigpos"].astype(np.uint64))
    assert all(s == "M" for s in res["states"]) == bool(g[p + "all_M"])
    if len(g[p + "prob"]):
        assert np.abs(np.asarray(res["probabilities"]) - g[p + "prob"]).max() <= prob_tol
    zg = float(g[p + "Z"])
    assert abs(res["Z"] - zg) <= z_rel * max(1.0, abs(zg))
