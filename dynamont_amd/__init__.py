"""dynamont_amd -- MI355X-native NT ("basic" mode) resquiggling core.

Drop-in for the reference's ``from dynamont import Aligner, PoreType``
(src/dynamont/__init__.py:8-12): same surface, backed by hand-written gfx950 kernels behind
the C ABI of include/dynamont_mi.h. No CPU or PyTorch compute path exists in this package.
"""
__version__ = "0.1.0"

from ._dynamont import Aligner, MultiAligner, PoreType, pore_type, release_cached_memory  # noqa: E402,F401

__all__ = ["Aligner", "MultiAligner", "PoreType", "pore_type", "release_cached_memory", "__version__"]
