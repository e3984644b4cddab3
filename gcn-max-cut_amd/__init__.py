"""gcn-max-cut_amd - MI355X-native GCN max-cut training/inference path.

Drop-in for the hot path of MJavaadAkhtar/GCN-max-cut (``python/Training/TrainingNeural.py``
fed by ``python/DataGenerator/graphExtender.py``): same names, arguments and error
behaviour, with the arithmetic in hand-written HIP kernels for gfx950 behind the C ABI of
``include/gcnmaxcut.h``.  Import as ``gcn_max_cut_amd`` (the directory name carries a
hyphen; ``gcn_max_cut_amd/`` at the repo root is the importable alias), or put this
directory on ``sys.path`` to get the reference's own import roots
(``Training.TrainingNeural``, ``DataGenerator.graphExtender``, ``commons``, ``python.*``).
"""
from . import hip  # noqa: F401
from .graph import BatchArrays, DGLError, GraphBatch, GraphHandle, from_networkx  # noqa: F401
from .engine import FusedEngine, shard_by_weight, shard_for_rank  # noqa: F401

__version__ = "0.1.0"


def install_compat() -> str:
    """Put the reference's import roots (``Training.TrainingNeural``, ``DataGenerator.graphExtender``,
    ``commons``, ``python.*``) on ``sys.path`` so unmodified notebook imports resolve here."""
    import os
    import sys
    path = os.path.join(__path__[0], "compat")
    if path not in sys.path:
        sys.path.insert(0, path)
    return path
