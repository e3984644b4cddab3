import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """In-tree HIP library + C oracle (compiled on demand; hipcc needs no GPU)."""
    import __graft_entry__ as entry
    import gcn_max_cut_amd as pkg
    if not os.path.exists(pkg.hip.LIB_PATH):
        entry.build()
    from oracle import c_oracle
    c_oracle.build()
    return pkg
