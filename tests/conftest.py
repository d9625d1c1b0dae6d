import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def kats():
    """The small known-answer cases produced by the real reference (tests/golden/make_golden.py)."""
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "kat_small.npz"))
    out = []
    for i in range(int(z["n"])):
        out.append(dict(tag=str(z["tags"][i]), value=z[f"c{i}_value"], tx=z[f"c{i}_tx"], ty=z[f"c{i}_ty"],
                        path=z[f"c{i}_path"], q=z[f"c{i}_q"], neg=float(z[f"c{i}_neg"])))
    return out


@pytest.fixture(scope="session")
def appendix_a():
    import json
    import numpy as np
    with open(os.path.join(GOLDEN, "appendix_a.json")) as f:
        rec = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, "appendix_a_arrays.npz")))
    return rec, arrays


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libaligner_amd.so exists (cross-compiles without a GPU)."""
    from aligner_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()
