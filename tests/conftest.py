import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import _pkgload  # noqa: E402

_pkgload.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    gdir = os.path.join(ROOT, "tests", "golden")
    out = {}
    for f in os.listdir(gdir):
        if f.endswith(".npz"):
            with np.load(os.path.join(gdir, f)) as z:
                for k in z.files:
                    out[k] = z[k]
    return out
