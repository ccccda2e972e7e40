import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-separation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def fixture_samples(fx, prefix, n, keys):
    """Rebuild the list-of-dict samples stored by tests/golden/make_fixtures.py."""
    out = []
    for i in range(n):
        d = {}
        for k in keys:
            name = "%ssample%d_%s" % (prefix, i, k)
            if name in fx:
                v = fx[name]
                d[k] = str(v) if v.dtype.kind in "US" else v
        out.append(d)
    return out
