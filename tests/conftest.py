import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


_SD_CACHE = {}


@pytest.fixture(scope="session")
def synth_sd():
    """(kind, seed) -> reference-layout state_dict, cached for the session."""
    from ccvpe_amd import synth

    def get(kind, seed):
        key = (kind, seed)
        if key not in _SD_CACHE:
            _SD_CACHE[key] = synth.synthetic_state_dict(kind, seed)
        return _SD_CACHE[key]
    return get
