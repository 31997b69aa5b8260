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


_ORACLE_CACHE = {}


@pytest.fixture(scope="session")
def oracle_forward(synth_sd):
    """(case dict, batch, input seed) -> (grd, sat, oracle outputs) on the CPU, cached for the session: the fp32 and the bf16
    forward tests compare against the SAME oracle run on the same seeded inputs (30-40 s of CPU each on a GPU box)."""
    import torch
    from ccvpe_amd import synth
    from oracle import ccvpe_oracle as O

    def get(case, batch, seed):
        key = (case["kind"], case["ori_noise"], case["circular"], case["wseed"], case["grd"], batch, seed)
        if key not in _ORACLE_CACHE:
            grd, sat = synth.synthetic_pair(batch, case["grd"], seed)
            with torch.no_grad():
                ref = O.forward(synth_sd(case["kind"], case["wseed"]), grd, sat, case["kind"], case["circular"], case["ori_noise"])
            _ORACLE_CACHE[key] = (grd, sat, [t.detach() for t in ref])
        return _ORACLE_CACHE[key]
    return get
