"""The 3x3 convolution has three stage-loop forms selected by environment switches read when the library is loaded
(ccvpe_amd/csrc/conv_igemm.hip): the default row-of-taps stage with early DMA, one tap per stage (CCVPE_CONV3_TPS=1, also
what the 8-wave and ragged-N variants use) and the W-from-L2 experiment (CCVPE_CONV3_WREG=1).  The non-default forms are
kept for A/B measurements: run the 3x3 parity tests under each switch in a fresh process so that they stay correct."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"CCVPE_CONV3_TPS": "1"}, {"CCVPE_CONV3_WREG": "1"}, {"CCVPE_CONV3_NW8": "0"}])
def test_conv3x3_parity_under_switch(env):
    e = dict(os.environ)
    e.update(env)
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_ops_gpu.py", "tests/test_backward_gpu.py",
                          "-k", "igemm_3x3_two_sources or conv_wgrad_and_dgrad"],
                         cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
