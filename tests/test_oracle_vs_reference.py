"""Live check of the oracle against the reference itself — only where /root/reference exists
(the build container).  Skipped on the GPU box."""
import pytest
import torch

from ref_import import reference_available, import_reference
from ccvpe_amd import synth
from oracle import ccvpe_oracle as O

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference not present")


def test_vigor_ori_prior_live(synth_sd):
    ref_models, _ = import_reference()
    sd = synth_sd("vigor", 0)
    net = ref_models.CVM_VIGOR_ori_prior("cpu", 36, True)
    net.load_state_dict(sd, strict=True)
    net.eval()
    grd, sat = synth.synthetic_pair(1, "vigor", 4321)
    with torch.no_grad():
        want = net(grd, sat)
        got = O.forward(sd, grd, sat, "vigor", True, ori_noise=36)
    assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in want]
    assert torch.equal(got[0].argmax(1), want[0].argmax(1))
    assert (got[0] - want[0]).abs().max() < 2e-5
    for a, b in zip(got[3:], want[3:]):
        assert (a - b).abs().max() < 2e-6


def test_state_dict_layout_matches_reference():
    ref_models, _ = import_reference()
    for kind, net in (("vigor", ref_models.CVM_VIGOR("cpu", True)),
                      ("kitti", ref_models.CVM_KITTI("cpu")),
                      ("oxford", ref_models.CVM_OxfordRobotCar("cpu"))):
        ref = net.state_dict()
        spec = synth.state_dict_spec(kind)
        assert [k for k, _, _ in spec] == list(ref.keys())
        for k, shape, _ in spec:
            assert tuple(ref[k].shape) == tuple(shape), k
