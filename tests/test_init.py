"""Boundary: a freshly constructed CVM_* starts from the reference's construction-time distributions (torch default
Conv2d / ConvTranspose2d / Linear / BatchNorm initialisation; /root/reference/models.py:55-148, efficientnet_pytorch/
model.py:376-410) and accepts a local lukemelas EfficientNet-B0 checkpoint for both encoders (utils.py:729-761).
Fixture: tests/golden/init_stats.npz = per-tensor statistics of fresh REFERENCE instances (tools/make_golden_init.py)."""
import math

import numpy as np
import pytest
import torch

import golden_util as G


@pytest.mark.parametrize("kind", ["vigor", "kitti"])
def test_fresh_instance_matches_reference_init_statistics(kind):
    from ccvpe_amd import models
    ref = G.load("init_stats")
    names = [str(n) for n in ref[kind + ":names"]]
    net = models.CVM_VIGOR("cpu", True) if kind == "vigor" else models.CVM_KITTI("cpu")
    sd = net.state_dict()
    assert list(sd.keys()) == names
    checked = 0
    for i, n in enumerate(names):
        t = sd[n].double()
        numel = int(ref[kind + ":numel"][i])
        assert t.numel() == numel, n
        r_mean, r_std, r_max = float(ref[kind + ":mean"][i]), float(ref[kind + ":std"][i]), float(ref[kind + ":absmax"][i])
        if "running_" in n or "num_batches" in n or "._bn" in n:
            # BatchNorm: weight 1, bias 0, running_mean 0, running_var 1, counter 0 — exactly
            assert float(t.mean()) == r_mean and float(t.abs().max()) == r_max, n
            continue
        # U(-b, b) with b = 1 / sqrt(fan_in), fan_in = weight.size(1) * receptive field (torch's rule, also for the
        # ConvTranspose2d weights [Cin, Cout, 2, 2]): nothing may exceed it, and the reference's own sample maximum of a
        # big tensor pins b from below
        w = sd[n if n.endswith(".weight") else n[:-len("bias")] + "weight"]
        bound = 1.0 / math.sqrt(w.shape[1] * int(np.prod(w.shape[2:])))
        assert float(t.abs().max()) <= bound * (1.0 + 1e-6), (n, float(t.abs().max()), bound)
        assert r_max <= bound * (1.0 + 1e-6), (n, r_max, bound)                  # the rule itself, against the fixture
        if numel >= 4096:
            assert r_max >= 0.99 * bound and float(t.abs().max()) >= 0.99 * bound, n
        if numel >= 256:
            tol = max(0.05, 4.0 / math.sqrt(numel))
            assert abs(float(t.std()) - r_std) <= tol * r_std, (n, float(t.std()), r_std)
            assert abs(float(t.mean())) <= 4.0 * r_std / math.sqrt(numel) + abs(r_mean) + 1e-9, n
            checked += 1
    assert checked >= 180, checked
    # two instances differ (fresh draws, like torch), and the decoder is not the synthetic test init
    other = (models.CVM_VIGOR("cpu", True) if kind == "vigor" else models.CVM_KITTI("cpu")).state_dict()
    assert not torch.equal(other["conv6.0.weight"], sd["conv6.0.weight"])


def test_local_efficientnet_checkpoint_loads_into_both_encoders(tmp_path):
    from ccvpe_amd import init, models
    base = models.CVM_VIGOR("cpu", True)
    own = base.state_dict()
    g = torch.Generator().manual_seed(3)
    ckpt = {k[len("grd_efficientnet."):]: (torch.randn(v.shape, generator=g) if v.is_floating_point() else torch.tensor(7))
            for k, v in own.items() if k.startswith("grd_efficientnet.")}
    assert "_fc.weight" in ckpt and "_blocks.15._project_conv.weight" in ckpt and len(ckpt) == 360      # lukemelas key set
    path = str(tmp_path / "efficientnet-b0.pth")
    torch.save(ckpt, path)
    net = models.CVM_VIGOR_ori_prior("cpu", 0, True, efficientnet_weights=path)
    sd = net.state_dict()
    for k, v in ckpt.items():
        assert torch.equal(sd["grd_efficientnet." + k], v) and torch.equal(sd["sat_efficientnet." + k], v), k
    assert not torch.equal(sd["conv6.0.weight"], own["conv6.0.weight"])          # decoder: its own fresh default init
    # the reference's checks (utils.py:752-758): missing / unexpected keys are errors
    bad = dict(ckpt)
    bad.pop("_bn0.weight")
    with pytest.raises(KeyError):
        init.load_efficientnet_b0(base, bad)
    bad = dict(ckpt)
    bad["_extra.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        init.load_efficientnet_b0(base, bad)
    nofc = {k: v for k, v in ckpt.items() if not k.startswith("_fc.")}
    init.load_efficientnet_b0(base, nofc, load_fc=False)                          # load_fc=False tolerates the missing head
    assert torch.equal(base.state_dict()["sat_efficientnet._conv_head.weight"], ckpt["_conv_head.weight"])
