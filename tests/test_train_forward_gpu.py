"""Train-mode forward on the MI355X: the batch-statistic BatchNorm kernels and the train-mode encoders
(drop_connect draws injected) against the oracle / the reference's own .train() golden."""
import pytest
import torch
import torch.nn.functional as F

import golden_util as G
from ccvpe_amd import synth
from oracle import ccvpe_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from ccvpe_amd import ops as _ops, _lib
    _lib.load()
    return _ops


def close(got, want, tol, what):
    got, want = got.detach().cpu().double(), want.detach().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("b,h,w,c", [(2, 9, 13, 96), (3, 16, 16, 1152), (2, 40, 72, 32), (1, 5, 7, 24)])
def test_bn_stats_and_running_update(ops, b, h, w, c):
    x = synth.normal((b, h, w, c), 700 + c, 1.7) + synth.normal((c,), 701, 2.0)      # non-zero channel means
    rm, rv = synth.normal((c,), 702, 0.3), synth.uniform((c,), 703, 0.5, 1.5)
    drm, drv = rm.clone().cuda(), rv.clone().cuda()
    mean, var = ops.bn_stats(x.cuda(), drm, drv, 0.01)
    xr = x.reshape(-1, c).double()
    close(mean, xr.mean(0), 1e-6, "mean")
    close(var, xr.var(0, unbiased=False), 1e-5, "biased var")
    close(drm, 0.99 * rm.double() + 0.01 * xr.mean(0), 1e-6, "running mean")
    close(drv, 0.99 * rv.double() + 0.01 * xr.var(0, unbiased=True), 1e-5, "running var (unbiased)")


def test_bn_act_matches_torch_batch_norm(ops):
    b, h, w, c = 3, 10, 12, 144
    x = synth.normal((b, c, h, w), 710, 1.5)
    ga, be = synth.uniform((c,), 711, 0.5, 1.5), synth.normal((c,), 712, 0.2)
    res = synth.normal((b, c, h, w), 713)
    dcs = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9])
    want = F.batch_norm(x, None, None, ga, be, True, 0.0, 1e-3)
    want_sw = O.swish(want)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    mean, var = ops.bn_stats(xd)
    y, part = ops.bn_act(xd, mean, var, ga.cuda(), be.cuda(), 1e-3, ops.ACT_SWISH, want_se=True)
    close(y.permute(0, 3, 1, 2), want_sw, 1e-5, "bn+swish")
    close(part.sum(1), want_sw.sum(dim=(2, 3)), 1e-5, "SE partials")
    y2 = ops.bn_act(xd, mean, var, ga.cuda(), be.cuda(), 1e-3, ops.ACT_NONE,
                    residual=res.permute(0, 2, 3, 1).contiguous().cuda(), dc_scale=dcs.cuda())
    close(y2.permute(0, 3, 1, 2), want * dcs.view(-1, 1, 1, 1) + res, 1e-5, "bn * drop_connect + residual")


@pytest.mark.parametrize("case", ["vigor", "kitti"])
def test_train_mode_forward_vs_reference_golden(synth_sd, case):
    """CVM_VIGOR and CVM_KITTI (BASELINE config C3's model) in .train() against the reference classes' own outputs."""
    from ccvpe_amd import models
    c = G.TRAIN_CASES[case]
    want = G.load("fwd_%s_trainmode" % case)
    sd = synth_sd(c["kind"], c["wseed"])
    net = models.CVM_KITTI("cuda") if c["kind"] == "kitti" else models.CVM_VIGOR("cuda", c["circular"])
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").train()
    grd, sat = synth.synthetic_pair(c["batch"], c["grd"], c["pseed"])
    masks, scales, _ = G.train_drop_masks(c["batch"])
    out = net(grd.cuda(), sat.cuda(), drop_masks=masks)
    torch.cuda.synchronize()
    got = G.summarize_forward([t.cpu() for t in out])
    scale = abs(want["logits_s4"]).max()
    assert (got["top4_idx"][:, 0] == want["top4_idx"][:, 0]).all()
    G.assert_close(got["logits_s4"], want["logits_s4"], 0, 1e-3 * scale, "train-mode logits")
    for i in range(1, 7):
        s = abs(want["score%d" % i]).max()
        G.assert_close(got["score%d" % i], want["score%d" % i], 0, 1e-3 * s, "train-mode score%d" % i)
    after = net.state_dict()
    for k in G.RUNNING_STAT_SAMPLES:
        G.assert_close(after[k + ".running_mean"].cpu().numpy(), want["rm:" + k], 1e-4, 1e-5, "running_mean " + k)
        G.assert_close(after[k + ".running_var"].cpu().numpy(), want["rv:" + k], 1e-4, 1e-5, "running_var " + k)
        assert int(after[k + ".num_batches_tracked"]) == 1
    # eval after a train step must use the UPDATED running statistics (folded pack invalidated)
    net.eval()
    out_eval = net(grd.cuda(), sat.cuda())
    sd2 = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref_eval = O.forward(sd2, grd, sat, c["kind"], c["circular"], None)
    assert ((out_eval[0].cpu() - ref_eval[0]).abs().max() / ref_eval[0].abs().max()).item() < 1e-3


def test_train_mode_random_drop_connect_runs(synth_sd):
    """Without injected masks the draws follow utils.py:145-150 (stochastic): shapes/finite only."""
    from ccvpe_amd import models
    net = models.CVM_KITTI("cuda")
    net.load_state_dict(synth_sd("kitti", 1), strict=True)
    net = net.to("cuda:0").train()
    grd, sat = synth.synthetic_pair(2, "kitti", 5)
    out = net(grd.cuda(), sat.cuda())
    assert tuple(out[0].shape) == (2, 262144) and all(torch.isfinite(t).all() for t in out)
