"""GraphedTrainStep (ccvpe_amd/graph.py): a captured forward + losses + backward replays to the same loss and gradients as
the eager step on the same weights, inputs and random-generator state, and a few graphed steps with the eager Adam train."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(synth_sd, batch):
    from ccvpe_amd import models, synth, targets
    net = models.CVM_VIGOR("cuda", True)
    net.load_state_dict(synth_sd("vigor", 0), strict=True)
    net = net.to("cuda:0").train()
    grd, sat = synth.synthetic_pair(batch, "vigor", 321)
    grd, sat = grd.cuda(), sat.cuda()
    u = synth.uniform((batch, 3), 17)
    center = ((u[:, :2] - 0.5) * 300.0).cuda()
    angle = (u[:, 2] * 359.0).cuda()

    def loss_fn():
        from ccvpe_amd import losses
        gt, gt_flat, gt_ori, labels = targets.train_targets(center, angle, 20)
        out = net(grd, sat)
        nce = 0.0
        for lvl in range(6):
            nce = nce + losses.infoNCELoss(torch.flatten(out[3 + lvl], start_dim=1), torch.flatten(labels[lvl], start_dim=1))
        return losses.cross_entropy_loss(out[0], gt_flat) + 1e4 * nce / 6 + 1e1 * losses.orientation_loss(out[2], gt_ori, gt)
    return net, loss_fn


def test_graphed_train_step_matches_eager_and_trains(synth_sd):
    from ccvpe_amd import graph, optim
    batch = 2
    net, loss_fn = _setup(synth_sd, batch)
    torch.manual_seed(5)
    le = loss_fn()
    le.backward()
    le = le.detach()
    want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # the same model (its weights are untouched: no optimizer step yet; the running statistics the warm-up steps move do not
    # enter a train-mode forward)
    step = graph.GraphedTrainStep(loss_fn, net)
    assert getattr(net, "_pack_plan", None) is not None               # the re-pack inside the graph is the gather launch
    torch.manual_seed(5)
    lg = step()
    torch.cuda.synchronize()
    assert torch.isfinite(lg).all()
    assert abs(float(lg) - float(le)) <= 1e-5 * abs(float(le)), (float(lg), float(le))
    got = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert set(got) == set(want)
    worst = 0.0
    for n in want:
        scale = float(want[n].abs().max()) + 1e-30
        worst = max(worst, float((got[n] - want[n]).abs().max()) / scale)
    assert worst <= 1e-4, worst
    # a few steps with the eager optimizer: the loss moves and stays finite, the re-pack inside the graph follows the weights
    opt = optim.Adam(net.parameters(), lr=1e-4)
    losses_seen = []
    static = [p.grad for p in net.parameters()]
    for it in range(3):
        if it == 1:
            opt.zero_grad(set_to_none=True)       # detaches p.grad; the next replay re-attaches the tensors it writes
        loss = step()
        assert all(p.grad is g for p, g in zip(net.parameters(), static))
        opt.step()
        losses_seen.append(float(loss))
    step.zero_grad()
    assert all(g is None or float(g.abs().max()) == 0.0 for g in static)
    assert all(x == x and abs(x) < 1e9 for x in losses_seen) and len(set(losses_seen)) > 1
