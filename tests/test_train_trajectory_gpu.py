"""A whole optimisation TRAJECTORY, not one gradient: STEPS (3; 10 with CCVPE_SLOW_TESTS=1) training steps of CVM_VIGOR at B = 2 on the MI355X (device-side
ground truth, train-mode forward with running-statistic updates, the reference's loss mix train_VIGOR.py:137-146, the HIP
backward through the gradient arena, the one-launch Adam) against the CPU oracle driven by autograd + torch.optim.Adam on
the same weights, pairs, drop_connect draws and targets.  Per-step gradients agree to ~1 % per tensor (golden_util
compare_grads); here the bar is on what that does over time: the loss sequence and the parameter displacement."""
import numpy as np
import pytest
import torch

import golden_util as G
from ccvpe_amd import synth
from oracle import ccvpe_oracle as O

pytestmark = pytest.mark.gpu

import os
# 10 steps in round 2 (150 s of CPU oracle steps); 3 keep the GPU suite inside its 900 s time box, CCVPE_SLOW_TESTS=1 runs the 10
STEPS = 10 if os.environ.get("CCVPE_SLOW_TESTS") == "1" else 3
LR = 1e-4                      # train_VIGOR.py:104


def _loss_mix(mod, out, gt, gt_flat, gt_ori, labels):
    nce = 0.0
    for lvl in range(6):
        nce = nce + mod[0](torch.flatten(out[3 + lvl], start_dim=1), torch.flatten(labels[lvl], start_dim=1))
    return mod[1](out[0], gt_flat) + 1e4 * nce / 6 + 1e1 * mod[2](out[2], gt_ori, gt)


def test_trajectory_vs_oracle_and_torch_adam(synth_sd):
    from ccvpe_amd import harness, losses, models, optim, targets
    c = G.TRAIN_CASE
    batch = c["batch"]
    sd = synth_sd(c["kind"], c["wseed"])
    grd, sat = synth.synthetic_pair(batch, c["grd"], c["pseed"])
    masks, scales, _ = G.train_drop_masks(batch)
    centers = [[37.0, -120.0], [-60.0, 25.0]]
    angles = [200.25, 33.0]

    # ---- MI355X -------------------------------------------------------------------------------------------
    net = models.CVM_VIGOR("cuda", c["circular"])
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").train()
    opt = optim.Adam(net.parameters(), lr=LR, betas=(0.9, 0.999))
    harness.GradientAllReducer(net.parameters()).attach(net, optimizer=opt)        # single rank: the arena path, no collective
    g, s = grd.cuda(), sat.cuda()
    ctr, ang = torch.tensor(centers).cuda(), torch.tensor(angles).cuda()
    hip_losses = []
    for _ in range(STEPS):
        opt.zero_grad(set_to_none=True)
        gt, gt_flat, gt_ori, labels = targets.train_targets(ctr, ang, 20)
        out = net(g, s, drop_masks=masks)
        loss = _loss_mix((losses.infoNCELoss, losses.cross_entropy_loss, losses.orientation_loss), out, gt, gt_flat, gt_ori, labels)
        loss.backward()
        opt.step()
        hip_losses.append(float(loss.detach()))
    torch.cuda.synchronize()

    # ---- CPU oracle + torch.optim.Adam --------------------------------------------------------------------------
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
              for k, v in sd.items()}
    learn = [v for k, v in params.items() if v.requires_grad]
    ref_opt = torch.optim.Adam(learn, lr=LR, betas=(0.9, 0.999))
    gt, gt_flat, gt_ori, labels = O.train_targets(centers, angles, 20)
    ref_losses = []
    stats = {}
    for _ in range(STEPS):
        ref_opt.zero_grad(set_to_none=True)
        out = O.forward(params, grd, sat, c["kind"], c["circular"], None, train_stats=stats, drop_scales=scales)
        loss = _loss_mix((O.infonce_loss, O.cross_entropy_loss, O.orientation_loss), out, gt, gt_flat, gt_ori, labels)
        loss.backward()
        ref_opt.step()
        for k, v in stats.items():                     # the running statistics the train-mode forward updated
            params[k] = v
        ref_losses.append(float(loss.detach()))

    hip, ref = np.array(hip_losses), np.array(ref_losses)
    rel = np.abs(hip - ref) / np.abs(ref)
    print("loss trajectory  hip: %s\n                 ref: %s\n  rel diff per step: %s" % (np.round(hip, 4), np.round(ref, 4), np.round(rel, 6)))
    assert rel[0] < 1e-4, "first loss (identical weights) differs: %g" % rel[0]
    assert rel.max() < 5e-3, "loss trajectories diverge: %s" % rel
    assert ref[-1] < ref[0] and hip[-1] < hip[0], "the Adam steps did not reduce the loss"
    # displacement of the parameters after the steps (Adam's early steps are ~ lr * sign(g): entries whose gradient is
    # round-off-level noise move in arbitrary directions on both sides, so the bar is on the big, well-conditioned tensors)
    live = dict(net.named_parameters())
    checked = 0
    for name in ("conv6.0.weight", "conv3.2.weight", "deconv5.weight", "sat_feature_to_descriptors.1.weight", "conv1.0.weight",
                 "conv4_ori.0.weight", "sat_efficientnet._blocks.15._project_conv.weight", "grd_efficientnet._conv_head.weight"):
        d_ref = (params[name].detach() - sd[name]).double()
        d_hip = (live[name].detach().cpu() - sd[name]).double()
        cos = float((d_ref * d_hip).sum() / (d_ref.norm() * d_hip.norm() + 1e-300))
        ratio = float(d_hip.norm() / (d_ref.norm() + 1e-300))
        print("%-52s displacement cosine %.4f, norm ratio %.4f" % (name, cos, ratio))
        assert cos > 0.9 and 0.9 < ratio < 1.1, (name, cos, ratio)
        checked += 1
    assert checked == 8
    # running statistics followed the same trajectory
    rm = "sat_efficientnet._blocks.10._bn2.running_var"
    got = dict(net.named_buffers())[rm].cpu()
    assert torch.allclose(got, params[rm], rtol=2e-3, atol=1e-6)
