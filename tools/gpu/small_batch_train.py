"""Small-batch training step, eager vs GraphedTrainStep (forward + losses + backward in one hipGraph, eager Adam):
    python tools/gpu/small_batch_train.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ccvpe_amd import graph, losses, models, optim, synth, targets      # noqa: E402


def build(batch):
    net = models.CVM_VIGOR("cuda", True)
    net.load_state_dict(synth.synthetic_state_dict("vigor", 0), strict=True)
    net = net.to("cuda:0").train()
    grd, sat = synth.synthetic_pair(batch, "vigor", 321)
    grd, sat = grd.cuda(), sat.cuda()
    u = synth.uniform((batch, 3), 17)
    center, angle = ((u[:, :2] - 0.5) * 300.0).cuda(), (u[:, 2] * 359.0).cuda()

    def loss_fn():
        gt, gt_flat, gt_ori, labels = targets.train_targets(center, angle, 20)
        out = net(grd, sat)
        nce = 0.0
        for lvl in range(6):
            nce = nce + losses.infoNCELoss(torch.flatten(out[3 + lvl], start_dim=1), torch.flatten(labels[lvl], start_dim=1))
        return losses.cross_entropy_loss(out[0], gt_flat) + 1e4 * nce / 6 + 1e1 * losses.orientation_loss(out[2], gt_ori, gt)
    return net, loss_fn


def timed(fn, steps=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for batch in [int(a) for a in sys.argv[1:]] or [8]:
    net, loss_fn = build(batch)
    opt = optim.Adam(net.parameters(), lr=1e-4)

    def eager():
        opt.zero_grad(set_to_none=True)
        loss_fn().backward()
        opt.step()
    ms_e = timed(eager)
    del net, opt
    torch.cuda.empty_cache()
    net, loss_fn = build(batch)
    opt = optim.Adam(net.parameters(), lr=1e-4)
    step = graph.GraphedTrainStep(loss_fn, net)

    def graphed():
        step()
        opt.step()
    ms_g = timed(graphed)
    print("B = %d: eager %.2f ms/step (%.0f pairs/s), graphed %.2f ms/step (%.0f pairs/s), loss %.4g" %
          (batch, ms_e, batch / ms_e * 1e3, ms_g, batch / ms_g * 1e3, float(step.loss)))
    del net, opt, step
    torch.cuda.empty_cache()
