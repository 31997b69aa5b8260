"""Arg-max agreement of the bf16 storage path with fp32 over seeded pairs (run on the GPU box):
    python tools/bf16_argmax_rate.py [n_pairs=64] [model=vigor20|prior0|kitti]
For fp32_tail_levels in 0..3 prints: pairs whose heat-map arg-max pixel equals the fp32 HIP path's (itself arg-max exact
against the reference on the golden cases, tests/test_forward_gpu.py), the worst logit error relative to the logit range,
and how that compares with the fp32 top-1 / top-2 margin.  The first 8 pairs are also checked against the CPU oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccvpe_amd import models, synth        # noqa: E402
from oracle import ccvpe_oracle as O       # noqa: E402  (diagnostic tool, not product code)

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "vigor20"
kind = "kitti" if which == "kitti" else "vigor"
sd = synth.synthetic_state_dict(kind, 0)


def make():
    if which == "kitti":
        return models.CVM_KITTI("cuda")
    if which == "prior0":
        return models.CVM_VIGOR_ori_prior("cuda", 0, True)
    return models.CVM_VIGOR("cuda", True)


net = make()
net.load_state_dict(sd, strict=True)
net = net.to("cuda:0").eval()
chunk = 16
ref_logits, margins = [], []
pairs = []
for c0 in range(0, n_pairs, chunk):
    grd, sat = synth.synthetic_pair(min(chunk, n_pairs - c0), kind, 5000 + c0)
    pairs.append((grd, sat))
    lg = net.set_precision("fp32")(grd.cuda(), sat.cuda())[0]
    ref_logits.append(lg.cpu())
    top2 = lg.topk(2, dim=1)[0]
    margins.append(((top2[:, 0] - top2[:, 1]) / (lg.max(1)[0] - lg.min(1)[0])).cpu())
ref_logits = torch.cat(ref_logits)
margins = torch.cat(margins)
print("fp32 top1-top2 margin / logit range over %d pairs: min %.2e median %.2e" % (n_pairs, float(margins.min()), float(margins.median())))
with torch.no_grad():
    g8, s8 = pairs[0][0][:8], pairs[0][1][:8]
    ora = O.forward(sd, g8, s8, kind, True, 0 if which == "prior0" else None)[0]
print("fp32 HIP vs CPU oracle on 8 pairs: arg-max equal %d/8" % int((ora.argmax(1) == ref_logits[:8].argmax(1)).sum()))
for tail in (0, 1, 2, 3):
    net.set_precision("bf16", fp32_tail_levels=tail)
    got = torch.cat([net(g.cuda(), s.cuda())[0].cpu() for g, s in pairs])
    rng = ref_logits.max(1)[0] - ref_logits.min(1)[0]
    err = ((got - ref_logits).abs().max(1)[0] / rng)
    same = got.argmax(1) == ref_logits.argmax(1)
    # where the arg-max moved: how far below the fp32 maximum is the fp32 logit of the pixel bf16 picked?
    moved = (~same).nonzero().flatten().tolist()
    drop = [float((ref_logits[i].max() - ref_logits[i, got[i].argmax()]) / rng[i]) for i in moved]
    print("bf16 fp32_tail_levels=%d: arg-max equal %d/%d, logit err/range max %.2e median %.2e; moved pairs' fp32 deficit/range: %s"
          % (tail, int(same.sum()), n_pairs, float(err.max()), float(err.median()), ["%.1e" % d for d in drop]))
