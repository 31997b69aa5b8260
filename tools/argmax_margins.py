"""Exploration for tests/test_bf16_gpu.py: top-1 / top-2 margin (of the logit range) of the fp32 heat-map logits under a few
synthetic-weight variants, and the bf16 path's arg-max agreement on each: python tools/argmax_margins.py [pairs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import models, synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def variant(name):
    sd = synth.synthetic_state_dict("vigor", 0 if name != "seed3" else 3)
    if name == "center1":                       # conv1.0 / conv1.2 as centre-tap filters: logits pointwise in the deconv output
        for k in ("conv1.0.weight", "conv1.2.weight"):
            w = sd[k].clone()
            c = w[:, :, 1, 1].clone() * 3.0
            w.zero_()
            w[:, :, 1, 1] = c
            sd[k] = w
    if name == "hf":                            # strong last skip-free level: amplify deconv1 (per-parity weights -> pixel-level variation)
        sd["deconv1.weight"] = sd["deconv1.weight"] * 4.0
        sd["conv1.2.weight"] = sd["conv1.2.weight"] * torch.tensor([[[-1.0, 1.0, -1.0], [1.0, 2.0, 1.0], [-1.0, 1.0, -1.0]]])
    if name == "peaky":                         # the six ground descriptors and the aerial descriptor x8: cosine scores saturate locally
        for l in range(1, 7):
            sd["grd_feature_to_descriptor%d.0.weight" % l] = sd["grd_feature_to_descriptor%d.0.weight" % l] * 8.0
        sd["conv1.2.weight"] = sd["conv1.2.weight"] * torch.tensor([[[0.0, -1.0, 0.0], [-1.0, 5.0, -1.0], [0.0, -1.0, 0.0]]])
    return sd


for name in ("seed0", "seed3", "center1", "hf", "peaky"):
    net = models.CVM_VIGOR_ori_prior("cuda", 0, True)
    net.load_state_dict(variant(name), strict=True)
    net = net.to("cuda:0").eval()
    margins, same, errs = [], 0, []
    for c0 in range(0, pairs, 16):
        grd, sat = synth.synthetic_pair(16, "vigor", 9000 + c0)
        grd, sat = grd.cuda(), sat.cuda()
        ref = net.set_precision("fp32")(grd, sat)[0].clone()
        got = net.set_precision("bf16")(grd, sat)[0]
        rng = ref.max(1)[0] - ref.min(1)[0]
        top2 = ref.topk(2, dim=1)[0]
        margins += ((top2[:, 0] - top2[:, 1]) / rng).tolist()
        errs += ((got - ref).abs().max(1)[0] / rng).tolist()
        same += int((ref.argmax(1) == got.argmax(1)).sum())
    m = sorted(margins)
    print("%-8s margins/range: min %.2e  q25 %.2e  median %.2e  q75 %.2e  max %.2e | > 1.2e-2: %d of %d | bf16 arg-max equal %d, worst err %.2e"
          % (name, m[0], m[len(m) // 4], m[len(m) // 2], m[3 * len(m) // 4], m[-1], sum(x > 1.2e-2 for x in m), len(m), same, max(errs)), flush=True)
