"""Per-launch-shape timing of the dense GEMM-like kernels (igemm / conv3x3 / conv_wgrad) of ONE training step
(HIP events around every library call; run on the GPU box):  python tools/train_layer_times.py [kitti|vigor] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ccvpe_amd import models, ops, synth       # noqa: E402
import golden_util as G                         # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "kitti"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
net = models.CVM_KITTI("cuda") if kind == "kitti" else models.CVM_VIGOR("cuda", True)
net.load_state_dict(synth.synthetic_state_dict(kind, 0), strict=True)
net = net.to("cuda:0").train()
grd, sat = synth.synthetic_pair(batch, kind, 1)
grd, sat = grd.cuda(), sat.cuda()
for it in range(2):
    rec = ops.LaunchRecorder()
    if it == 1:
        ops.set_recorder(rec)
    out = net(grd, sat)
    loss = out[0].float().mean() + out[2].mean() + sum(o.mean() for o in out[3:])
    loss.backward()
    net.zero_grad(set_to_none=True)
torch.cuda.synchronize()
ops.set_recorder(None)
agg = {}
for name, tag, flops, nbytes, e0, e1 in rec.items:
    d = agg.setdefault((name, tag), [0, 0.0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += flops; d[3] += nbytes
tot = sum(d[1] for d in agg.values())
print("total recorded ms %.1f" % tot)
print("%-28s %-34s %3s %8s %8s %8s" % ("kernel", "shape", "n", "ms", "TFLOP/s", "GB/s"))
for (name, tag), d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-28s %-34s %3d %8.3f %8.1f %8.0f" % (name, tag, d[0], d[1], d[2] / d[1] / 1e9, d[3] / d[1] / 1e6))
