"""Forward / backward split of one training step (HIP events): python tools/train_phase_times.py [kitti|vigor] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import models, synth, train       # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "vigor"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
net = models.CVM_KITTI("cuda") if kind == "kitti" else models.CVM_VIGOR("cuda", True)
net.load_state_dict(synth.synthetic_state_dict(kind, 0), strict=True)
net = net.to("cuda:0").train()
grd, sat = synth.synthetic_pair(batch, kind, 1)
grd, sat = grd.cuda(), sat.cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for it in range(3):
    with torch.no_grad():
        ev[0].record()
        outs, tape = train.forward_train(net, grd, sat, None, rec=True)
        ev[1].record()
        gout = [torch.ones_like(o) / o.numel() for o in outs]
        ev[2].record()
        grads = train.backward_train(net, tape, gout)
        ev[3].record()
    torch.cuda.synchronize()
    del tape, grads
print("%s B=%d: forward (tape) %.1f ms, backward %.1f ms" % (kind, batch, ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])))
net.eval()
with torch.no_grad():
    net(grd, sat)
    ev[0].record()
    net(grd, sat)
    ev[1].record()
torch.cuda.synchronize()
print("eval forward %.1f ms" % ev[0].elapsed_time(ev[1]))
