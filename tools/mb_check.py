import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from ccvpe_amd import ops, synth
from ccvpe_amd.models import _pack_conv
from oracle import ccvpe_oracle as O
k, s, cin, h, w, circ = 3, 2, 16, 40, 72, False
b, mid = 2, 6 * cin
x = synth.normal((b, cin, h, w), 300 + cin)
w_exp = synth.normal((mid, cin, 1, 1), 301, (2.0 / cin) ** 0.5)
s0, b0 = synth.uniform((mid,), 302, 0.5, 1.5), synth.normal((mid,), 303, 0.2)
w_dw = synth.normal((mid, 1, k, k), 304, 1.0 / k)
s1, b1 = synth.uniform((mid,), 305, 0.5, 1.5), synth.normal((mid,), 306, 0.2)
t = O.swish(F.conv2d(x, w_exp) * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1))
want = O.swish(O.same_conv(t, w_dw, k, s, 224, circ, groups=mid) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
got, part = ops.mbconv_front(nh(x), _pack_conv(w_exp).cuda(), s0.cuda(), b0.cuda(), w_dw.reshape(mid, k, k).permute(1, 2, 0).contiguous().cuda(), s1.cuda(), b1.cuda(), mid, k, s, circ)
got = got.permute(0, 3, 1, 2).cpu()
err = (got - want).abs()
print("shape", got.shape, "max err", err.max().item())
print("per 16-ch chunk max err:", [round(err[:, c:c + 16].max().item(), 3) for c in range(0, mid, 16)])
print("per channel-in-chunk (cg) err chunk0:", [round(err[:, c].max().item(), 3) for c in range(16)])
print("rows err:", [round(err[:, :, r].max().item(), 2) for r in range(got.shape[2])])
print("cols err:", [round(err[:, :, :, c].max().item(), 2) for c in range(got.shape[3])])
print("sample got/want:", got[0, 0, 0, :6], want[0, 0, 0, :6])
