"""Where do the bf16 path's orientation outliers come from?  For 256 seeded pairs: the fp32 path's UN-normalised orientation
2-vector (conv1_ori's output, models.py:341's input) at the arg-max pixel, and the bf16 path's (cos, sin) error there.
    python tools/ori_norm_scan.py [pairs]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import models, synth       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sd = synth.synthetic_state_dict("vigor", 0)


def build(precision):
    net = models.CVM_VIGOR("cuda", True)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").eval()
    return net.set_precision("bf16") if precision == "bf16" else net


n32, nbf = build("fp32"), build("bf16")
n32.ori_raw_output = True
rows = []
scale = None
for c0 in range(0, n, 16):
    grd, sat = synth.synthetic_pair(16, "vigor", 5000 + c0, device="cuda")
    r = n32(grd, sat)
    raw = r[2].clone()
    ia = r[0].argmax(1)
    g = nbf(grd, sat)
    ref = F.normalize(raw, p=2, dim=1)
    for b in range(16):
        rv = raw[b].reshape(2, -1)[:, ia[b]]
        err = float((g[2][b].reshape(2, -1)[:, ia[b]] - ref[b].reshape(2, -1)[:, ia[b]]).abs().max())
        rows.append((float(rv.norm()), err, float(raw[b].reshape(2, -1).norm(dim=0).median())))
rows.sort()
print("raw |v| at the arg-max pixel (sorted), bf16 error of the unit vector there, err x |v|, median |v| of the field")
for nv, e, med in rows[:40] + rows[-8:]:
    print("%.4e  %.3e  %.3e  %.3e" % (nv, e, e * nv, med))
import statistics
print("max err x |v| = %.3e ; median |v| = %.3e" % (max(e * nv for nv, e, _ in rows), statistics.median(nv for nv, _, _ in rows)))
