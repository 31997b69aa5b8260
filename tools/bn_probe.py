"""Micro-probe of the train-mode BatchNorm backward (csrc/train_bwd.hip) on the BN layers of one EfficientNet-B0 encoder of the
B = 64 VIGOR training step (aerial: 512 x 512 image; `ground` = 320 x 640), fp32:
    python tools/bn_probe.py [reps] [aerial|ground]
per layer shape: the plain backward (reduce + apply), the SE form's reduce and apply, and a torch `add(x, dv, out=dx)` as the
yardstick of a 2-read-1-write streaming pass on the same tensors.  Buffers rotate so that every launch reads cold HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops, backward as bw       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ground = len(sys.argv) > 2 and sys.argv[2] == "ground"
B = 64
s = (160, 320) if ground else (256, 256)


def px(div):
    return (s[0] // div) * (s[1] // div)


# (name, channels, pixels, count in the encoder, se form?)
layers = [("stem / b0 dw", 32, px(1), 2, 1), ("b0 proj", 16, px(1), 1, 0), ("b1 exp", 96, px(1), 1, 0), ("b1 dw", 96, px(2), 1, 1),
          ("b1-2 proj", 24, px(2), 2, 0), ("b2-3 exp, b2 dw", 144, px(2), 3, 1), ("b3 dw", 144, px(4), 1, 1), ("b3-4 proj", 40, px(4), 2, 0),
          ("b4-5 exp, b4 dw", 240, px(4), 3, 1), ("b5 dw", 240, px(8), 1, 1), ("b5-7 proj", 80, px(8), 3, 0), ("b6-8 exp / dw", 480, px(8), 6, 1),
          ("b8-10 proj", 112, px(8), 3, 0), ("b9-11 exp / dw", 672, px(8), 5, 1), ("b11 dw", 672, px(16), 1, 1), ("b11-14 proj", 192, px(16), 4, 0),
          ("b12-15 exp / dw", 1152, px(16), 8, 1), ("b15 proj", 320, px(16), 1, 0)]


def timed(fns):
    n = len(fns)
    for f in fns[:2]:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % n]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {}
for (nm, c, rows, cnt, se) in layers:
    nbytes = 4.0 * B * rows * c
    nbuf = max(2, min(6, int(1.5e9 / nbytes) + 1))
    xs = [torch.randn((B, rows, c), device="cuda") for _ in range(nbuf)]
    dvs = [torch.randn((B, rows, c), device="cuda") for _ in range(nbuf)]
    mean, var = torch.randn((c,), device="cuda") * 0.1, torch.rand((c,), device="cuda") + 0.5
    gamma, beta = torch.rand((c,), device="cuda") + 0.5, torch.randn((c,), device="cuda") * 0.1
    gate, dmean = torch.rand((B, c), device="cuda"), torch.randn((B, c), device="cuda") * 1e-3
    out = torch.empty_like(xs[0])
    t_add = timed([lambda i=i: torch.add(xs[i], dvs[i], out=out) for i in range(nbuf)])
    t_plain = timed([lambda i=i: bw.bn_act_bwd(xs[i], dvs[i], mean, var, gamma, beta, 1e-3, ops.ACT_SWISH) for i in range(nbuf)])
    a = bw.se_bn_bwd_reduce(xs[0], dvs[0], mean, var, gamma, beta, 1e-3, ops.ACT_SWISH)
    t_red = timed([lambda i=i: bw.se_bn_bwd_reduce(xs[i], dvs[i], mean, var, gamma, beta, 1e-3, ops.ACT_SWISH) for i in range(nbuf)])
    t_app = timed([lambda i=i: bw.se_bn_bwd_apply(xs[i], dvs[i], mean, var, gamma, beta, 1e-3, ops.ACT_SWISH, gate, dmean, a) for i in range(nbuf)])
    gbs = lambda passes, us: passes * nbytes / us * 1e-6 / 1e3       # noqa: E731  TB/s
    print("%-18s C %4d px %6d (%6.0f MB) x%d  add %7.1f us (%.2f TB/s)  bwd %7.1f (%.2f over 5 passes)  se-reduce %7.1f (%.2f)  apply %7.1f (%.2f)"
          % (nm, c, rows, nbytes / 1e6, cnt, t_add, gbs(3, t_add), t_plain, gbs(5, t_plain), t_red, gbs(2, t_red), t_app, gbs(3, t_app)), flush=True)
    for k, v in (("add", t_add), ("bwd", t_plain), ("se_reduce", t_red), ("apply", t_app)):
        tot[k] = tot.get(k, 0.0) + v * cnt
    del xs, dvs, out
    torch.cuda.empty_cache()
print("weighted by layer count (us): " + "  ".join("%s %.0f" % kv for kv in tot.items()))
