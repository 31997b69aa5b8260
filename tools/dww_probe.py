"""Micro-probe of the depthwise weight gradient (csrc/train_bwd.hip: dw_wgrad_kernel + the partial-row merge) and the depthwise
data gradient on the 16 depthwise layers of one EfficientNet-B0 encoder at B = 64 (aerial 512 x 512 image; `ground` = 320 x 640):
    python tools/dww_probe.py [reps] [aerial|ground]
TB/s = (x + dy) bytes once / time.  Buffers rotate so that every launch reads cold HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import backward as bw       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ground = len(sys.argv) > 2 and sys.argv[2] == "ground"
B = 64
s0 = (160, 320) if ground else (256, 256)
# (block, C, plane divisor, k, stride)
layers = [(0, 32, 1, 3, 1), (1, 96, 1, 3, 2), (2, 144, 2, 3, 1), (3, 144, 2, 5, 2), (4, 240, 4, 5, 1), (5, 240, 4, 3, 2), (6, 480, 8, 3, 1),
          (7, 480, 8, 3, 1), (8, 480, 8, 5, 1), (9, 672, 8, 5, 1), (10, 672, 8, 5, 1), (11, 672, 8, 5, 2), (12, 1152, 16, 5, 1),
          (13, 1152, 16, 5, 1), (14, 1152, 16, 5, 1), (15, 1152, 16, 3, 1)]
seen = {}


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


tw = td = 0.0
for (blk, c, div, k, st) in layers:
    h, w = s0[0] // div, s0[1] // div
    key = (c, h, w, k, st)
    if key not in seen:
        ho, wo = (h + st - 1) // st, (w + st - 1) // st
        nb = 4.0 * B * c * (h * w + ho * wo)
        nbuf = max(2, min(6, int(1.5e9 / nb) + 1))
        xs = [torch.randn((B, h, w, c), device="cuda") for _ in range(nbuf)]
        dys = [torch.randn((B, ho, wo, c), device="cuda") for _ in range(nbuf)]
        wp = torch.randn((k * k, c), device="cuda")
        t_w = timed([lambda i=i: bw.dwconv_wgrad(xs[i], dys[i], k, st, ground) for i in range(nbuf)])
        t_d = timed([lambda i=i: bw.dwconv_dgrad(dys[i], wp, h, w, k, st, ground) for i in range(nbuf)])
        seen[key] = (t_w, t_d, nb)
        del xs, dys
        torch.cuda.empty_cache()
    t_w, t_d, nb = seen[key]
    tw += t_w
    td += t_d
    print("block %2d  C %4d %3dx%-3d k%d s%d (%6.0f MB)  wgrad %7.1f us (%.2f TB/s)  dgrad %7.1f us (%.2f TB/s)"
          % (blk, c, h, w, k, st, nb / 1e6, t_w, nb / t_w * 1e-6, t_d, nb / t_d * 1e-6), flush=True)
print("sum over the encoder: wgrad %.0f us, dgrad %.0f us" % (tw, td))
