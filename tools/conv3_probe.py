"""Micro-probe of the 3x3 convolution on decoder-shaped layers: python tools/conv3_probe.py [fp32|bf16] [reps] [B,H,W,C,N]
Prints HIP-event times per shape (run under rocprofv3 --pmc for counters)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops                        # noqa: E402
from ccvpe_amd.models import _pack_conv          # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.float32 if prec == "fp32" else torch.bfloat16
shapes = [(64, 16, 16, 1344, 640), (64, 16, 16, 640, 640), (64, 32, 32, 320, 320), (64, 64, 64, 160, 160), (64, 128, 128, 80, 80),
          (64, 16, 16, 1344, 1344), (64, 32, 32, 256, 256), (64, 256, 256, 40, 40), (64, 512, 512, 16, 16)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sys.argv[3].split(","))]
for (b, h, w, c, n) in shapes:
    x = torch.randn((b, h, w, c), device="cuda").to(dt)
    wt = _pack_conv((torch.randn((n, c, 3, 3), device="cuda") * (9 * c) ** -0.5), dt)
    sh = torch.randn((n,), device="cuda") * 0.1
    if os.environ.get("PROBE_ZERO") == "1":          # all-zero operands: same instructions, no data toggling in the matrix cores
        x.zero_()
        wt.zero_()
    for _ in range(3):
        ops.conv_igemm(x, c, wt, n, batch=b, in_h=h, in_w=w, kh=3, kw=3, pad=1, shift=sh, act=ops.ACT_RELU)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_igemm(x, c, wt, n, batch=b, in_h=h, in_w=w, kh=3, kw=3, pad=1, shift=sh, act=ops.ACT_RELU)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = b * h * w
    fl = 2.0 * m * 9 * c * n
    print("%s B%d %dx%d C%-5d N%-5d  %8.1f us  %7.1f TF" % (prec, b, h, w, c, n, ms * 1e3, fl / ms / 1e9), flush=True)
