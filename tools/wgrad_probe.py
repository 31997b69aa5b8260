"""Micro-probe of the conv weight-gradient kernel on decoder shapes: python tools/wgrad_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import backward as bw       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# (batch, hw, N, C, k)
shapes = [(64, 256, 40, 56, 3), (64, 128, 80, 104, 3), (64, 512, 16, 16, 3), (64, 16, 640, 1344, 3), (64, 64, 160, 200, 3),
          (64, 32, 320, 432, 3), (64, 256, 32, 48, 3), (64, 32, 672, 112, 1), (64, 32, 112, 672, 1)]
if os.environ.get("PROBE_SHAPE"):
    shapes = [tuple(int(v) for v in os.environ["PROBE_SHAPE"].split(","))]
for (b, hw, n, c, k) in shapes:
    x = torch.randn((b, hw, hw, c), device="cuda")
    dy = torch.randn((b, hw, hw, n), device="cuda")
    for _ in range(2):
        bw.conv_wgrad(x, dy, n, k, k, 1, k // 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        bw.conv_wgrad(x, dy, n, k, k, 1, k // 2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = b * hw * hw
    print("wgrad %dx%d M%-9d N%-4d C%-5d %9.1f us %7.1f TF %7.0f GB/s" % (k, k, m, n, c, ms * 1e3, 2.0 * m * n * c * k * k / ms / 1e9,
                                                                           4.0 * m * (n + c) / ms / 1e6))
