"""Micro-probe of the folded deconv + 3x3 kernels on the decoder's shapes: python tools/up_probe.py [reps] [bf16]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
# (batch, h1, c0, c1, n): loc levels 5..1, ori levels 5..2
shapes = [(64, 16, 648, 112, 320), (64, 32, 328, 40, 160), (64, 64, 168, 24, 80), (64, 128, 88, 16, 40), (64, 256, 48, 0, 16),
          (64, 16, 640, 112, 256), (64, 32, 256, 40, 128), (64, 64, 128, 24, 64), (64, 128, 64, 16, 32)]
if os.environ.get("PROBE_SHAPE"):
    shapes = [tuple(int(v) for v in os.environ["PROBE_SHAPE"].split(","))]
kmult = 16 if dt == torch.float32 else 32
for (b, h1, c0, c1, n) in shapes:
    k = 4 * c0 + 9 * c1
    kp = (k + kmult - 1) // kmult * kmult
    x = torch.randn((b, h1, h1, c0), device="cuda").to(dt)
    sk = torch.randn((b, 2 * h1, 2 * h1, c1), device="cuda").to(dt) if c1 else None
    w = (torch.randn((4, (n + 15) // 16 * 16, kp), device="cuda") * 0.01).to(dt)
    sh = torch.zeros((9, n), device="cuda")
    for _ in range(2):
        ops.upconv3x3(x, c0, w, sh, n, batch=b, h1=h1, w1=h1, src1=sk, c1=c1, act=ops.ACT_RELU)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.upconv3x3(x, c0, w, sh, n, batch=b, h1=h1, w1=h1, src1=sk, c1=c1, act=ops.ACT_RELU)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = 4 * b * h1 * h1
    print("upconv h1=%-3d c0=%-4d c1=%-3d N=%-3d %9.1f us %7.1f TF" % (h1, c0, c1, n, ms * 1e3, 2.0 * m * n * k / ms / 1e9))
