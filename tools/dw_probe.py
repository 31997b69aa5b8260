"""Micro-probe of the depthwise-conv backward kernels on EfficientNet-B0 shapes: python tools/dw_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import backward as bw       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# (k, s, C, h, w) aerial encoder, B = 64
shapes = [(3, 1, 32, 256, 256), (3, 2, 96, 256, 256), (3, 1, 144, 128, 128), (5, 2, 144, 128, 128), (5, 1, 240, 64, 64),
          (3, 2, 240, 64, 64), (3, 1, 480, 32, 32), (5, 1, 672, 32, 32), (5, 2, 672, 32, 32), (5, 1, 1152, 16, 16), (3, 1, 1152, 16, 16)]
b = 64
for (k, s, c, h, w) in shapes:
    tot = (k - 1) if s == 1 else (k - 2)
    ho, wo = (h + tot - k) // s + 1, (w + tot - k) // s + 1
    x = torch.randn((b, h, w, c), device="cuda")
    dy = torch.randn((b, ho, wo, c), device="cuda")
    wt = torch.randn((k * k, c), device="cuda")
    res = []
    for name, fn in (("wgrad", lambda: bw.dwconv_wgrad(x, dy, k, s, False)), ("dgrad", lambda: bw.dwconv_dgrad(dy, wt, h, w, k, s, False))):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append("%s %7.1f us %5.0f GB/s" % (name, ms * 1e3, 4.0 * (x.numel() + dy.numel()) / ms / 1e6))
    print("k%d s%d C%-4d %3dx%-3d  %s" % (k, s, c, h, w, "   ".join(res)))
