"""Micro-probe of the fused MBConv front kernel on the EfficientNet-B0 blocks it serves (aerial 256^2 and ground 160x320 stem
outputs): python tools/mb_probe.py [reps] [bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
kmult = 16 if dt == torch.float32 else 32
# (block, k, s, cin, mid, h, w)
blocks = [(1, 3, 2, 16, 96, 256, 256), (2, 3, 1, 24, 144, 128, 128), (3, 5, 2, 24, 144, 128, 128), (4, 5, 1, 40, 240, 64, 64),
          (5, 3, 2, 40, 240, 64, 64), (1, 3, 2, 16, 96, 160, 320), (2, 3, 1, 24, 144, 80, 160), (3, 5, 2, 24, 144, 80, 160),
          (4, 5, 1, 40, 240, 40, 80), (5, 3, 2, 40, 240, 40, 80)]
b = 64
for (blk, k, s, cin, mid, h, w) in blocks:
    x = torch.randn((b, h, w, cin), device="cuda").to(dt)
    kp = (cin + kmult - 1) // kmult * kmult
    we = (torch.randn((mid, kp), device="cuda") * 0.2).to(dt)
    s0, b0, s1, b1 = (torch.rand((mid,), device="cuda") + 0.5 for _ in range(4))
    wd = torch.randn((k, k, mid), device="cuda") * 0.2
    for _ in range(2):
        y, part = ops.mbconv_front(x, we, s0, b0, wd, s1, b1, mid, k, s, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.mbconv_front(x, we, s0, b0, wd, s1, b1, mid, k, s, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    esz = 4 if dt == torch.float32 else 2
    nbytes = esz * (x.numel() + y.numel())
    print("block %d k%d s%d %3d->%3d  %3dx%-3d %8.1f us  %6.0f GB/s (x read once + y written once)" % (blk, k, s, cin, mid, h, w, ms * 1e3, nbytes / ms / 1e6))
