"""Micro-probe of the pointwise (1x1) GEMM on encoder-shaped layers: python tools/pw_probe.py [fp32|bf16] [reps]
Prints HIP-event times per shape; run under rocprofv3 --pmc for counters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops, synth                 # noqa: E402
from ccvpe_amd.models import _pack_conv          # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.float32 if prec == "fp32" else torch.bfloat16
shapes = [(65536, 112, 672), (65536, 672, 112), (65536, 80, 480), (65536, 480, 80), (16384, 192, 1152), (16384, 1152, 192),
          (16384, 320, 1280), (4194304, 32, 16), (1048576, 144, 24), (262144, 240, 40)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sys.argv[3].split(","))]
for (m, k, n) in shapes:
    b = 64
    hw = m // b
    h = int(round(hw ** 0.5))
    w = hw // h
    x = torch.randn((b, h, w, k), device="cuda").to(dt)
    wt = _pack_conv((torch.randn((n, k, 1, 1), device="cuda") * k ** -0.5), dt)
    sc = torch.rand((n,), device="cuda") + 0.5
    sh = torch.randn((n,), device="cuda") * 0.1
    for _ in range(3):
        ops.conv_igemm(x, k, wt, n, batch=b, in_h=h, in_w=w, scale=sc, shift=sh, act=ops.ACT_SWISH)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_igemm(x, k, wt, n, batch=b, in_h=h, in_w=w, scale=sc, shift=sh, act=ops.ACT_SWISH)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * m * k * n
    by = (m * k + m * n + n * k) * (4 if dt == torch.float32 else 2)
    print("%s M%-8d K%-5d N%-5d  %8.1f us  %7.1f TF  %7.0f GB/s" % (prec, m, k, n, ms * 1e3, fl / ms / 1e9, by / ms / 1e6))
