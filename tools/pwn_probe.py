"""Micro-probe of the narrow projection kernel (csrc/pwn.hip; `python tools/pwn_probe.py [reps] [fp32]`) on the early-block projections of both encoders at B = 64, against
the generic kernel on the same tensors, interleaved in one process:   python tools/pwn_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops, _lib       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"
lib = _lib.load()
BF = torch.float32 if F32 else torch.bfloat16
# (block, cin, cout, h, w, residual)
layers = [(0, 32, 16, 256, 256, 0), (1, 96, 24, 128, 128, 0), (2, 144, 24, 128, 128, 1), (3, 144, 40, 64, 64, 0), (4, 240, 40, 64, 64, 1),
          (0, 32, 16, 160, 320, 0), (1, 96, 24, 80, 160, 0), (2, 144, 24, 80, 160, 1), (3, 144, 40, 40, 80, 0), (4, 240, 40, 40, 80, 1)]
b = 64


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0]
for (blk, cin, cout, h, w, res) in layers:
    x = torch.randn((b, h, w, cin), device="cuda").to(BF)
    km = 16 if F32 else 32
    kp = (cin + km - 1) // km * km
    wt = torch.zeros((16 * ((cout + 15) // 16), kp), device="cuda")
    wt[:cout, :cin] = torch.randn((cout, cin), device="cuda") * cin ** -0.5
    wt = wt.to(BF)
    sc, sh = torch.rand((cout,), device="cuda") + 0.5, torch.randn((cout,), device="cuda") * 0.1
    gate = torch.rand((b, cin), device="cuda")
    r = torch.randn((b, h, w, cout), device="cuda").to(BF) if res else None
    fn = lambda: ops.conv_igemm(x, cin, wt, cout, batch=b, in_h=h, in_w=w, scale=sc, shift=sh, gate=gate, residual=r)
    lib.ccvpe_set_pwn_kernels(0)
    t0 = timed(fn)
    lib.ccvpe_set_pwn_kernels(1)
    t1 = timed(fn)
    nbytes = (4.0 if F32 else 2.0) * b * h * w * (cin + cout * (2 if res else 1))
    tot[0] += t0
    tot[1] += t1
    print("block %d %3d -> %2d %3dx%-3d %s  generic %6.1f us (%4.2f TB/s)   streaming %6.1f us (%4.2f TB/s)" % (
        blk, cin, cout, h, w, "skip" if res else "    ", t0, nbytes / t0 / 1e6, t1, nbytes / t1 / 1e6), flush=True)
print("sum: generic %.0f us, streaming %.0f us" % tuple(tot))
