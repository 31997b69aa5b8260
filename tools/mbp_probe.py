"""Micro-probe of the late-block MBConv front (csrc/mbconv_plane.hip) on the EfficientNet-B0 blocks 6-15 of both encoders at
B = 64, against the round-5 chain on the same tensors, interleaved in one process:
    python tools/mbp_probe.py [reps] [bf16|fp32] [plane KB ...]
columns: chain = pointwise GEMM + dwconv_plane_kernel (mode 0); dw = pointwise GEMM + the depthwise-only form (mode 2);
fused = one launch (mode 3), once per LDS budget given; band = the band-owner kernel (mode 7, bf16 only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops, _lib       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
budgets = [int(a) for a in sys.argv[3:]] or [72]
kmult = 16 if dt == torch.float32 else 32
lib = _lib.load()
# (block, k, s, cin, mid, h, w, circular)
blocks = [(6, 3, 1, 80, 480, 32, 32, 0), (8, 5, 1, 80, 480, 32, 32, 0), (9, 5, 1, 112, 672, 32, 32, 0), (11, 5, 2, 112, 672, 32, 32, 0),
          (12, 5, 1, 192, 1152, 16, 16, 0), (15, 3, 1, 192, 1152, 16, 16, 0),
          (6, 3, 1, 80, 480, 20, 40, 1), (8, 5, 1, 80, 480, 20, 40, 1), (9, 5, 1, 112, 672, 20, 40, 1), (11, 5, 2, 112, 672, 20, 40, 1),
          (12, 5, 1, 192, 1152, 10, 20, 1), (15, 3, 1, 192, 1152, 10, 20, 1)]
if os.environ.get("MBP_FEW"):
    blocks = [blocks[i] for i in (0, 2, 4, 8, 10)]
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


tot = {}
for (blk, k, s, cin, mid, h, w, circ) in blocks:
    x = torch.randn((b, h, w, cin), device="cuda").to(dt)
    kp = (cin + kmult - 1) // kmult * kmult
    we = torch.zeros((mid, kp), device="cuda")
    we[:, :cin] = torch.randn((mid, cin), device="cuda") * (2.0 / cin) ** 0.5
    we = we.to(dt)
    s0, b0, s1, b1 = (torch.rand((mid,), device="cuda") + 0.5 for _ in range(4))
    wd = torch.randn((k, k, mid), device="cuda") * 0.2

    def chain():
        t = ops.conv_igemm(x, cin, we, mid, batch=b, in_h=h, in_w=w, scale=s0, shift=b0, act=ops.ACT_SWISH)
        return ops.dwconv(t, wd, s1, b1, k, s, circ)

    def expand_only():
        return ops.conv_igemm(x, cin, we, mid, batch=b, in_h=h, in_w=w, scale=s0, shift=b0, act=ops.ACT_SWISH)

    def fused():
        return ops.mbconv_front(x, we, s0, b0, wd, s1, b1, mid, k, s, circ)

    res = []
    if os.environ.get("MBP_BAND_ONLY"):
        lib.ccvpe_set_mbconv_plane_kernels(7 | (72 << 8))
        print("block %2d k%d s%d %3d->%4d %2dx%-2d  band %6.1f" % (blk, k, s, cin, mid, h, w, timed(fused)), flush=True)
        continue
    lib.ccvpe_set_mbconv_plane_kernels(0)
    res.append(("expand", timed(expand_only)))
    res.append(("chain", timed(chain)))
    lib.ccvpe_set_mbconv_plane_kernels(2 | (budgets[0] << 8))
    res.append(("dw", timed(chain)))
    y0 = None
    for kb in budgets:
        lib.ccvpe_set_mbconv_plane_kernels(3 | (kb << 8))
        nb = ops.mbconv_front_supported(h, w, cin, mid, k, s)
        res.append(("fused@%dKB(nb%d)" % (kb, nb), timed(fused)))
    if dt == torch.bfloat16:
        lib.ccvpe_set_mbconv_plane_kernels(7 | (72 << 8))
        res.append(("band", timed(fused)))
    lib.ccvpe_set_mbconv_plane_kernels(7 | (72 << 8))
    for nm, us in res:
        tot[nm.split("(")[0]] = tot.get(nm.split("(")[0], 0.0) + us
    print("block %2d k%d s%d %3d->%4d %2dx%-2d  " % (blk, k, s, cin, mid, h, w) + "  ".join("%s %6.1f" % r for r in res), flush=True)
print("sum (one launch of each row; a forward runs blocks 6-7, 9-10, 12-14 more than once): " + "  ".join("%s %.0f" % kv for kv in tot.items()))
