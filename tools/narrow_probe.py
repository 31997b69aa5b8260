"""A/B probe of the narrow bf16 decoder kernels (csrc/narrow_impl.h) against the tiled kernels they replace, in ONE process:
python tools/narrow_probe.py [reps].  Per layer: HIP-event time with ccvpe_set_narrow_kernels(1) and (0), interleaved rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops, _lib                  # noqa: E402
from ccvpe_amd.models import _pack_conv          # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()
BF = torch.bfloat16
shapes = [(64, 256, 256, 40, 40, False), (64, 256, 256, 40, 40, True), (64, 256, 256, 32, 32, False), (64, 128, 128, 64, 64, False),
          (64, 128, 128, 80, 80, False),
          (32, 256, 256, 40, 40, False), (256, 256, 256, 32, 32, False)]


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


FEW = os.environ.get("NARROW_PROBE_FEW") == "1"
if FEW:
    shapes = shapes[:1]
for (b, h, w, c, n, f32out) in shapes:
    x = torch.randn((b, h, w, c), device="cuda").to(BF)
    wt = _pack_conv((torch.randn((n, c, 3, 3), device="cuda") * (9 * c) ** -0.5), BF)
    sh = torch.randn((n,), device="cuda") * 0.1

    def run():
        ops.conv_igemm(x, c, wt, n, batch=b, in_h=h, in_w=w, kh=3, kw=3, pad=1, shift=sh, act=ops.ACT_RELU, out_f32=f32out)
    res = {0: [], 1: []}
    for rnd in range(3):
        for on in (1, 0):
            lib.ccvpe_set_narrow_kernels(on)
            res[on].append(timed(run))
    lib.ccvpe_set_narrow_kernels(1)
    m = b * h * w
    fl = 2.0 * m * 9 * c * n
    by = m * (c * 2 + n * (4 if f32out else 2))
    t1, t0 = min(res[1]), min(res[0])
    print("3x3 B%d %dx%d %d->%d %s  narrow %7.1f us (%6.1f TF, %5.2f TB/s)   tiled %7.1f us   x%.2f"
          % (b, h, w, c, n, "f32out" if f32out else "bf16  ", t1, fl / t1 / 1e6, by / t1 / 1e6, t0, t0 / t1), flush=True)


from ccvpe_amd.models import _pack_upconv       # noqa: E402
for (b, h1, cp, cref, cd, c1, co) in [(64, 128, 88, 81, 40, 16, 40), (64, 128, 64, 64, 32, 16, 32), (32, 128, 88, 81, 40, 16, 40),
                                      (64, 128, 136, 129, 32, 16, 32)][:1 if FEW else 4]:
    x = torch.randn((b, h1, h1, cp), device="cuda").to(BF)
    x[..., cref:] = 0
    sk = torch.randn((b, 2 * h1, 2 * h1, c1), device="cuda").to(BF)
    wd = torch.randn((cref, cd, 2, 2), device="cuda") * cref ** -0.5
    bd = torch.randn((cd,), device="cuda") * 0.3
    w3 = torch.randn((co, cd + c1, 3, 3), device="cuda") * (9 * (cd + c1)) ** -0.5
    b3 = torch.randn((co,), device="cuda") * 0.1
    fw, fshift = _pack_upconv(wd, bd, [(0, 0, cref)], cp, w3, b3, BF)

    def run():
        ops.upconv3x3(x, cp, fw, fshift, co, batch=b, h1=h1, w1=h1, src1=sk, c1=c1, act=ops.ACT_RELU)
    res = {0: [], 1: []}
    for rnd in range(3):
        for on in (1, 0):
            lib.ccvpe_set_narrow_kernels(on)
            res[on].append(timed(run))
    lib.ccvpe_set_narrow_kernels(1)
    m = b * h1 * h1 * 4
    fl = 2.0 * m * co * (4 * cref + 9 * c1)
    by = 2.0 * (b * h1 * h1 * cp + m * c1 + m * co)
    t1, t0 = min(res[1]), min(res[0])
    print("up3x3 B%d %dx%d %d|%d->%d  narrow %7.1f us (%6.1f TF, %5.2f TB/s)   tiled %7.1f us   x%.2f"
          % (b, h1, h1, cref, c1, co, t1, fl / t1 / 1e6, by / t1 / 1e6, t0, t0 / t1), flush=True)

