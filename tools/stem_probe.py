"""Micro-probe: stem conv + block-0 depthwise conv as two launches vs the fused launch (csrc/stem_dw.hip), B = 64, the aerial
(512 x 512) and ground (320 x 640, circular) images of CVM_VIGOR:  python tools/stem_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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


few = os.environ.get("STEM_PROBE_FEW") == "1"      # ablation runs: the aerial shape, fused launch only
for (h, w, circ) in ((512, 512, False),) if few else ((512, 512, False), (320, 640, True)):
    x = torch.randn((b, 3, h, w), device="cuda")
    wt = torch.randn((3, 3, 3, 32), device="cuda") * 0.2
    wd = torch.randn((3, 3, 32), device="cuda") * 0.3
    s0, b0, s1, b1 = (torch.rand(32, device="cuda") + 0.5, torch.randn(32, device="cuda") * 0.1,
                      torch.rand(32, device="cuda") + 0.5, torch.randn(32, device="cuda") * 0.1)
    for dt in (torch.bfloat16, torch.float32):
        y = ops.stem_conv(x, wt, s0, b0, circ, out_dtype=dt)
        t_stem = 0.0 if few else timed(lambda: ops.stem_conv(x, wt, s0, b0, circ, out_dtype=dt))
        t_dw = 0.0 if few else timed(lambda: ops.dwconv(y, wd, s1, b1, 3, 1, circ))
        t_f = timed(lambda: ops.stem_dw(x, wt, s0, b0, wd, s1, b1, circ, out_dtype=dt))
        nbytes = 4.0 * x.numel() + y.numel() * y.element_size()
        print("%dx%d %-5s  stem %6.1f + dw %6.1f = %6.1f us   fused %6.1f us (%.2f TB/s of image + output)   x%.2f"
              % (h, w, "bf16" if dt == torch.bfloat16 else "fp32", t_stem, t_dw, t_stem + t_dw, t_f, nbytes / t_f / 1e6,
                 (t_stem + t_dw) / t_f))
