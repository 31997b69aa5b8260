"""Micro-probe of ccvpe_match_level on the six levels of C2 (CVM_VIGOR, 20 hypotheses, B = 32; the 256 x 256 level in fp32 = the
bf16 path's fp32-class tail) and of C4 (ori_prior(180), FoV 180: partial windows, 21 hypotheses, B = 256 — B = 64 here), matrix-core
form vs vector-ALU form, interleaved in one process:   python tools/match_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvpe_amd import ops, _lib       # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()
BF = torch.bfloat16
# (label, B, C, L, stride, hw, n_shifts, dtype)
cases = []
for (c, hw, st) in ((640, 16, 32), (320, 32, 16), (160, 64, 8), (80, 128, 4)):
    cases.append(("C2", 32, c, c, st, hw, 20, BF))
cases.append(("C2 tail", 32, 40, 40, 2, 256, 20, torch.float32))
for (c, hw, st) in ((640, 16, 32), (320, 32, 16), (160, 64, 8), (80, 128, 4)):
    cases.append(("C4", 64, c, c // 2, st, hw, 21, BF))
cases.append(("C4 tail", 64, 40, 20, 2, 256, 21, torch.float32))


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


tot = [0.0, 0.0, 0.0]
for (label, b, c, L, st, hw, nsh, dt) in cases:
    x = torch.randn((b, hw, hw, c), device="cuda").to(dt)
    g = torch.randn((b, L), device="cuda")
    shifts = list(range(nsh)) if L == c else list(range(-(nsh // 2), nsh - nsh // 2))
    ldo = (c + 1 + 7) // 8 * 8
    fn = lambda: ops.match_level(x, g, L, shifts, nsh, 0, st, ldo)
    lib.ccvpe_set_match_mfma(0)
    t0 = timed(fn)
    lib.ccvpe_set_match_mfma(1)
    t1 = timed(fn)
    lib.ccvpe_set_match_mfma(2)
    t2 = timed(fn)
    esz = 2 if dt == BF else 4
    nbytes = b * hw * hw * (esz * c + esz * ldo + 4 * nsh)
    tot[0] += t0
    tot[1] += t1
    tot[2] += t2
    print("%-8s B%-3d C%-4d L%-4d %3dx%-3d n%-2d %s  valu %7.1f us (%4.2f TB/s)   mfma tiled %7.1f us (%4.2f TB/s)   + streaming %7.1f us (%4.2f TB/s)" % (
        label, b, c, L, hw, hw, nsh, "bf16" if dt == BF else "fp32", t0, nbytes / t0 / 1e6, t1, nbytes / t1 / 1e6, t2, nbytes / t2 / 1e6), flush=True)
print("sum: valu %.0f us, mfma tiled %.0f us, with the streaming form of the narrow levels %.0f us" % tuple(tot))
