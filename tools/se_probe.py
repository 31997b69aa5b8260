"""se_gate per MBConv block shape at B = 64 (aerial encoder): which launches are the slow ones.  gpurun -- python tools/se_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccvpe_amd import _lib, ops, synth

lib = _lib.load()
b = int(os.environ.get("B", 64))
h = w = 256
for i, (k, s, e, cin, cout) in enumerate(synth.B0_BLOCKS):
    mid = cin * e
    nb = lib.ccvpe_mbconv_front_nblk(h, w, cin, mid, k, s) if e != 1 else 0
    if nb == 0:
        nb = lib.ccvpe_dwconv_nblk(h, w, mid, s)
    ho, wo = (h + s - 1) // s, (w + s - 1) // s
    cs = max(1, cin // 4)
    part = torch.randn((b, nb, mid), device="cuda")
    w1 = torch.randn((cs, mid), device="cuda") * 0.1
    b1 = torch.zeros((cs,), device="cuda")
    w2 = torch.randn((cs, mid), device="cuda") * 0.1
    b2 = torch.zeros((mid,), device="cuda")
    for _ in range(3):
        ops.se_gate(part, ho * wo, w1, b1, w2, b2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.se_gate(part, ho * wo, w1, b1, w2, b2)
    e1.record()
    torch.cuda.synchronize()
    print("block %2d C %4d Cs %3d nblk %3d: %.1f us" % (i, mid, cs, nb, 1e3 * e0.elapsed_time(e1) / 20), flush=True)
    h, w = ho, wo
