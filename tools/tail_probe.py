"""Times the fused 512 x 512 level (ccvpe_tail512) against the unfused pair (upconv3x3 + head_conv3x3) at the benched size
(B = 64, 256 x 256 low-res), fp32 / bf16, loc / ori shapes.   gpurun -- python tools/tail_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccvpe_amd import ops, synth
from ccvpe_amd.models import _pack_upconv


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    b, h1 = int(os.environ.get("B", 64)), 256
    for cp, cref, cout, dt in ((48, 41, 1, torch.float32), (32, 32, 2, torch.float32), (32, 32, 2, torch.bfloat16), (48, 41, 1, torch.bfloat16)):
        x = synth.normal((b, h1, h1, cp), 1).to(dt).cuda()
        wd = synth.normal((cref, 16, 2, 2), 702, (1.0 / cref) ** 0.5).cuda()
        bd = synth.normal((16,), 703, 0.3).cuda()
        w3 = synth.normal((16, 16, 3, 3), 704, (1.0 / 144) ** 0.5).cuda()
        b3 = synth.normal((16,), 705, 0.1).cuda()
        w2 = synth.normal((cout, 3, 3, 16), 706, (1.0 / 144) ** 0.5).cuda()
        b2 = synth.normal((cout,), 707, 0.1).cuda()
        fw, fshift = _pack_upconv(wd, bd, [(0, 0, cref)], cp, w3, b3, dt)
        m = 4 * b * h1 * h1
        gf = 2.0 * m * 16 * 4 * cp / 1e9

        def unfused():
            y = ops.upconv3x3(x, cp, fw, fshift, 16, batch=b, h1=h1, w1=h1, act=ops.ACT_RELU)
            return ops.head_conv3x3(y, w2, b2, cout, cout == 2)
        t0 = timeit(unfused)
        line = "%s cp=%d cout=%d: unfused %.3f ms" % ("f32" if dt == torch.float32 else "bf16", cp, cout, t0)
        ref = unfused()
        f = lambda: ops.tail512(x, cp, fw, fshift, w2, b2, cout, cout == 2, batch=b, h1=h1, w1=h1)
        t = timeit(f)
        err = float((f() - ref).abs().max())
        line += " | fused %.3f ms (%.0f TF on %.0f GFLOP, max diff vs unfused %.2e)" % (t, gf / t, gf, err)
        if dt == torch.float32 and cout == 1:
            f2 = lambda: ops.tail512(x, cp, fw, fshift, w2, b2, cout, False, batch=b, h1=h1, w1=h1, split=True)
            t2 = timeit(f2)
            line += " | split %.3f ms (max diff vs fp32 fused %.2e of %.2e)" % (t2, float((f2() - f()).abs().max()), float(f().abs().max()))
        print(line, flush=True)


if __name__ == "__main__":
    main()
