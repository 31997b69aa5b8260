"""Practical matrix-core ceiling of this GPU: TFLOP/s of the vendor GEMM (hipBLASLt / rocBLAS through torch.mm) in fp32 and
bf16 on shapes like the decoder's convolutions seen as GEMMs.  Context for the roofline fractions in bench.py: a hand-written
kernel is judged against the dense peak of MI355X_MICROARCH.md, but what the vendor library reaches on the same silicon says
how much of that peak is attainable under the chip's power management.
    python tools/gemm_ceiling.py [out.json]"""
import json
import sys
import torch

assert torch.cuda.is_available()
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda:0"
shapes = [(8192, 8192, 8192), (16384, 640, 12096), (65536, 320, 2880), (262144, 160, 1440), (4096, 4096, 4096)]
res = {}
for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
    for m, n, k in shapes:
        a = torch.randn((m, k), device=dev, dtype=dt)
        b = torch.randn((k, n), device=dev, dtype=dt)
        for _ in range(3):
            torch.mm(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            torch.mm(a, b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * m * n * k / ms / 1e9
        res["%s_%dx%dx%d" % (name, m, n, k)] = {"ms": round(ms, 4), "tflops": round(tf, 1)}
        print(name, m, n, k, "%.3f ms  %.1f TFLOP/s" % (ms, tf))
        del a, b
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
