"""Ablation of conv3x3_kernel<float,4,5,2> (run on the MI355X box): builds variants of csrc/conv_igemm.hip
with pieces of the K-stage loop removed (results are WRONG by construction, only the time matters) and
times them on the level-6 shape (M=16384, N=640, C=640).  Usage: python tools/ablation/run_ablation.py
Variants: 0 full | 1 no W global loads / LDS stores in the loop | 2 = 1 + no barriers |
          3 = 2 + no LDS fragment reads (MFMA on registers only)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SRC = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "conv_igemm.hip")).read()


def variant(n):
    s = SRC
    a = s.index("template <typename T, int MT, int NT, int WN>\n__global__ __launch_bounds__(256) void conv3x3_kernel(")
    b = s.index("// ConvTranspose2d(k2,s2) folded into the following 3x3 conv")
    k = s[a:b]
    if n >= 1:
        k = k.replace("    if (more) load_w(nchunk, ntap);\n", "")
        k = k.replace("    if (more) store_w((s + 1) & 1);\n", "")
        k = k.replace("    if (next_halo) load_halo(chunk + 1);\n", "")
    if n >= 2:
        k = k.replace("    if (more) store_w((s + 1) & 1);\n    __syncthreads();", "    __syncthreads();")
        k = k.replace("    __syncthreads();\n    if (tap == 8 && more) {", "    if (false) {")
    if n >= 3:
        k = k.replace("    f32x4 af[MT], bf[NT];\n#pragma unroll\n    for (int i = 0; i < MT; ++i)\n      af[i] = *reinterpret_cast<const f32x4*>(hb +",
                      "    f32x4 af[MT], bf[NT];\n#pragma unroll\n    for (int i = 0; i < MT; ++i)\n      if (s == 0) af[i] = *reinterpret_cast<const f32x4*>(hb +")
        k = k.replace("    for (int j = 0; j < NT; ++j)\n      bf[j] = *reinterpret_cast<const f32x4*>(&Bs[s & 1]",
                      "    for (int j = 0; j < NT; ++j)\n      if (s == 0) bf[j] = *reinterpret_cast<const f32x4*>(&Bs[s & 1]")
        k = k.replace("    f32x4 af[MT], bf[NT];", "    static_assert(true, \"\");")
        k = k.replace("  int chunk = 0, tap = 0;\n  for (int s = 0; s < nstages; ++s) {", "  f32x4 af[MT], bf[NT];\n  int chunk = 0, tap = 0;\n  for (int s = 0; s < nstages; ++s) {")
    return s[:a] + k + s[b:]


def main():
    out = os.path.join(ROOT, "gpurun_out", "abl")
    os.makedirs(out, exist_ok=True)
    from ccvpe_amd import _lib
    from ccvpe_amd.models import _pack_conv
    b, hw, c, n = 64, 16, 640, 640
    x = torch.randn(b, hw, hw, c, device="cuda")
    w = _pack_conv(torch.randn(n, c, 3, 3, device="cuda") * 0.01)
    bias = torch.zeros(n, device="cuda")
    dst = torch.empty(b, hw, hw, n, device="cuda")
    for v in range(4):
        src = os.path.join(out, "conv_abl%d.hip" % v)
        open(src, "w").write(variant(v).replace('#include "common.h"', '#include "%s/ccvpe_amd/csrc/common.h"' % ROOT)
                             + "\nthread_local char ccvpe::g_err[512] = \"\";\n")
        so = os.path.join(out, "libabl%d.so" % v)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", src, "-o", so],
                           capture_output=True, text=True)
        if r.returncode:
            print("variant", v, "build failed:\n", r.stderr[-1500:])
            continue
        lib = ctypes.CDLL(so)
        d = _lib.ConvDesc()
        d.src0, d.w, d.shift, d.dst = x.data_ptr(), w.data_ptr(), bias.data_ptr(), dst.data_ptr()
        d.c0, d.ld0, d.batch, d.in_h, d.in_w = c, c, b, hw, hw
        d.kh = d.kw = 3
        d.stride, d.pad, d.n, d.kpad, d.ldd, d.act = 1, 1, n, w.shape[1], n, 1
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.ccvpe_conv_igemm_f32.restype = ctypes.c_int
        for _ in range(3):
            assert lib.ccvpe_conv_igemm_f32(ctypes.byref(d), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.ccvpe_conv_igemm_f32(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tf = 2.0 * b * hw * hw * n * 9 * c / (ms * 1e-3) / 1e12
        print("variant %d: %.3f ms  %.1f TFLOP/s-equivalent" % (v, ms, tf))


if __name__ == "__main__":
    main()
