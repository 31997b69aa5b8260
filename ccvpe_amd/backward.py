"""Backward building blocks for the dense convolutions of the path (fp32).  NOT yet wired into autograd:
these are the tested pieces the training path will be assembled from.

  * weight gradients: ccvpe_conv_wgrad_f32 (MFMA pixel-reduction GEMM, csrc/conv_wgrad.hip);
  * input gradients: the FORWARD kernels with re-packed weights —
      1x1 conv            -> 1x1 conv with W^T
      3x3 conv (s1, p1)   -> 3x3 conv with the taps flipped and in/out channels swapped
      ConvTranspose2d k2s2-> conv 2x2 stride 2 of the output gradient
      conv 2x2 stride 2   -> the deconv (pixel-shuffle) GEMM of the output gradient
  * bias gradients: ccvpe_colsum_f32.
Every function takes/returns NHWC fp32 device tensors; weights are in the reference layout.
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check
from .models import _pack_conv, _pack_deconv


def conv_wgrad(x0, dy, n, kh, kw, stride, pad, x1=None):
    """d(conv2d)/d(weight) in the reference OIHW layout.  x0 [B,H,W,c0] (+ x1 [B,H,W,c1] concatenated),
    dy [B,Ho,Wo,n]."""
    lib = _lib.load()
    for t, nm in ((x0, "x0"), (x1, "x1"), (dy, "dy")):
        ops._chk(t, nm)
    b, h, w, c0 = x0.shape
    c1 = x1.shape[-1] if x1 is not None else 0
    ctot = c0 + c1
    nfl = lib.ccvpe_conv_wgrad_scratch_floats(b, h, w, kh, kw, stride, pad, ctot, n)
    if nfl <= 0:
        raise _lib.CcvpeError("ccvpe_conv_wgrad_scratch_floats rejected the shape")
    scratch = torch.empty((nfl,), device=x0.device, dtype=torch.float32)
    dw = torch.empty((n, kh * kw, ctot), device=x0.device, dtype=torch.float32)
    check(lib.ccvpe_conv_wgrad_f32(ops._ptr(x0), c0, c0, ops._ptr(x1), c1, c1, ops._ptr(dy), dy.shape[-1], ops._ptr(dw),
                                   ops._ptr(scratch), b, h, w, kh, kw, stride, pad, n, ops._stream()),
          "ccvpe_conv_wgrad_f32")
    return dw.reshape(n, kh, kw, ctot).permute(0, 3, 1, 2)                 # OIHW view


def deconv_wgrad(x, dy_hi):
    """d(ConvTranspose2d k2 s2)/d(weight) [Cin,Cout,2,2]: x [B,H,W,Cin] forward input, dy_hi [B,2H,2W,Cout]."""
    cin = x.shape[-1]
    g = conv_wgrad(dy_hi, x, cin, 2, 2, 2, 0)                               # [Cin, Cout, 2, 2] already
    return g


def bias_grad(dy):
    lib = _lib.load()
    ops._chk(dy, "dy")
    c = dy.shape[-1]
    rows = dy.numel() // c
    out = torch.empty((c,), device=dy.device, dtype=torch.float32)
    scratch = torch.empty((((rows + 255) // 256) * c,), device=dy.device, dtype=torch.float32)
    check(lib.ccvpe_colsum_f32(ops._ptr(dy), rows, c, c, ops._ptr(out), ops._ptr(scratch), ops._stream()), "ccvpe_colsum_f32")
    return out


def conv1x1_dgrad(dy, w):
    """w [N,C,1,1] -> dx [B,H,W,C]."""
    b, h, wd, n = dy.shape
    c = w.shape[1]
    return ops.conv_igemm(dy, n, _pack_conv(w.permute(1, 0, 2, 3)), c, batch=b, in_h=h, in_w=wd)


def conv3x3_dgrad(dy, w):
    """w [N,C,3,3] (stride 1, pad 1) -> dx [B,H,W,C] (split it along C for a two-source forward)."""
    b, h, wd, n = dy.shape
    c = w.shape[1]
    wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()
    return ops.conv_igemm(dy, n, _pack_conv(wt), c, batch=b, in_h=h, in_w=wd, kh=3, kw=3, pad=1)


def deconv_dgrad(dy_hi, w):
    """ConvTranspose2d weight w [Cin,Cout,2,2]; dy_hi [B,2H,2W,Cout] -> dx [B,H,W,Cin]."""
    b, h2, w2, cout = dy_hi.shape
    cin = w.shape[0]
    return ops.conv_igemm(dy_hi, cout, _pack_conv(w), cin, batch=b, in_h=h2, in_w=w2, kh=2, kw=2, stride=2)


def conv2x2s2_dgrad(dy, w):
    """Forward conv 2x2 stride 2 with w [N,C,2,2]; dy [B,H/2,W/2,N] -> dx [B,H,W,C]."""
    b, h, wd, n = dy.shape
    c = w.shape[1]
    wp, _ = _pack_deconv(w, w.new_zeros((c,)), [(0, 0, n)], n)
    return ops.conv_igemm(dy, n, wp, 4 * c, batch=b, in_h=h, in_w=wd, out_mode=ops.OUT_DECONV2X)
