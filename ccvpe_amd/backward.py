"""Backward building blocks of the path (fp32): thin wrappers over the backward entry points of libccvpe_hip.so.
ccvpe_amd/train.py assembles them into the model's single autograd node.

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


def conv_wgrad(x0, dy, n, kh, kw, stride, pad, x1=None, c0=None, c1=None, want_bias=False):
    """d(conv2d)/d(weight) in the reference OIHW layout.  x0 [B,H,W,ld0] (+ x1 [B,H,W,ld1] concatenated),
    dy [B,Ho,Wo,ldy]; c0 / c1 select the first channels of wider pixel rows (default: all).
    want_bias: also return the conv's bias gradient sum_pixels dy[:, n] from the SAME launch (ccvpe_conv_wgrad_bias_f32)."""
    lib = _lib.load()
    for t, nm in ((x0, "x0"), (x1, "x1"), (dy, "dy")):
        ops._chk(t, nm)
    b, h, w, ld0 = x0.shape
    ld1 = x1.shape[-1] if x1 is not None else 0
    c0 = ld0 if c0 is None else c0
    c1 = ld1 if c1 is None else c1
    ctot = c0 + c1
    nfl = lib.ccvpe_conv_wgrad_scratch_floats(b, h, w, kh, kw, stride, pad, ctot, n)
    if nfl <= 0:
        raise _lib.CcvpeError("ccvpe_conv_wgrad_scratch_floats rejected the shape")
    scratch = torch.empty((nfl,), device=x0.device, dtype=torch.float32)
    dw = torch.empty((n, kh * kw, ctot), device=x0.device, dtype=torch.float32)
    rec = ops._recorder
    ev0 = rec.begin() if rec is not None else None
    dbias = torch.empty((n,), device=x0.device, dtype=torch.float32) if want_bias else None
    if want_bias:
        check(lib.ccvpe_conv_wgrad_bias_f32(ops._ptr(x0), c0, ld0, ops._ptr(x1), c1, ld1, ops._ptr(dy), dy.shape[-1],
                                            ops._ptr(dw), ops._ptr(dbias), ops._ptr(scratch), b, h, w, kh, kw, stride, pad, n,
                                            ops._stream()), "ccvpe_conv_wgrad_bias_f32")
    else:
        check(lib.ccvpe_conv_wgrad_f32(ops._ptr(x0), c0, ld0, ops._ptr(x1), c1, ld1, ops._ptr(dy), dy.shape[-1], ops._ptr(dw),
                                       ops._ptr(scratch), b, h, w, kh, kw, stride, pad, n, ops._stream()),
              "ccvpe_conv_wgrad_f32")
    if rec is not None:
        m = dy.numel() // dy.shape[-1]
        tile = lib.ccvpe_conv_wgrad_tile(n, kh * kw * ctot)
        rec.end("conv_wgrad_kernel<%d,%d>" % (tile >> 16, tile & 0xffff), "%dx%d s%d M%d N%d C%d" % (kh, kw, stride, m, n, ctot),
                2.0 * m * n * kh * kw * ctot, 4.0 * (b * h * w * ctot + m * n + n * kh * kw * ctot), ev0)
    dw = dw.reshape(n, kh, kw, ctot).permute(0, 3, 1, 2)                   # OIHW view
    return (dw, dbias) if want_bias else dw


def conv1x1_wgrad_gated(u, gate, dy, n):
    """Weight gradient [n, C, 1, 1] of a 1x1 conv whose input is u[b, px, c] * gate[b, c] (the SE product in front of the MBConv
    projection) without materialising that product (ccvpe_conv_wgrad_gated_f32)."""
    lib = _lib.load()
    for t, nm in ((u, "u"), (gate, "gate"), (dy, "dy")):
        ops._chk(t, nm)
    b, h, w, c = u.shape
    nfl = lib.ccvpe_conv_wgrad_scratch_floats(b, h, w, 1, 1, 1, 0, c, n)
    if nfl <= 0:
        raise _lib.CcvpeError("ccvpe_conv_wgrad_scratch_floats rejected the shape")
    scratch = torch.empty((nfl,), device=u.device, dtype=torch.float32)
    dw = torch.empty((n, 1, c), device=u.device, dtype=torch.float32)
    rec = ops._recorder
    ev0 = rec.begin() if rec is not None else None
    check(lib.ccvpe_conv_wgrad_gated_f32(ops._ptr(u), c, c, ops._ptr(gate), ops._ptr(dy), dy.shape[-1], ops._ptr(dw),
                                         ops._ptr(scratch), b, h, w, n, ops._stream()), "ccvpe_conv_wgrad_gated_f32")
    if rec is not None:
        m = b * h * w
        tile = lib.ccvpe_conv_wgrad_tile(n, c)
        rec.end("conv_wgrad_kernel<%d,%d>" % (tile >> 16, tile & 0xffff), "1x1 gated M%d N%d C%d" % (m, n, c),
                2.0 * m * n * c, 4.0 * (m * c + m * n + n * c), ev0)
    return dw.reshape(n, 1, 1, c).permute(0, 3, 1, 2)


def deconv_wgrad(x, dy_hi):
    """d(ConvTranspose2d k2 s2)/d(weight) [Cin,Cout,2,2]: x [B,H,W,Cin] forward input, dy_hi [B,2H,2W,Cout]."""
    cin = x.shape[-1]
    g = conv_wgrad(dy_hi, x, cin, 2, 2, 2, 0)                               # [Cin, Cout, 2, 2] already
    return g


def bias_grad(dy, channels=None):
    """Column sums of dy [.., ld]; `channels` restricts them to the first columns of wider pixel rows."""
    lib = _lib.load()
    ops._chk(dy, "dy")
    ld = dy.shape[-1]
    c = ld if channels is None else channels
    rows = dy.numel() // ld
    out = torch.empty((c,), device=dy.device, dtype=torch.float32)
    scratch = torch.empty((((rows + 255) // 256) * c,), device=dy.device, dtype=torch.float32)
    check(lib.ccvpe_colsum_f32(ops._ptr(dy), rows, c, ld, ops._ptr(out), ops._ptr(scratch), ops._stream()), "ccvpe_colsum_f32")
    return out


def conv1x1_dgrad(dy, w, wp=None):
    """w [N,C,1,1] -> dx [B,H,W,C].  wp: the weight already in the backward layout (models._pack_backward)."""
    b, h, wd, n = dy.shape
    c = w.shape[1]
    if wp is None:
        wp = _pack_conv(w.permute(1, 0, 2, 3))
    return ops.conv_igemm(dy, n, wp, c, batch=b, in_h=h, in_w=wd)


def conv3x3_dgrad(dy, w, relu_out=None, wp=None):
    """w [N,C,3,3] (stride 1, pad 1) -> dx [B,H,W,C] (split it along C for a two-source forward).
    relu_out [B,H,W,C]: the convolution's input was this ReLU output; return the gradient in FRONT of the ReLU
    (dx where relu_out > 0, else 0) from the same launch instead of a relu_bwd pass over dx."""
    b, h, wd, n = dy.shape
    c = w.shape[1]
    if wp is None:
        wp = _pack_conv(w.flip(2, 3).permute(1, 0, 2, 3).contiguous())
    if relu_out is None:
        return ops.conv_igemm(dy, n, wp, c, batch=b, in_h=h, in_w=wd, kh=3, kw=3, pad=1)
    return ops.conv_igemm(dy, n, wp, c, batch=b, in_h=h, in_w=wd, kh=3, kw=3, pad=1, residual=relu_out,
                          act=ops.ACT_RELU_MASK)


def deconv_dgrad(dy_hi, w):
    """ConvTranspose2d weight w [Cin,Cout,2,2]; dy_hi [B,2H,2W,Cout] -> dx [B,H,W,Cin]."""
    b, h2, w2, cout = dy_hi.shape
    cin = w.shape[0]
    return ops.conv_igemm(dy_hi, cout, _pack_conv(w), cin, batch=b, in_h=h2, in_w=w2, kh=2, kw=2, stride=2)


def conv2x2s2_dgrad(dy, w, wp=None):
    """Forward conv 2x2 stride 2 with w [N,C,2,2]; dy [B,H/2,W/2,N] -> dx [B,H,W,C]."""
    b, h, wd, n = dy.shape
    c = w.shape[1]
    if wp is None:
        wp, _ = _pack_deconv(w, w.new_zeros((c,)), [(0, 0, n)], n)
    return ops.conv_igemm(dy, n, wp, 4 * c, batch=b, in_h=h, in_w=wd, out_mode=ops.OUT_DECONV2X)


# ---- EfficientNet train-mode pieces (csrc/train_bwd.hip) ----------------------------------------------

def _bc(x):
    b, c = x.shape[0], x.shape[-1]
    return b, x.numel() // (b * c), c


def bn_act_bwd(x, dv, mean, var, gamma, beta, eps, act, gate=None, dmean=None, dc_scale=None):
    """Backward of ops.bn_act: returns (dx, dgamma, dbeta).  gate / dmean [B,C] fold the SE branches in."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (dv, "dv"), (mean, "mean"), (var, "var"), (gamma, "gamma"), (beta, "beta"), (gate, "gate"),
                  (dmean, "dmean"), (dc_scale, "dc_scale")):
        ops._chk(t, nm)
    b, rps, c = _bc(x)
    dx = torch.empty_like(x)
    both = torch.empty((2, c), device=x.device, dtype=torch.float32)       # dbeta, dgamma adjacent: one merge launch
    dbeta, dgamma = both[0], both[1]
    scratch = torch.empty((2 * c * b * lib.ccvpe_bn_bwd_nblk(rps),), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_bn_act_bwd_f32(ops._ptr(x), ops._ptr(dv), ops._ptr(mean), ops._ptr(var), ops._ptr(gamma), ops._ptr(beta),
                                   ops._ptr(gate), ops._ptr(dmean), ops._ptr(dc_scale), float(eps), act, ops._ptr(dx),
                                   ops._ptr(dgamma), ops._ptr(dbeta), ops._ptr(scratch), b, rps, c, ops._stream()),
          "ccvpe_bn_act_bwd_f32")
    return dx, dgamma, dbeta


def se_dgate_partials(x, dv, mean, var, gamma, beta, eps, act):
    lib = _lib.load()
    for t, nm in ((x, "x"), (dv, "dv")):
        ops._chk(t, nm)
    b, rps, c = _bc(x)
    part = torch.empty((b, lib.ccvpe_bn_bwd_nblk(rps), c), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_se_dgate_f32(ops._ptr(x), ops._ptr(dv), ops._ptr(mean), ops._ptr(var), ops._ptr(gamma), ops._ptr(beta),
                                 float(eps), act, ops._ptr(part), b, rps, c, ops._stream()), "ccvpe_se_dgate_f32")
    return part


def se_bn_bwd_reduce(x, dv, mean, var, gamma, beta, eps, act):
    """First pass of the BatchNorm + squeeze-excite backward: A [5, B, C] (ccvpe_se_bn_bwd_reduce_f32); A[0] is dgate."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (dv, "dv"), (mean, "mean"), (var, "var"), (gamma, "gamma"), (beta, "beta")):
        ops._chk(t, nm)
    b, rps, c = _bc(x)
    a = torch.empty((5, b, c), device=x.device, dtype=torch.float32)
    scratch = torch.empty((b * lib.ccvpe_bn_bwd_nblk(rps) * 5 * c,), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_se_bn_bwd_reduce_f32(ops._ptr(x), ops._ptr(dv), ops._ptr(mean), ops._ptr(var), ops._ptr(gamma), ops._ptr(beta),
                                         float(eps), act, ops._ptr(a), ops._ptr(scratch), b, rps, c, ops._stream()),
          "ccvpe_se_bn_bwd_reduce_f32")
    return a


def se_bn_bwd_apply(x, dv, mean, var, gamma, beta, eps, act, gate, dmean, a):
    """Second pass: dbeta / dgamma from A, gate and dmean, then dx.  Returns (dx, dgamma, dbeta)."""
    lib = _lib.load()
    for t, nm in ((gate, "gate"), (dmean, "dmean"), (a, "A")):
        ops._chk(t, nm)
    b, rps, c = _bc(x)
    dx = torch.empty_like(x)
    both = torch.empty((2, c), device=x.device, dtype=torch.float32)
    dbeta, dgamma = both[0], both[1]
    check(lib.ccvpe_se_bn_bwd_apply_f32(ops._ptr(x), ops._ptr(dv), ops._ptr(mean), ops._ptr(var), ops._ptr(gamma), ops._ptr(beta),
                                        ops._ptr(gate), ops._ptr(dmean), float(eps), act, ops._ptr(a), ops._ptr(dx),
                                        ops._ptr(dgamma), ops._ptr(dbeta), b, rps, c, ops._stream()), "ccvpe_se_bn_bwd_apply_f32")
    return dx, dgamma, dbeta


def se_bwd(se_partial, hw, dgate_partial, w1, b1, w2t, b2):
    """Returns (dmean [B,C] (already / HW), dw1 [Cs,C], db1, dw2 [C,Cs], db2)."""
    lib = _lib.load()
    for t, nm in ((se_partial, "se_partial"), (dgate_partial, "dgate_partial"), (w1, "w1"), (b1, "b1"), (w2t, "w2t"),
                  (b2, "b2")):
        ops._chk(t, nm)
    b, nblk, c = se_partial.shape
    cs = w1.shape[0]
    dev = se_partial.device
    dmean = torch.empty((b, c), device=dev, dtype=torch.float32)
    dw1 = torch.empty((cs, c), device=dev, dtype=torch.float32)
    db1 = torch.empty((cs,), device=dev, dtype=torch.float32)
    dw2 = torch.empty((c, cs), device=dev, dtype=torch.float32)
    db2 = torch.empty((c,), device=dev, dtype=torch.float32)
    scratch = torch.empty((2 * b * (c + cs),), device=dev, dtype=torch.float32)
    check(lib.ccvpe_se_bwd_f32(ops._ptr(se_partial), nblk, 1.0 / float(hw), ops._ptr(dgate_partial), dgate_partial.shape[1],
                               ops._ptr(w1), ops._ptr(b1), ops._ptr(w2t), ops._ptr(b2), ops._ptr(dmean), ops._ptr(dw1),
                               ops._ptr(db1), ops._ptr(dw2), ops._ptr(db2), ops._ptr(scratch), b, c, cs, ops._stream()),
          "ccvpe_se_bwd_f32")
    return dmean, dw1, db1, dw2, db2


def dwconv_dgrad(dy, w, in_h, in_w, k, stride, circular, w_flipped=None):
    """w: packed depthwise weights [k*k, C] or [k, k, C].  Stride 1 is the forward depthwise kernel with the taps
    reversed (symmetric SAME padding, zero or circular): the register-sliding strip kernel instead of a gather."""
    if stride == 1:
        c = dy.shape[-1]
        if w_flipped is None:
            w_flipped = w.reshape(k * k, c).flip(0).contiguous()
        return ops.dwconv_raw(dy, w_flipped, k, 1, circular)
    lib = _lib.load()
    ops._chk(dy, "dy")
    ops._chk(w, "w")
    b, c = dy.shape[0], dy.shape[-1]
    dx = torch.empty((b, in_h, in_w, c), device=dy.device, dtype=torch.float32)
    check(lib.ccvpe_dwconv_dgrad_f32(ops._ptr(dy), ops._ptr(w), ops._ptr(dx), b, in_h, in_w, c, k, stride,
                                     int(bool(circular)), ops._stream()), "ccvpe_dwconv_dgrad_f32")
    return dx


def dwconv_wgrad(x, dy, k, stride, circular):
    """Depthwise weight gradient in the packed layout [k*k][C]."""
    lib = _lib.load()
    ops._chk(x, "x")
    ops._chk(dy, "dy")
    b, h, w, c = x.shape
    nblk = lib.ccvpe_dwconv_wgrad_nblk(h, w, k, stride)
    scratch = torch.empty((b * nblk * k * k * c,), device=x.device, dtype=torch.float32)
    dw = torch.empty((k * k, c), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_dwconv_wgrad_f32(ops._ptr(x), ops._ptr(dy), ops._ptr(dw), ops._ptr(scratch), b, h, w, c, k, stride,
                                     int(bool(circular)), ops._stream()), "ccvpe_dwconv_wgrad_f32")
    return dw


def relu_bwd(y, dy):
    lib = _lib.load()
    ops._chk(y, "y")
    ops._chk(dy, "dy")
    dx = torch.empty_like(dy)
    check(lib.ccvpe_relu_bwd_f32(ops._ptr(y), ops._ptr(dy), ops._ptr(dx), dy.numel(), ops._stream()), "ccvpe_relu_bwd_f32")
    return dx


# ---- heads / glue (csrc/heads_bwd.hip, csrc/matching_bwd.hip) --------------------------------------------

HEAD_BWD_SCRATCH = (1024 + 1) * 2 * 145       # CCVPE_HEAD_BWD_SCRATCH


def softmax_bwd(heatmap, dheatmap, dlogits_direct=None):
    lib = _lib.load()
    for t, nm in ((heatmap, "heatmap"), (dheatmap, "dheatmap"), (dlogits_direct, "dlogits_direct")):
        ops._chk(t, nm)
    rows, n = heatmap.shape
    out = torch.empty_like(heatmap)
    check(lib.ccvpe_softmax_bwd_f32(ops._ptr(heatmap), ops._ptr(dheatmap), ops._ptr(dlogits_direct), ops._ptr(out), rows, n,
                                    ops._stream()), "ccvpe_softmax_bwd_f32")
    return out


def l2norm2_bwd(raw, dout):
    lib = _lib.load()
    ops._chk(raw, "raw")
    ops._chk(dout, "dout")
    b = raw.shape[0]
    hw = raw.numel() // (2 * b)
    out = torch.empty_like(raw)
    check(lib.ccvpe_l2norm2_bwd_f32(ops._ptr(raw), ops._ptr(dout), ops._ptr(out), b, hw, ops._stream()),
          "ccvpe_l2norm2_bwd_f32")
    return out


def head_conv3x3_bwd(x, w, dout, relu_mask_x=False):
    """x [B,H,W,16], w [cout,3,3,16] packed, dout [B,cout,H,W] -> (dx, dw [cout,3,3,16], dbias [cout]).
    relu_mask_x: x is a ReLU output and dx is wanted in front of that ReLU (dx zeroed where x <= 0)."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (w, "w"), (dout, "dout")):
        ops._chk(t, nm)
    b, h, wd, _ = x.shape
    cout = dout.shape[1]
    dx = torch.empty_like(x)
    dw = torch.empty((cout, 3, 3, 16), device=x.device, dtype=torch.float32)
    db = torch.empty((cout,), device=x.device, dtype=torch.float32)
    scratch = torch.empty((HEAD_BWD_SCRATCH,), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_head_conv3x3_bwd_f32(ops._ptr(x), ops._ptr(w), ops._ptr(dout), ops._ptr(dx), ops._ptr(dw), ops._ptr(db),
                                         ops._ptr(scratch), b, h, wd, cout, int(bool(relu_mask_x)), ops._stream()),
          "ccvpe_head_conv3x3_bwd_f32")
    return dx, dw, db


def ground_descriptor_bwd(y1, wh, cd, dout):
    """-> (dy1 [B,h,w,ld], dwh [6,h], dbh [6])."""
    lib = _lib.load()
    for t, nm in ((y1, "y1"), (wh, "wh"), (dout, "dout")):
        ops._chk(t, nm)
    b, h, w, ld = y1.shape
    cd_arr = (ctypes.c_int * 6)(*cd)
    dy1 = torch.empty_like(y1)
    dwh = torch.empty((6, h), device=y1.device, dtype=torch.float32)
    dbh = torch.empty((6,), device=y1.device, dtype=torch.float32)
    check(lib.ccvpe_ground_descriptor_bwd_f32(ops._ptr(y1), ld, ops._ptr(wh), cd_arr, ops._ptr(dout), ops._ptr(dy1),
                                              ops._ptr(dwh), ops._ptr(dbh), b, h, w, ops._stream()),
          "ccvpe_ground_descriptor_bwd_f32")
    return dy1, dwh, dbh


def add_cols(src, col_off, channels, dst, accumulate=True):
    """dst[..., :channels] (+)= src[..., col_off:col_off+channels] (both NHWC, any pixel strides)."""
    lib = _lib.load()
    ops._chk(src, "src")
    ops._chk(dst, "dst")
    rows = src.numel() // src.shape[-1]
    check(lib.ccvpe_add_cols_f32(ops._ptr(src), src.shape[-1], col_off, ops._ptr(dst), dst.shape[-1], channels, rows,
                                 int(bool(accumulate)), ops._stream()), "ccvpe_add_cols_f32")
    return dst


def stem_conv_wgrad(img, dy, circular):
    """img [B,3,H,W] NCHW, dy [B,H/2,W/2,32] -> dw in the packed layout [3,3,3,32] (ky,kx,ci,co)."""
    lib = _lib.load()
    ops._chk(img, "img")
    ops._chk(dy, "dy")
    b, _, h, w = img.shape
    scratch = torch.empty((864 * lib.ccvpe_stem_wgrad_nblk(b, h, w),), device=img.device, dtype=torch.float32)
    dw = torch.empty((3, 3, 3, 32), device=img.device, dtype=torch.float32)
    check(lib.ccvpe_stem_conv_wgrad_f32(ops._ptr(img), ops._ptr(dy), ops._ptr(dw), ops._ptr(scratch), b, h, w,
                                        int(bool(circular)), ops._stream()), "ccvpe_stem_conv_wgrad_f32")
    return dw


def match_level_bwd(x, g, L, shifts, n_max, n_tail, stride, scores, dscores, ddst, channels, dg_out, window_offset=0):
    """Backward of ops.match_level.  x [B,H,W,ldx]; g [B,ldg] view; scores/dscores [B,n,H,W]; ddst [B,H,W,ldo];
    dg_out [B,ldg'] view that receives dg (first L entries per row).  Returns dx [B,H,W,channels]."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (scores, "scores"), (dscores, "dscores"), (ddst, "ddst")):
        ops._chk(t, nm)
    b, h, w, ldx = x.shape
    hw = h * w
    n = len(shifts)
    sh = (ctypes.c_int * n)(*shifts)
    dx = torch.empty((b, h, w, channels), device=x.device, dtype=torch.float32)
    scratch = torch.empty((b * lib.ccvpe_match_bwd_nblk(hw, b, channels) * (L + 1),), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_match_level_bwd_f32(ops._ptr(x), ldx, ops._ptr(g), g.stride(0), L, sh, n, n_max, n_tail, stride, window_offset,
                                        ops._ptr(scores), ops._ptr(dscores), ops._ptr(ddst), ddst.shape[-1], ops._ptr(dx),
                                        channels, ops._ptr(dg_out), dg_out.stride(0), ops._ptr(scratch), b, hw, channels,
                                        ops._stream()), "ccvpe_match_level_bwd_f32")
    return dx


def gate_mul(u, gate):
    lib = _lib.load()
    ops._chk(u, "u")
    ops._chk(gate, "gate")
    b, rps, c = _bc(u)
    v = torch.empty_like(u)
    check(lib.ccvpe_gate_mul_f32(ops._ptr(u), ops._ptr(gate), ops._ptr(v), b, rps, c, ops._stream()), "ccvpe_gate_mul_f32")
    return v
