"""bf16 storage path (BASELINE configs C2 / C4) on the MI355X.

Operators: inputs and weights are rounded to bf16 first, the expected value is the fp32 oracle op
on those rounded values, so what is compared is accumulation order + ONE output rounding:
tolerance 1e-2 of the tensor's scale (bf16 has 8 significand bits: 2^-9 = 2e-3 per rounding).
Whole forward: ~60 layers of bf16 re-rounding; documented tolerance 5e-2 of the logit range, the
oracle's arg-max pixel must be within that tolerance of the bf16 maximum, matching scores within
2e-2 absolute (they are cosines in [-1, 1]).  The fp32 path keeps the 1e-3 / exact-arg-max bar.
"""
import pytest
import torch
import torch.nn.functional as F

from ccvpe_amd import synth
from oracle import ccvpe_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from ccvpe_amd import ops as _ops, _lib
    _lib.load()
    return _ops


def r(t):
    """round to bf16 and back (the values the device actually sees)"""
    return t.to(BF).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def dev(t, dt=BF):
    return t.to(dt).cuda().contiguous()


def close(got, want, tol, what):
    got, want = got.detach().float().cpu().double(), want.detach().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e)" % (what, err, scale, err / scale)


def pack(w):
    from ccvpe_amd.models import _pack_conv
    return _pack_conv(w, BF)


@pytest.mark.parametrize("cin,cout", [(16, 96), (24, 144), (40, 240), (112, 672), (192, 1152), (96, 24), (480, 112),
                                      (320, 1280), (32, 16), (1152, 320), (240, 80), (64, 64), (8, 160), (72, 48)])
def test_igemm_bf16_1x1(ops, cin, cout):
    b, h, w = 2, 9, 13
    x = r(synth.normal((b, cin, h, w), 100 + cin))
    wt = r(synth.normal((cout, cin, 1, 1), 200 + cout, (1.0 / cin) ** 0.5))
    sc = synth.uniform((cout,), 300, 0.5, 1.5)
    sh = synth.normal((cout,), 301, 0.1)
    want = O.swish(F.conv2d(x, wt) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    got = ops.conv_igemm(dev(nhwc(x)), cin, dev(pack(wt)), cout, batch=b, in_h=h, in_w=w,
                         scale=dev(sc, torch.float32), shift=dev(sh, torch.float32), act=ops.ACT_SWISH)
    assert got.dtype == BF
    close(nchw(got), want, 1e-2, "bf16 1x1 %d->%d" % (cin, cout))
    got32 = ops.conv_igemm(dev(nhwc(x)), cin, dev(pack(wt)), cout, batch=b, in_h=h, in_w=w,
                           scale=dev(sc, torch.float32), shift=dev(sh, torch.float32), act=ops.ACT_SWISH, out_f32=True)
    assert got32.dtype == torch.float32
    close(nchw(got32), want, 2e-3, "bf16 1x1 fp32-out")


@pytest.mark.parametrize("k,s,cin,h,w,circ", [(3, 1, 80, 32, 32, False), (5, 1, 112, 32, 32, False), (5, 2, 112, 32, 32, False),
                                              (5, 1, 192, 16, 16, False), (5, 1, 112, 20, 40, True), (5, 2, 112, 20, 40, True),
                                              (3, 1, 192, 10, 20, True), (5, 1, 80, 7, 12, True)])
def test_mbconv_plane_late_blocks_bf16(ops, k, s, cin, h, w, circ):
    """bf16 storage of x / w_exp / y; the expanded plane and the depthwise arithmetic stay fp32 in LDS (better than the
    round-5 chain, which rounded the expanded tensor to bf16 on its way through HBM)."""
    b, mid = 3, 6 * cin
    x = r(synth.normal((b, cin, h, w), 700 + cin + h))
    w_exp = r(synth.normal((mid, cin, 1, 1), 701, (2.0 / cin) ** 0.5))
    s0, b0 = synth.uniform((mid,), 702, 0.5, 1.5), synth.normal((mid,), 703, 0.2)
    w_dw = synth.normal((mid, 1, k, k), 704, 1.0 / k)
    s1, b1 = synth.uniform((mid,), 705, 0.5, 1.5), synth.normal((mid,), 706, 0.2)
    t = O.swish(F.conv2d(x, w_exp) * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1))
    want = O.swish(O.same_conv(t, w_dw, k, s, 224, circ, groups=mid) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    f32 = torch.float32
    assert ops.mbconv_front_supported(h, w, cin, mid, k, s) > 0
    wd = dev(w_dw.reshape(mid, k, k).permute(1, 2, 0), f32)
    got, part = ops.mbconv_front(dev(nhwc(x)), dev(pack(w_exp)), dev(s0, f32), dev(b0, f32), wd, dev(s1, f32), dev(b1, f32),
                                 mid, k, s, circ)
    assert got.dtype == BF
    close(nchw(got), want, 1e-2, "bf16 plane front k%d s%d" % (k, s))
    close(part.sum(1), want.sum(dim=(2, 3)), 2e-3, "squeeze partials (fp32 sums of unrounded outputs)")
    # the default above is the band-owner kernel (producer / consumer waves); the slice-per-workgroup kernel behind the same entry
    # point must agree with it to the last bf16 bit or two (same arithmetic, another summation order inside the MFMA chain)
    from ccvpe_amd import _lib
    lib = _lib.load()
    prev = lib.ccvpe_set_mbconv_plane_kernels(3)
    try:
        got1, part1 = ops.mbconv_front(dev(nhwc(x)), dev(pack(w_exp)), dev(s0, f32), dev(b0, f32), wd, dev(s1, f32), dev(b1, f32),
                                       mid, k, s, circ)
    finally:
        lib.ccvpe_set_mbconv_plane_kernels(prev)
    close(nchw(got1), want, 1e-2, "bf16 plane front (slice kernel) k%d s%d" % (k, s))
    close(got1, got.float().cpu(), 8e-3, "slice kernel vs band kernel")
    close(part1, part.cpu(), 1e-4, "squeeze partials: slice kernel vs band kernel")
    tr = r(t)
    want2 = O.swish(O.same_conv(tr, w_dw, k, s, 224, circ, groups=mid) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    got2, part2 = ops.dwconv(dev(nhwc(tr)), wd, dev(s1, f32), dev(b1, f32), k, s, circ)
    close(nchw(got2), want2, 1e-2, "bf16 plane depthwise k%d s%d" % (k, s))
    close(part2.sum(1), want2.sum(dim=(2, 3)), 2e-3, "squeeze partials (depthwise form)")


@pytest.mark.parametrize("cin,cout,b,hw,with_res", [(32, 16, 1, 256, False), (96, 24, 4, 128, False), (144, 24, 4, 128, True),
                                                     (144, 40, 16, 64, False), (240, 40, 17, 64, True), (96, 24, 3, 160, False)])
def test_narrow_projection_kernel(ops, cin, cout, b, hw, with_res):
    """The MBConv projections of the early blocks (efficientnet_pytorch/model.py:118-131: SE gate on the input, 1x1 conv, BN, skip)
    on the streaming kernel with the weights x gate in registers (csrc/pwn.hip, route CCVPE_ROUTE_PWN) behind the unchanged
    ccvpe_conv_igemm_bf16: against the fp32 oracle on bf16-rounded inputs, and against the generic kernel (switch off).  A wave's run
    of tiles crosses sample boundaries (B = 17: per-sample gates are re-folded), the last tile is ragged (160 x 160 x 3 pixels)."""
    from ccvpe_amd import _lib
    lib = _lib.load()
    w_ = hw if hw != 160 else 137                      # 160 x 137 x 3 = 65 760 pixels: not a multiple of 16 per sample -> ungated case only
    gated = (hw * w_) % 16 == 0
    x = r(synth.normal((b, cin, hw, w_), 900 + cin))
    wt = r(synth.normal((cout, cin, 1, 1), 901 + cout, (1.0 / cin) ** 0.5))
    sc, sh = synth.uniform((cout,), 902, 0.5, 1.5), synth.normal((cout,), 903, 0.1)
    gate = synth.uniform((b, cin), 904, 0.1, 1.0) if gated else None
    res = r(synth.normal((b, cout, hw, w_), 905)) if with_res else None
    xin = x * gate.view(b, cin, 1, 1) if gated else x
    want = F.conv2d(xin, wt) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if with_res:
        want = want + res
    f32 = torch.float32
    kw = dict(batch=b, in_h=hw, in_w=w_, scale=dev(sc, f32), shift=dev(sh, f32), gate=dev(gate, f32) if gated else None,
              residual=dev(nhwc(res)) if with_res else None)
    xd, wd = dev(nhwc(x)), dev(pack(wt))
    assert ops.conv_igemm(xd, cin, wd, cout, route_only=True, **kw)[0] == "pwn"
    got = ops.conv_igemm(xd, cin, wd, cout, **kw)
    assert got.dtype == BF
    close(nchw(got), want, 1e-2, "narrow projection %d->%d" % (cin, cout))
    prev = lib.ccvpe_set_pwn_kernels(0)
    try:
        assert ops.conv_igemm(xd, cin, wd, cout, route_only=True, **kw)[0] == "igemm"
        ref = ops.conv_igemm(xd, cin, wd, cout, **kw)
    finally:
        lib.ccvpe_set_pwn_kernels(prev)
    close(got, ref.float().cpu(), 8e-3, "streaming kernel vs generic kernel (the gate is folded into W instead of x: one bf16 rounding elsewhere)")


def test_igemm_bf16_gate_residual(ops):
    b, h, w, cin, cout = 3, 7, 10, 96, 24
    x = r(synth.normal((b, cin, h, w), 1))
    gate = synth.uniform((b, cin), 2)
    res = r(synth.normal((b, cout, h, w), 3))
    wt = r(synth.normal((cout, cin, 1, 1), 4, 0.1))
    xg = r(x * gate.view(b, cin, 1, 1))                    # the kernel re-rounds the gated operand to bf16
    want = F.conv2d(xg, wt) + res
    got = ops.conv_igemm(dev(nhwc(x)), cin, dev(pack(wt)), cout, batch=b, in_h=h, in_w=w, gate=dev(gate, torch.float32),
                         residual=dev(nhwc(res)))
    close(nchw(got), want, 1e-2, "bf16 gate+residual")


@pytest.mark.parametrize("c0,c1,cout,hw", [(40, 16, 40, 12), (80, 24, 80, 9), (16, 0, 16, 17), (320, 112, 320, 6),
                                           (1024, 320, 640, 4), (160, 40, 160, 20)])
def test_conv3x3_bf16_two_sources(ops, c0, c1, cout, hw):
    b = 2
    a = r(synth.normal((b, c0, hw, hw), 10 + c0))
    s = r(synth.normal((b, c1, hw, hw), 11 + c1)) if c1 else None
    wt = r(synth.normal((cout, c0 + c1, 3, 3), 12, (1.0 / (9 * (c0 + c1))) ** 0.5))
    bias = synth.normal((cout,), 13, 0.1)
    xin = torch.cat([a, s], 1) if c1 else a
    want = F.relu(F.conv2d(xin, wt, bias, padding=1))
    got = ops.conv_igemm(dev(nhwc(a)), c0, dev(pack(wt)), cout, batch=b, in_h=hw, in_w=hw, kh=3, kw=3, pad=1,
                         src1=dev(nhwc(s)) if c1 else None, c1=c1, shift=dev(bias, torch.float32), act=ops.ACT_RELU)
    close(nchw(got), want, 1e-2, "bf16 3x3 %d+%d->%d" % (c0, c1, cout))


@pytest.mark.parametrize("c,cout,hw,b,act,f32out", [(40, 40, 256, 2, True, False), (40, 40, 256, 3, False, True),
                                                    (32, 32, 256, 2, True, False), (64, 64, 128, 8, True, False),
                                                    (40, 40, 272, 2, True, False), (80, 80, 128, 5, True, False)])
def test_narrow_conv3x3_weights_in_registers(ops, c, cout, hw, b, act, f32out):
    """c3n_kernel (csrc/narrow_impl.h): the narrow last-level 3x3 layers with the weights resident in registers, flat (tap, octet)
    K, persistent workgroups and LDS-DMA halo tiles.  Enough tiles that the dispatcher picks it (route checked), several tiles
    per workgroup, image borders on every side, odd batch, an image of 17 x 17 tiles, fp32 output; also compared with conv3x3_kernel's
    result class (same operands, other summation order) through the A/B switch."""
    from ccvpe_amd import _lib
    lib = _lib.load()
    a = r(synth.normal((b, c, hw, hw), 10 + c))
    wt = r(synth.normal((cout, c, 3, 3), 12, (1.0 / (9 * c)) ** 0.5))
    bias = synth.normal((cout,), 13, 0.1)
    want = F.conv2d(a, wt, bias, padding=1)
    want = F.relu(want) if act else want
    kw = dict(batch=b, in_h=hw, in_w=hw, kh=3, kw=3, pad=1, shift=dev(bias, torch.float32),
              act=ops.ACT_RELU if act else ops.ACT_NONE, out_f32=f32out)
    x, wp = dev(nhwc(a)), dev(pack(wt))
    assert ops.conv_igemm(x, c, wp, cout, route_only=True, **kw)[0] == "c3n"
    got = ops.conv_igemm(x, c, wp, cout, **kw)
    assert got.dtype == (torch.float32 if f32out else BF)
    close(nchw(got), want, 1e-2 if not f32out else 2e-3, "narrow 3x3 %d->%d" % (c, cout))
    prev = lib.ccvpe_set_narrow_kernels(0)
    try:
        assert ops.conv_igemm(x, c, wp, cout, route_only=True, **kw)[0] == "conv3x3"
        ref = ops.conv_igemm(x, c, wp, cout, **kw)
    finally:
        lib.ccvpe_set_narrow_kernels(prev)
    close(nchw(got), nchw(ref).float().cpu(), 1e-2 if not f32out else 1e-5, "narrow vs tiled 3x3")


@pytest.mark.parametrize("c,hw,b,L,shift,stride,f32out", [(40, 256, 2, 40, 0, 2, True), (40, 256, 3, 40, 0, 2, False),
                                                          (40, 256, 2, 20, -3, 2, True), (40, 256, 2, 33, 7, 4, False)])
def test_narrow_conv3x3_with_the_next_levels_matching_in_its_epilogue(ops, c, hw, b, L, shift, stride, f32out):
    """ccvpe_conv3x3_match1_bf16 == ccvpe_conv_igemm_bf16 (convK.2) followed by ccvpe_match_level with one rotation hypothesis
    (models.py:211-228 with a single shift): scores and decoder-input rows, full and partial windows, a rolled window, fp32 and
    bf16 rows; and against the oracle's rotational_matching on the CPU convolution."""
    a = r(synth.normal((b, c, hw, hw), 20 + c))
    wt = r(synth.normal((c, c, 3, 3), 22, (1.0 / (9 * c)) ** 0.5))
    bias = synth.normal((c,), 23, 0.1)
    g = synth.normal((b, L + 8), 24)
    ldo = (c + 1 + 7) // 8 * 8
    x_ref = F.conv2d(a, wt, bias, padding=1)
    sc_ref = O.rotational_matching(x_ref, g[:, :L], [shift], stride)                       # [B,1,H,W]
    xd, wp, bd, gd = dev(nhwc(a)), dev(pack(wt)), dev(bias, torch.float32), dev(g, torch.float32)
    kw = dict(batch=b, in_h=hw, in_w=hw, bias=bd, out_f32=f32out)
    assert ops.conv3x3_match1(xd, c, wp, c, None, L, shift, stride, ldo, query_only=True, **kw)
    sc, cat = ops.conv3x3_match1(xd, c, wp, c, gd[:, :L], L, shift, stride, ldo, **kw)
    assert tuple(sc.shape) == (b, 1, hw, hw) and tuple(cat.shape) == (b, hw, hw, ldo) and cat.dtype == (torch.float32 if f32out else BF)
    assert float((sc.cpu() - sc_ref).abs().max()) <= 2e-5 + 2e-6 * c
    xn = x_ref / x_ref.norm(dim=1, keepdim=True).clamp_min(1e-12)
    tol = 1e-5 if f32out else 1e-2
    close(nchw(cat[..., :c]), xn, tol, "fused: normalised x")
    assert float((cat[..., c].float().cpu() - sc_ref[:, 0]).abs().max()) <= (2e-5 + 2e-6 * c if f32out else 1e-2)
    assert float(cat[..., c + 1:].float().abs().max()) == 0.0
    # the two separate launches on the same operands
    x2 = ops.conv_igemm(xd, c, wp, c, batch=b, in_h=hw, in_w=hw, kh=3, kw=3, pad=1, shift=bd, out_f32=f32out)
    sc2, cat2 = ops.match_level(x2, gd[:, :L], L, [shift], 1, 0, stride, ldo, channels=c)
    assert float((sc - sc2).abs().max()) <= (1e-5 if f32out else 1e-2)                   # (bf16 rows: x is rounded before the matching there)
    close(cat.float().cpu(), cat2.float().cpu(), 1e-5 if f32out else 1.2e-2, "fused vs conv + match_level")


def test_igemm_bf16_2x2s2_and_deconv(ops):
    from ccvpe_amd.models import _pack_deconv
    b, c, n = 2, 64, 48
    vol = r(synth.normal((b, c, 8, 8), 20))
    wl = r(synth.normal((n, c * 4), 21, 0.05))
    bias = synth.normal((n,), 22, 0.1)
    want = F.conv2d(vol, wl.view(n, c, 2, 2), bias, stride=2)
    got = ops.conv_igemm(dev(nhwc(vol)), c, dev(pack(wl.view(n, c, 2, 2))), n, batch=b, in_h=8, in_w=8, kh=2, kw=2,
                         stride=2, shift=dev(bias, torch.float32))
    close(nchw(got), want, 1e-2, "bf16 2x2s2")
    cin, cout, hw = 168, 40, 5
    x = r(synth.normal((b, cin, hw, hw), 30))
    wt = r(synth.normal((cin, cout, 2, 2), 31, (1.0 / cin) ** 0.5))
    bias = synth.normal((cout,), 32, 0.1)
    want = F.conv_transpose2d(x, wt, bias, stride=2)
    wp, b4 = _pack_deconv(wt, bias, [(0, 0, cin)], cin, BF)
    got = ops.conv_igemm(dev(nhwc(x)), cin, dev(wp), 4 * cout, batch=b, in_h=hw, in_w=hw, shift=dev(b4, torch.float32),
                         out_mode=ops.OUT_DECONV2X)
    close(nchw(got), want, 1e-2, "bf16 deconv")


@pytest.mark.parametrize("k,s,c,h,w,circ", [(3, 1, 32, 9, 12, False), (5, 2, 144, 8, 12, True), (5, 1, 1152, 5, 9, True),
                                            (3, 2, 240, 7, 9, False),
                                            (3, 1, 32, 128, 130, False), (3, 1, 32, 96, 200, True)])      # block 0's geometry (RB > 1)
def test_dwconv_bf16(ops, k, s, c, h, w, circ):
    b = 2
    x = r(synth.normal((b, c, h, w), 40 + c))
    wt = synth.normal((c, 1, k, k), 41, 1.0 / k)
    sc = synth.uniform((c,), 42, 0.5, 1.5)
    sh = synth.normal((c,), 43, 0.1)
    want = O.swish(O.same_conv(x, wt, k, s, 224, circ, groups=c) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    got, part = ops.dwconv(dev(nhwc(x)), dev(wt.reshape(c, k, k).permute(1, 2, 0), torch.float32),
                           dev(sc, torch.float32), dev(sh, torch.float32), k, s, circ)
    close(nchw(got), want, 1e-2, "bf16 dwconv")
    close(part.sum(1), want.sum(dim=(2, 3)), 1e-2, "squeeze partials (fp32 sums of unrounded outputs)")


@pytest.mark.parametrize("k,s,cin,h,w,circ", [(3, 2, 16, 40, 72, False), (3, 1, 24, 36, 40, True), (5, 2, 24, 34, 48, True),
                                              (5, 1, 40, 20, 24, False)])
def test_mbconv_front_bf16(ops, k, s, cin, h, w, circ):
    b, mid = 2, 6 * cin
    x = r(synth.normal((b, cin, h, w), 300 + cin))
    w_exp = r(synth.normal((mid, cin, 1, 1), 301, (2.0 / cin) ** 0.5))
    s0, b0 = synth.uniform((mid,), 302, 0.5, 1.5), synth.normal((mid,), 303, 0.2)
    w_dw = synth.normal((mid, 1, k, k), 304, 1.0 / k)
    s1, b1 = synth.uniform((mid,), 305, 0.5, 1.5), synth.normal((mid,), 306, 0.2)
    t = O.swish(F.conv2d(x, w_exp) * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1))       # stays fp32 in LDS
    want = O.swish(O.same_conv(t, w_dw, k, s, 224, circ, groups=mid) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    f32 = torch.float32
    got, part = ops.mbconv_front(dev(nhwc(x)), dev(pack(w_exp)), dev(s0, f32), dev(b0, f32),
                                 dev(w_dw.reshape(mid, k, k).permute(1, 2, 0), f32), dev(s1, f32), dev(b1, f32), mid, k, s, circ)
    assert got.dtype == BF
    close(nchw(got), want, 1e-2, "bf16 fused front")
    close(part.sum(1), want.sum(dim=(2, 3)), 1e-2, "squeeze partials")


@pytest.mark.parametrize("C,L,stride,hw,shifts,n_max,n_tail", [
    (1280, 1280, 64, 8, list(range(20)), 20, 20),
    (1280, 640, 64, 8, list(range(-10, 11)) + list(range(20)), 21, 20),
    (40, 40, 2, 20, list(range(20)), 20, 0),
    (2048, 512, 128, 8, list(range(16)), 16, 16),
    (160, 160, 8, 23, [0], 1, 0),
])
def test_match_level_bf16(ops, C, L, stride, hw, shifts, n_max, n_tail):
    b = 2
    x = r(synth.normal((b, C, hw, hw), 70 + C))
    g = synth.normal((b, L), 71 + L)
    ldo = (C + 1 + n_tail + 7) // 8 * 8
    sc, cat = ops.match_level(dev(nhwc(x)), dev(g, torch.float32), L, shifts, n_max, n_tail, stride, ldo)
    want = O.rotational_matching(x, g, shifts, stride)
    assert sc.dtype == torch.float32 and cat.dtype == BF
    close(sc, want, 2e-5 * 50, "scores (fp32 math on bf16 inputs)")
    catc = nchw(cat).float().cpu()
    close(catc[:, :C], F.normalize(x, p=2, dim=1), 1e-2, "normalised features")
    close(catc[:, C], want[:, :n_max].max(dim=1)[0], 1e-2, "max over rotations")


def test_stem_and_head_bf16(ops):
    x = synth.normal((2, 3, 32, 48), 4242)
    w = synth.normal((32, 3, 3, 3), 5, 0.3)
    sc, sh = synth.uniform((32,), 6, 0.5, 1.5), synth.normal((32,), 7, 0.1)
    want = O.swish(O.same_conv(x, w, 3, 2, 224, True) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    f32 = torch.float32
    got = ops.stem_conv(dev(x, f32), dev(w.permute(2, 3, 1, 0), f32), dev(sc, f32), dev(sh, f32), True, out_dtype=BF)
    assert got.dtype == BF
    close(nchw(got), want, 1e-2, "bf16 stem")
    y = r(synth.normal((2, 16, 21, 21), 80))
    wt = synth.normal((2, 16, 3, 3), 81, 0.1)
    bias = synth.normal((2,), 82, 0.1)
    want = F.normalize(F.conv2d(y, wt, bias, padding=1), p=2, dim=1)
    got = ops.head_conv3x3(dev(nhwc(y)), dev(wt.permute(0, 2, 3, 1), f32), dev(bias, f32), 2, True)
    close(got, want, 1e-4, "head conv (bf16 in, fp32 math/out)")


@pytest.mark.parametrize("case", [
    dict(kind="vigor", ori_noise=None, circular=True, wseed=0, grd="vigor"),                 # C2: N_rot = 20
    dict(kind="vigor", ori_noise=180, circular=False, wseed=0, grd="vigor_fov180"),          # C4: FoV 180
    dict(kind="kitti", ori_noise=None, circular=False, wseed=1, grd="kitti"),
])
def test_forward_bf16_vs_oracle(case, synth_sd, oracle_forward):
    from test_forward_gpu import build
    net = build(case, synth_sd).set_precision("bf16")
    grd, sat, ref = oracle_forward(case, 2, 991)       # (the fp32 test of the same case uses the same inputs and oracle run)
    out = net(grd.cuda(), sat.cuda())
    torch.cuda.synchronize()
    assert [tuple(t.shape) for t in out] == [tuple(t.shape) for t in ref]
    assert all(t.dtype == torch.float32 for t in out)
    lg, rl = out[0].cpu(), ref[0]
    rng = (rl.max() - rl.min()).item()
    err = (lg - rl).abs().max().item()
    assert err < 1e-2 * rng, "logits err %.3e vs range %.3e (measured: ~4e-3 with the default fp32 tail)" % (err, rng)
    # north_star: heat-map arg-max identical to the reference — with the fp32 tail (set_precision default) the bf16 path
    # picks the oracle's pixel on these cases (rate over 64 seeded pairs: test_bf16_argmax_match_rate)
    assert torch.equal(lg.argmax(1), rl.argmax(1)), "bf16 arg-max pixel differs from the fp32 oracle's"
    for a, bb in zip(out[3:], ref[3:]):
        assert (a.cpu() - bb).abs().max().item() < 2e-2
    assert abs(out[1].sum().item() - 2.0) < 1e-3
    # the orientation at the arg-max pixel (what the evaluation reads, train_VIGOR.py:310-324): vector within bf16
    # resolution, and the 18-degree orientation BIN equal to the oracle's (north_star: "orientation bin exact") wherever
    # the oracle's angle is further from a bin edge than the angle the vector bound allows (asin(3e-2) = 1.7 deg -> 2 deg)
    for b in range(2):
        i = int(rl[b].argmax())
        o_got, o_ref = out[2][b].reshape(2, -1)[:, i].cpu(), ref[2][b].reshape(2, -1)[:, i]
        assert (o_got - o_ref).abs().max().item() < 3e-2
        same, near_edge = ori_bin_check(o_got, o_ref)
        assert same or near_edge, "bf16 orientation bin differs away from a bin edge"
    # fp32 precision is restored by switching back (packed weights are re-derived)
    net.set_precision("fp32")
    out32 = net(grd.cuda(), sat.cuda())
    assert ((out32[0].cpu() - rl).abs().max() / rl.abs().max()).item() < 1e-3


LOGIT_ERR_BOUND = 6e-3        # bf16 storage path (default fp32 tail): max |logit error| / logit range, asserted below (measured 4.5e-3)
ORI_EDGE_DEG = 2.0            # asin(3e-2): the angle the asserted orientation-vector bound can move (the B = 2 cases above)
# The (cos, sin) error over hundreds of samples is heavy-tailed, and round 6 pinned down why (tools/ori_norm_scan.py over these 256
# pairs): F.normalize (models.py:341) divides by the norm of conv1_ori's raw 2-vector, whose median is 0.37 but which drops to 0.012 at
# one sample's arg-max pixel — exactly the sample with the 1.05e-1 unit-vector error; error x |raw| never exceeds 7.5e-3.  So the
# invariant is a bound on the error of the RAW vector, and it holds for EVERY sample (no excused outliers):
#     |unit-vector error| <= ORI_RAW_ERR_BOUND / |raw|          (and therefore <= ORI_VEC_BOUND_AT_SCALE wherever |raw| >= 1/6)
# the orientation bin must be equal unless the fp32 angle is within the angle that bound allows, asin(ORI_RAW_ERR_BOUND / |raw|) + 0.5
# degrees, of a bin edge; the samples with |raw| < ORI_SMALL_NORM (where that angle exceeds 5.7 degrees) are counted and reported.
ORI_RAW_ERR_BOUND = 1e-2       # measured 7.5e-3 (max over 256 + 64 pairs of |bf16 unit vector - fp32 unit vector| x |fp32 raw vector|)
ORI_VEC_BOUND_AT_SCALE = 6e-2  # what the raw bound implies for |raw| >= 1/6 (asin(6e-2) = 3.4 deg)
ORI_SMALL_NORM = 0.1


def ori_edge_deg(raw_norm):
    """The angle (degrees) the raw-vector error bound can turn a vector of this norm, + 0.5 degrees of slack."""
    import math
    return math.degrees(math.asin(min(1.0, ORI_RAW_ERR_BOUND / max(raw_norm, 1e-30)))) + 0.5


def ori_bin_check(o_got, o_ref, edge_deg=ORI_EDGE_DEG):
    """(bins equal, oracle angle within edge_deg of an 18-degree bin edge) for two (cos, sin) vectors."""
    ang = lambda v: float(torch.atan2(v[1], v[0]) * 180 / 3.14159265) % 360
    a_g, a_r = ang(o_got), ang(o_ref)
    d = a_r % 18.0
    return int(a_g // 18) == int(a_r // 18), min(d, 18.0 - d) < edge_deg


def _centre_tap_head(sd):
    """A weight set with PEAKED heat-maps: conv1.0 / conv1.2 as centre-tap filters (x3), so the 512 x 512 logits are a pointwise
    function of the deconv output instead of a 5 x 5-smoothed one — the top-1 / top-2 margin then clears twice the bf16 error
    bound on ~3 of 4 samples (tools/argmax_margins.py: median margin 2.2e-2 of the range against 1.4e-2 for the default set)."""
    sd = {k: v.clone() for k, v in sd.items()}
    for k in ("conv1.0.weight", "conv1.2.weight"):
        w = sd[k]
        c = w[:, :, 1, 1].clone() * 3.0
        w.zero_()
        w[:, :, 1, 1] = c
    return sd


def test_bf16_argmax_margin_rule(synth_sd):
    """Arg-max pixel + orientation bin of the bf16 storage path (default fp32 tail) against the fp32 HIP path — itself arg-max
    and bin exact against the reference's goldens (tests/test_forward_gpu.py) — over 256 seeded pairs of CVM_VIGOR (N_rot = 20)
    with the default synthetic weights plus 64 pairs with a PEAKED-heat-map weight set (_centre_tap_head).

    The rule (plus an absolute floor on the match count): (1) the logit error is bounded: |bf16 - fp32| <= LOGIT_ERR_BOUND x range for every
    pixel of every sample; (2) wherever the fp32 top-1 / top-2 margin exceeds TWICE that bound the arg-max pixel is EQUAL — and
    that case must actually occur: >= 25 % of the default-weight samples and >= 60 % of the peaked ones (measured 53 % / 73 %);
    (3) the samples inside the bound ("near ties") are counted and reported, and if the bf16 arg-max moves there it moves to a
    pixel whose fp32 logit is within twice the bound of the maximum — never anywhere else; (4) for EVERY sample the (cos, sin) vector at the
    fp32 arg-max pixel is within ORI_RAW_ERR_BOUND / |raw| of the fp32 one (|raw| = norm of the fp32 path's un-normalised
    conv1_ori output there: the test runs the fp32 model with its raw-output hook and normalises in torch), and the orientation
    bin is equal wherever the fp32 angle is further from a bin edge than the angle that bound can move; small-norm samples are
    counted, not excused.  Pure bf16 storage
    (fp32_tail_levels = 0) is held to the same rule with its own measured bound (first 32 pairs).
    Three model instances (fp32, bf16, pure bf16) share the weights: no re-pack per chunk."""
    from ccvpe_amd import models

    def build(sd, precision, tail=1):
        net = models.CVM_VIGOR("cuda", True)
        net.load_state_dict(sd, strict=True)
        return net.to("cuda:0").eval().set_precision(precision, fp32_tail_levels=tail) if precision == "bf16" else net.to("cuda:0").eval()

    def fresh():
        return dict(n=0, same=0, near=0, moved_far=0, worst=0.0, bins_bad=0, bins_edge=0, ori_worst=0.0, ori_errs=[], raw_worst=0.0, small=0)
    stats = {"tail": fresh(), "pure": fresh(), "peaked": fresh()}
    bound = {"tail": LOGIT_ERR_BOUND, "pure": 8e-3, "peaked": 7e-3}

    def account(st, E, ref, got, ref_raw, got_ori):
        ref_ori = F.normalize(ref_raw, p=2, dim=1)                     # models.py:341 on the fp32 path's raw output
        rng = ref.max(1)[0] - ref.min(1)[0]
        err = (got - ref).abs().max(1)[0] / rng
        top2 = ref.topk(2, dim=1)[0]
        margin = (top2[:, 0] - top2[:, 1]) / rng
        ia, ib = ref.argmax(1), got.argmax(1)
        for b in range(ref.shape[0]):
            st["n"] += 1
            st["worst"] = max(st["worst"], float(err[b]))
            near = float(margin[b]) <= 2 * E
            st["near"] += int(near)
            if int(ia[b]) == int(ib[b]):
                st["same"] += 1
            else:
                assert near, "arg-max moved although the fp32 margin %.2e exceeds twice the error bound" % float(margin[b])
                drop = float((ref[b, ia[b]] - ref[b, ib[b]]) / rng[b])
                st["moved_far"] += int(drop > 2 * E)
            o_g = got_ori[b].reshape(2, -1)[:, ia[b]].cpu()
            o_r = ref_ori[b].reshape(2, -1)[:, ia[b]].cpu()
            raw_norm = float(ref_raw[b].reshape(2, -1)[:, ia[b]].norm())
            same, edge = ori_bin_check(o_g, o_r, ori_edge_deg(raw_norm))
            oerr = float((o_g - o_r).abs().max())
            st["ori_worst"] = max(st["ori_worst"], oerr)
            st["raw_worst"] = max(st["raw_worst"], oerr * raw_norm)
            st["ori_errs"].append(oerr)
            st["small"] += int(raw_norm < ORI_SMALL_NORM)
            st["bins_edge"] += int(edge)
            st["bins_bad"] += int(not same and not edge)
            assert oerr * raw_norm <= ORI_RAW_ERR_BOUND, "orientation: unit-vector error %.3e at |raw| = %.3e" % (oerr, raw_norm)

    sd = synth_sd("vigor", 0)
    n32, nbf, npure = build(sd, "fp32"), build(sd, "bf16"), build(sd, "bf16", 0)
    n32.ori_raw_output = True                    # test hook: conv1_ori's output before F.normalize (models.py:341)
    for c0 in range(0, 256, 16):
        grd, sat = synth.synthetic_pair(16, "vigor", 5000 + c0, device="cuda")
        r = n32(grd, sat)
        ref, ref_ori = r[0].clone(), r[2].clone()
        g = nbf(grd, sat)
        account(stats["tail"], bound["tail"], ref, g[0], ref_ori, g[2])
        if c0 < 32:
            g = npure(grd, sat)
            account(stats["pure"], bound["pure"], ref, g[0], ref_ori, g[2])
    del n32, nbf, npure
    sdp = _centre_tap_head(sd)
    n32, nbf = build(sdp, "fp32"), build(sdp, "bf16")
    n32.ori_raw_output = True
    for c0 in range(0, 64, 16):
        grd, sat = synth.synthetic_pair(16, "vigor", 9000 + c0, device="cuda")
        r = n32(grd, sat)
        ref, ref_ori = r[0].clone(), r[2].clone()
        g = nbf(grd, sat)
        account(stats["peaked"], bound["peaked"], ref, g[0], ref_ori, g[2])
    for k, st in stats.items():
        print("bf16 (%s): arg-max equal %d/%d, near ties (margin <= 2 x %.0e of range) %d, worst logit error %.2e of range, "
              "orientation: vector error p50 %.2e p99 %.2e, largest %s, worst error x |raw| %.2e, %d samples with |raw| < %.2f, bins: %d at a "
              "bin edge, %d wrong"
              % (k, st["same"], st["n"], bound[k], st["near"], st["worst"], sorted(st["ori_errs"])[len(st["ori_errs"]) // 2],
                 sorted(st["ori_errs"])[int(0.99 * (len(st["ori_errs"]) - 1))], ["%.2e" % e for e in sorted(st["ori_errs"])[-4:]],
                 st["raw_worst"], st["small"], ORI_SMALL_NORM, st["bins_edge"], st["bins_bad"]))
        assert st["worst"] <= bound[k], "logit error bound exceeded"
        assert st["raw_worst"] <= ORI_RAW_ERR_BOUND, "orientation: raw-vector error bound exceeded"
        assert st["moved_far"] == 0, "the arg-max moved to a pixel outside the error bound"
        # an absolute floor next to the margin rule: however many near ties the synthetic weights produce, the arg-max must not
        # move on more than 6 % of the samples with the fp32 tail / 6 % in pure bf16 (round 4, 64 pairs: 64/64 and 63/64)
        floor = {"tail": st["n"] - 16, "pure": st["n"] - 2, "peaked": st["n"] - 3}[k]
        assert st["same"] >= floor, "bf16 arg-max equal on %d of %d samples (floor %d)" % (st["same"], st["n"], floor)
        assert st["bins_bad"] == 0, "orientation bin differs away from a bin edge"
    # "exact" is actually exercised: clear-margin samples exist in number (on them the arg-max was asserted EQUAL above)
    assert stats["tail"]["n"] - stats["tail"]["near"] >= 64, stats["tail"]
    assert stats["peaked"]["n"] - stats["peaked"]["near"] >= 38, stats["peaked"]
