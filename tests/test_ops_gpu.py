"""Per-operator parity on the MI355X: every call goes through the C ABI (ccvpe_amd.ops ->
libccvpe_hip.so) and is compared with the CPU oracle / a torch fp32 CPU restatement of the same
op on identical seeded inputs.  fp32 everywhere; tolerances are round-off class
(1e-4 relative to the tensor's scale unless noted; north_star allows 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

import golden_util as G
from ccvpe_amd import synth
from oracle import ccvpe_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from ccvpe_amd import ops as _ops, _lib
    _lib.load()
    return _ops


def dev(t):
    return t.cuda().contiguous()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(got, want, tol=1e-4, what=""):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e)" % (what, err, scale, err / scale)


def pack_conv(w):
    from ccvpe_amd.models import _pack_conv
    return _pack_conv(w)


# ------------------------------------------------------------------------------------------
# implicit GEMM
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout", [(16, 96), (24, 144), (40, 240), (80, 480), (112, 672), (192, 1152),
                                      (96, 24), (144, 40), (480, 112), (320, 1280), (32, 16), (672, 192),
                                      (1152, 320), (240, 80), (64, 64), (8, 160)])
def test_igemm_1x1_all_tile_configs(ops, cin, cout):
    b, h, w = 2, 9, 13                                   # M = 234: exercises the M tail
    x = synth.normal((b, cin, h, w), 100 + cin)
    wt = synth.normal((cout, cin, 1, 1), 200 + cout, (1.0 / cin) ** 0.5)
    sc = synth.uniform((cout,), 300, 0.5, 1.5)
    sh = synth.normal((cout,), 301, 0.1)
    want = O.swish(F.conv2d(x, wt) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    got = ops.conv_igemm(dev(nhwc(x)), cin, dev(pack_conv(wt)), cout, batch=b, in_h=h, in_w=w,
                         scale=dev(sc), shift=dev(sh), act=ops.ACT_SWISH)
    close(nchw(got), want, 1e-4, "1x1 %d->%d" % (cin, cout))


@pytest.mark.parametrize("cin,cout,ldd", [(112, 126, 128), (1280, 126, 128), (200, 36, 40), (16, 24, 24)])
def test_generic_gemm_ragged_n_k_tail_many_tiles(ops, cin, cout, ldd):
    """GENERIC igemm kernel (ReLU keeps a 1x1 conv away from pw_gemm_kernel, as in the model's fused ground-descriptor
    conv: N = 126, ldd = 128): several M tiles + M tail, ragged N written into wider rows, residual rows."""
    b, h, w = 3, 19, 23                                  # M = 1311
    x = synth.normal((b, cin, h, w), 410 + cin)
    wt = synth.normal((cout, cin, 1, 1), 420 + cout, (1.0 / cin) ** 0.5)
    sh = synth.normal((cout,), 430, 0.1)
    res = synth.normal((b, cout, h, w), 431)
    resp = torch.zeros((b, h, w, ldd))
    resp[..., :cout] = nhwc(res)
    want = torch.relu(F.conv2d(x, wt) + sh.view(1, -1, 1, 1)) + res
    dst = torch.full((b, h, w, ldd), -7.0, device="cuda")
    kw = dict(batch=b, in_h=h, in_w=w, shift=dev(sh), act=ops.ACT_RELU, residual=dev(resp), dst=dst, ldd=ldd)
    assert ops.conv_igemm(dev(nhwc(x)), cin, dev(pack_conv(wt)), cout, route_only=True, **kw)[0] == "igemm"
    got = ops.conv_igemm(dev(nhwc(x)), cin, dev(pack_conv(wt)), cout, **kw)
    close(nchw(got[..., :cout]), want, 1e-4, "generic ragged %d->%d" % (cin, cout))
    if ldd > cout:
        assert float((got[..., cout:] + 7.0).abs().max()) == 0.0, "columns beyond N were written"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("act", ["none", "swish"])
@pytest.mark.parametrize("cin,cout,ldd,with_res", [(112, 126, 128, True), (1280, 126, 128, False), (200, 70, 72, True),
                                                    (72, 50, 56, True), (96, 101, 104, False), (40, 242, 248, True)])
def test_pw_gemm_ragged_n_k_tail_many_tiles(ops, cin, cout, ldd, with_res, act, dtype):
    """pw_gemm_kernel proper (the route is ASSERTED: 1x1, stride 1, one source, act NONE / SWISH, N > 48): N % 4 != 0 and
    N % 8 != 0 (the `full == false` element tail of its 16-byte residual / store path, fp32 and bf16), K not a multiple of
    the 128-byte stage, several M tiles + an M tail, output and residual rows wider than N (ldd > N)."""
    b, h, w = 3, 19, 23                                  # M = 1311
    bf = dtype == "bf16"
    tdt = torch.bfloat16 if bf else torch.float32
    x = synth.normal((b, cin, h, w), 510 + cin)
    wt = synth.normal((cout, cin, 1, 1), 520 + cout, (1.0 / cin) ** 0.5)
    sc = synth.uniform((cout,), 529, 0.5, 1.5)
    sh = synth.normal((cout,), 530, 0.1)
    res = synth.normal((b, cout, h, w), 531)
    if bf:      # the oracle sees the values the kernel reads
        x, wt, res = x.bfloat16().float(), wt.bfloat16().float(), res.bfloat16().float()
    resp = torch.zeros((b, h, w, ldd))
    resp[..., :cout] = nhwc(res)
    want = F.conv2d(x, wt) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if act == "swish":
        want = O.swish(want)
    if with_res:
        want = want + res
    from ccvpe_amd.models import _pack_conv
    dst = torch.full((b, h, w, ldd), -7.0, device="cuda", dtype=tdt)
    kw = dict(batch=b, in_h=h, in_w=w, scale=dev(sc), shift=dev(sh), act=ops.ACT_SWISH if act == "swish" else ops.ACT_NONE,
              residual=dev(resp.to(tdt)) if with_res else None, dst=dst, ldd=ldd)
    xs, wp = dev(nhwc(x).to(tdt)), dev(_pack_conv(wt, tdt))
    route = ops.conv_igemm(xs, cin, wp, cout, route_only=True, **kw)
    assert route[0] == "pw_gemm", route
    got = ops.conv_igemm(xs, cin, wp, cout, **kw)
    close(nchw(got[..., :cout].float()), want, 1.2e-2 if bf else 1e-4, "pw ragged %d->%d %s %s" % (cin, cout, act, dtype))
    if ldd > cout:
        assert float((got[..., cout:].float() + 7.0).abs().max()) == 0.0, "columns beyond N were written"


@pytest.mark.parametrize("cin,cout,hw,gated,with_res,act,b", [
    (112, 672, 32, False, False, "swish", 2),      # expand of blocks 9-11 (128 x 112 tile, 4 ring stages), K = 3.5 stages
    (672, 112, 32, True, True, "none", 2),         # their project conv: SE gate on the A fragments + skip
    (80, 480, 32, False, False, "swish", 2),       # 128 x 160 tile: 3 ring stages, K = 2.5 stages
    (192, 1152, 16, False, False, "swish", 3),     # 128 x 128 tile, hw = 256: two tiles per sample
    (1152, 192, 16, True, True, "none", 2),        # 128 x 96 tile, 36 stages, the longest gate vector
    (320, 1280, 16, False, False, "swish", 1),     # the head conv
    (672, 112, 32, True, False, "none", 64),       # full size: 512 tiles over 256 workgroups (two tiles each, ring across the tile edge)
    (112, 672, 32, False, False, "swish", 64),     # full size: 3 072 tiles, 12 per workgroup
    (112, 672, 32, False, False, "raw", 2),        # train mode / 1x1 input gradients: no BN vectors, plain store (fp32 only)
    (672, 112, 32, True, False, "raw", 2),         # train-mode project conv: SE gate, raw output
    (1152, 192, 16, False, False, "raw", 3),       # input gradient of a 192 -> 1152 expand conv
])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pw_ring_kernel_vs_oracle_and_pw_gemm(ops, cin, cout, hw, gated, with_res, act, b, dtype):
    """csrc/conv_pw2_impl.h (route ASSERTED): the pointwise GEMM with the LDS-DMA ring, fp32 and bf16 storage, against the oracle
    convolution (small batches) and BIT-IDENTICAL to pw_gemm_kernel, which ccvpe_set_pw_ring_kernels(0) brings back behind the same
    entry point.  bf16: K = 112 and 80 end in a HALF stage (16 of 32 channels)."""
    from ccvpe_amd.models import _pack_conv
    from ccvpe_amd import _lib
    big = b > 8
    bf = dtype == "bf16"
    raw = act == "raw"
    if raw and bf:
        pytest.skip("the raw forms are instantiated for fp32 (train mode runs in fp32)")
    tdt = torch.bfloat16 if bf else torch.float32
    gen = dict(device="cuda") if big else {}
    x = synth.normal((b, cin, hw, hw), 610 + cin, **gen)
    wt = synth.normal((cout, cin, 1, 1), 620 + cout, (1.0 / cin) ** 0.5)
    sc, sh = synth.uniform((cout,), 629, 0.5, 1.5), synth.normal((cout,), 630, 0.1)
    gate = synth.uniform((b, cin), 631, 0.1, 1.0) if gated else None
    res = synth.normal((b, cout, hw, hw), 632, **gen) if with_res else None
    if bf:      # the oracle sees the values the kernel reads
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
        res = res.bfloat16().float() if with_res else None
    kw = dict(batch=b, in_h=hw, in_w=hw, scale=None if raw else dev(sc), shift=None if raw else dev(sh),
              act=ops.ACT_SWISH if act == "swish" else ops.ACT_NONE,
              gate=dev(gate) if gated else None, residual=dev(nhwc(res)).to(tdt) if with_res else None)
    xs, wp = dev(nhwc(x)).to(tdt), dev(_pack_conv(wt, tdt))
    route = ops.conv_igemm(xs, cin, wp, cout, route_only=True, **kw)
    assert route[0] == "pw_ring", route
    got = ops.conv_igemm(xs, cin, wp, cout, **kw)
    lib = _lib.load()
    prev = lib.ccvpe_set_pw_ring_kernels(0)
    try:
        assert ops.conv_igemm(xs, cin, wp, cout, route_only=True, **kw)[0] == "pw_gemm"
        old = ops.conv_igemm(xs, cin, wp, cout, **kw)
    finally:
        lib.ccvpe_set_pw_ring_kernels(prev)
    assert torch.equal(got, old), "pw_ring differs from pw_gemm: max %.3e" % float((got.float() - old.float()).abs().max())
    if not big:
        xin = x * gate.view(b, cin, 1, 1) if gated else x
        if bf and gated:
            xin = xin.bfloat16().float()                 # the gated activations are rounded back to bf16 before the matrix product
        want = F.conv2d(xin, wt)
        if not raw:
            want = want * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        if act == "swish":
            want = O.swish(want)
        if with_res:
            want = want + res
        close(nchw(got).float(), want, 1.2e-2 if bf else 1e-4, "pw ring %d->%d %s" % (cin, cout, dtype))


def test_conv_route_matches_dispatch_rules(ops):
    """ccvpe_conv_igemm_route: 3x3 s1 p1 -> conv3x3; 1x1 N > 48 act NONE/SWISH -> pw_gemm; ReLU / narrow N / gather forms ->
    the generic kernel (csrc/conv_igemm.hip: conv_igemm_any)."""
    x = torch.zeros((1, 8, 8, 64), device="cuda")
    w1 = torch.zeros((96, 64), device="cuda")
    w3 = torch.zeros((96, 9 * 64), device="cuda")
    w2 = torch.zeros((96, 4 * 64), device="cuda")
    r = lambda **k: ops.conv_igemm(x, 64, k.pop("w", w1), k.pop("n", 96), batch=1, in_h=8, in_w=8, route_only=True, **k)
    assert r()[0] == "pw_gemm" and r(act=ops.ACT_SWISH)[0] == "pw_gemm"
    assert r(act=ops.ACT_RELU)[0] == "igemm"
    assert r(n=48, w=torch.zeros((48, 64), device="cuda"))[0] == "igemm"
    assert r(w=w3, kh=3, kw=3, pad=1)[0] == "conv3x3"
    assert r(w=w2, kh=2, kw=2, stride=2)[0] == "igemm"
    assert r(n=80, w=torch.zeros((80, 64), device="cuda")) == ("pw_gemm", 4, 3, 2)      # 256 x 80 tile re-routed to 128 x 96


def test_igemm_gate_and_residual(ops):
    b, h, w, cin, cout = 3, 7, 10, 96, 24
    x = synth.normal((b, cin, h, w), 1)
    gate = synth.uniform((b, cin), 2)
    res = synth.normal((b, cout, h, w), 3)
    wt = synth.normal((cout, cin, 1, 1), 4, 0.1)
    sc = synth.uniform((cout,), 5, 0.5, 1.5)
    sh = synth.normal((cout,), 6, 0.1)
    want = F.conv2d(x * gate.view(b, cin, 1, 1), wt) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res
    got = ops.conv_igemm(dev(nhwc(x)), cin, dev(pack_conv(wt)), cout, batch=b, in_h=h, in_w=w,
                         gate=dev(gate), scale=dev(sc), shift=dev(sh), residual=dev(nhwc(res)))
    close(nchw(got), want, 1e-4, "gate+residual")


@pytest.mark.parametrize("c0,c1,cout,hw", [(40, 16, 40, 12), (80, 24, 80, 9), (16, 0, 16, 17), (320, 112, 320, 6),
                                           (1024, 320, 640, 4)])
def test_igemm_3x3_two_sources(ops, c0, c1, cout, hw):
    b = 2
    a = synth.normal((b, c0, hw, hw), 10 + c0)
    s = synth.normal((b, c1, hw, hw), 11 + c1) if c1 else None
    wt = synth.normal((cout, c0 + c1, 3, 3), 12, (1.0 / (9 * (c0 + c1))) ** 0.5)
    bias = synth.normal((cout,), 13, 0.1)
    xin = torch.cat([a, s], 1) if c1 else a
    want = F.relu(F.conv2d(xin, wt, bias, padding=1))
    got = ops.conv_igemm(dev(nhwc(a)), c0, dev(pack_conv(wt)), cout, batch=b, in_h=hw, in_w=hw, kh=3, kw=3, pad=1,
                         src1=dev(nhwc(s)) if c1 else None, c1=c1, shift=dev(bias), act=ops.ACT_RELU)
    close(nchw(got), want, 1e-4, "3x3 %d+%d->%d" % (c0, c1, cout))


def test_igemm_2x2_stride2_linear_patches(ops):
    """models.py:173-184: Linear over flattened (c,h,w) 2x2 patches."""
    b, c, n = 2, 64, 48
    vol = synth.normal((b, c, 8, 8), 20)
    wl = synth.normal((n, c * 4), 21, 0.05)
    bias = synth.normal((n,), 22, 0.1)
    want = F.conv2d(vol, wl.view(n, c, 2, 2), bias, stride=2)
    got = ops.conv_igemm(dev(nhwc(vol)), c, dev(pack_conv(wl.view(n, c, 2, 2))), n, batch=b, in_h=8, in_w=8,
                         kh=2, kw=2, stride=2, shift=dev(bias))
    close(nchw(got), want, 1e-4, "2x2s2")


@pytest.mark.parametrize("cin,cout,hw", [(48, 16, 9), (168, 40, 5), (648, 320, 3)])
def test_igemm_deconv_pixel_shuffle(ops, cin, cout, hw):
    from ccvpe_amd.models import _pack_deconv
    b = 2
    x = synth.normal((b, cin, hw, hw), 30 + cin)
    wt = synth.normal((cin, cout, 2, 2), 31, (1.0 / cin) ** 0.5)
    bias = synth.normal((cout,), 32, 0.1)
    want = F.conv_transpose2d(x, wt, bias, stride=2)
    wp, b4 = _pack_deconv(wt, bias, [(0, 0, cin)], cin)
    got = ops.conv_igemm(dev(nhwc(x)), cin, dev(wp), 4 * cout, batch=b, in_h=hw, in_w=hw, shift=dev(b4),
                         out_mode=ops.OUT_DECONV2X)
    close(nchw(got), want, 1e-4, "deconv")


@pytest.mark.parametrize("cp,cref,cd,c1,co,h1,w1,bf16", [
    (48, 41, 16, 0, 16, 9, 11, False),        # level 1 (no skip)
    (88, 81, 40, 16, 40, 6, 7, False),        # level 2
    (168, 161, 80, 24, 80, 5, 5, False),
    (648, 641, 320, 112, 320, 3, 4, False),
    (1304, 1281, 1024, 320, 640, 2, 2, False),
    (64, 64, 32, 16, 32, 7, 6, False),        # ori branch: plain input
    (88, 81, 40, 16, 40, 6, 7, True),
    (648, 641, 320, 112, 320, 3, 4, True),
    # w1 >= 16: the low-res-halo kernel (upconv_halo_kernel; tiles + ragged borders in both directions, channel counts that
    # do not fill the last 64-byte chunk of either source, 1- and 2-column tiles, every tile family of the decoder)
    (48, 41, 16, 0, 16, 17, 33, False),       # level 1 (no skip), <4,1,1>
    (88, 81, 40, 16, 40, 18, 16, False),      # level 2, <4,3,1>
    (168, 161, 80, 24, 80, 9, 20, False),     # level 3, <4,5,1>
    (328, 321, 160, 40, 160, 7, 17, False),   # level 4, <4,5,2>
    (648, 641, 320, 112, 320, 5, 16, False),  # level 5, 2 x <4,5,2>
    (64, 64, 32, 16, 32, 19, 35, False),      # ori level 2, <4,1,2>
    (128, 128, 64, 24, 64, 8, 32, False),     # ori level 3, <4,2,2>
    (256, 256, 128, 40, 128, 6, 16, False),   # ori level 4, <4,4,2>
    (88, 81, 40, 16, 40, 18, 16, True),
    (168, 161, 80, 24, 80, 9, 20, True),
    (648, 641, 320, 112, 320, 5, 16, True),
    (64, 64, 32, 16, 32, 19, 35, True),
    (48, 41, 16, 0, 16, 17, 33, True),        # no skip, c0 % 32 != 0: the DMA kernel's W-row precondition fails -> halo kernel
    (328, 321, 160, 40, 160, 7, 17, True),    # level 4 in bf16 (DMA kernel, two K stages per barrier in phase B)
    (256, 256, 128, 40, 128, 6, 16, True),    # ori level 4 in bf16, <4,4,2>
    # up2_kernel (csrc/narrow_impl.h; bf16, >= 2 tiles per CU): level 2 of both decoders and KITTI's localisation level 2 — four
    # parities per workgroup, weights in registers, LDS-DMA halos, several tiles per persistent workgroup, every image border
    (88, 81, 40, 16, 40, 128, 256, True),
    (64, 64, 32, 16, 32, 128, 256, True),
    (136, 129, 32, 16, 32, 64, 512, True),
])
def test_upconv_folds_deconv_into_conv3x3(ops, cp, cref, cd, c1, co, h1, w1, bf16):
    """relu(conv3x3(cat[deconv2x2s2(x)+b, skip])+b)  ==  the folded per-parity GEMM (incl. borders)."""
    from ccvpe_amd.models import _pack_upconv
    b = 2
    dt = torch.bfloat16 if bf16 else torch.float32
    rnd = (lambda t: t.to(dt).float())
    x = rnd(synth.normal((b, cp, h1, w1), 600 + cp))
    x[:, cref:] = 0                                          # padding channels of the concat buffer
    skip = rnd(synth.normal((b, c1, 2 * h1, 2 * w1), 601)) if c1 else None
    wd = synth.normal((cref, cd, 2, 2), 602, (1.0 / cref) ** 0.5)
    bd = synth.normal((cd,), 603, 0.3)
    w3 = synth.normal((co, cd + c1, 3, 3), 604, (1.0 / (9 * (cd + c1))) ** 0.5)
    b3 = synth.normal((co,), 605, 0.1)
    d = F.conv_transpose2d(x[:, :cref], wd, bd, stride=2)
    want = F.relu(F.conv2d(torch.cat([d, skip], 1) if c1 else d, w3, b3, padding=1))
    fw, fshift = _pack_upconv(wd.cuda(), bd.cuda(), [(0, 0, cref)], cp, w3.cuda(), b3.cuda(), dt)
    xd, sk = nhwc(x).to(dt).cuda().contiguous(), (nhwc(skip).to(dt).cuda().contiguous() if c1 else None)
    got = ops.upconv3x3(xd, cp, fw, fshift, co, batch=b, h1=h1, w1=w1, src1=sk, c1=c1, act=ops.ACT_RELU)
    close(nchw(got).float(), want, 2e-2 if bf16 else 1e-4, "upconv cp=%d" % cp)
    if h1 * w1 >= 128 * 256:                                # the narrow-level kernel must be the one that ran, and agree with the tiled one
        from ccvpe_amd import _lib
        lib = _lib.load()
        assert ops.upconv3x3(xd, cp, fw, fshift, co, batch=b, h1=h1, w1=w1, src1=sk, c1=c1, act=ops.ACT_RELU, route_only=True)[0] == "up2_kernel"
        prev = lib.ccvpe_set_narrow_kernels(0)
        try:
            assert ops.upconv3x3(xd, cp, fw, fshift, co, batch=b, h1=h1, w1=w1, src1=sk, c1=c1, act=ops.ACT_RELU, route_only=True)[0] != "up2_kernel"
            ref = ops.upconv3x3(xd, cp, fw, fshift, co, batch=b, h1=h1, w1=w1, src1=sk, c1=c1, act=ops.ACT_RELU)
        finally:
            lib.ccvpe_set_narrow_kernels(prev)
        close(nchw(got).float(), nchw(ref).float().cpu(), 1e-2, "up2 vs tiled upconv")


@pytest.mark.parametrize("b,bf16", [(3, True), (4, True), (3, False)])
def test_upconv_level6_packs_two_images_per_tile(ops, b, bf16):
    """Level 6 (8 x 8 low-res images, N = 640): upconv_dma_kernel's PAIR form puts two images side by side in one 8 x 16
    tile (their halos 16 slots apart); an odd batch leaves the last tile half empty.  Against the torch composition, per image."""
    from ccvpe_amd.models import _pack_upconv
    cp, cref, cd, c1, co, h1, w1 = 1304, 1281, 1024, 320, 640, 8, 8
    dt = torch.bfloat16 if bf16 else torch.float32
    rnd = (lambda t: t.to(dt).float())
    x = rnd(synth.normal((b, cp, h1, w1), 650 + b))
    x[:, cref:] = 0
    skip = rnd(synth.normal((b, c1, 2 * h1, 2 * w1), 651))
    wd = synth.normal((cref, cd, 2, 2), 652, (1.0 / cref) ** 0.5)
    bd = synth.normal((cd,), 653, 0.3)
    w3 = synth.normal((co, cd + c1, 3, 3), 654, (1.0 / (9 * (cd + c1))) ** 0.5)
    b3 = synth.normal((co,), 655, 0.1)
    d = F.conv_transpose2d(x[:, :cref], wd, bd, stride=2)
    want = F.relu(F.conv2d(torch.cat([d, skip], 1), w3, b3, padding=1))
    fw, fshift = _pack_upconv(wd.cuda(), bd.cuda(), [(0, 0, cref)], cp, w3.cuda(), b3.cuda(), dt)
    got = ops.upconv3x3(nhwc(x).to(dt).cuda().contiguous(), cp, fw, fshift, co, batch=b, h1=h1, w1=w1,
                        src1=nhwc(skip).to(dt).cuda().contiguous(), c1=c1, act=ops.ACT_RELU)
    for i in range(b):
        close(nchw(got).float()[i], want[i], 2e-2 if bf16 else 1e-4, "level-6 pair, image %d" % i)


@pytest.mark.parametrize("cp,cref,cout,h1,w1,bf16", [
    (48, 41, 1, 32, 48, False),      # VIGOR loc level 1: [X/|X| 40 | max score | pad], fp32 (the arg-max decides here)
    (32, 32, 2, 16, 32, False),      # ori level 1, fp32
    (40, 33, 1, 32, 16, False),      # KITTI loc level 1 (ld 40: the third chunk is half empty)
    (32, 32, 2, 32, 32, True),       # ori level 1 in bf16 storage (one 32-channel chunk)
    (48, 41, 1, 16, 16, True),       # bf16 loc (two chunks, the second half empty)
    (32, 25, 1, 16, 32, True),       # bf16, one chunk, cout 1
])
def test_tail512_fuses_the_512_level(ops, cp, cref, cout, h1, w1, bf16):
    """deconv (k2 s2) -> conv3x3 + ReLU -> conv3x3 (16 -> cout) [-> F.normalize]  ==  ccvpe_tail512 (models.py:124-127,
    145-148,319,341), incl. image borders, tile aprons and the ragged last MFMA tile; and == the unfused HIP pair."""
    from ccvpe_amd.models import _pack_upconv
    b = 3
    dt = torch.bfloat16 if bf16 else torch.float32
    rnd = (lambda t: t.to(dt).float())
    x = rnd(synth.normal((b, cp, h1, w1), 700 + cp + cout))
    x[:, cref:] = 0
    wd = synth.normal((cref, 16, 2, 2), 702, (1.0 / cref) ** 0.5)
    bd = synth.normal((16,), 703, 0.3)
    w3 = synth.normal((16, 16, 3, 3), 704, (1.0 / (9 * 16)) ** 0.5)
    b3 = synth.normal((16,), 705, 0.1)
    w2 = synth.normal((cout, 16, 3, 3), 706, (1.0 / (9 * 16)) ** 0.5)
    b2 = synth.normal((cout,), 707, 0.1)
    mid = F.relu(F.conv2d(F.conv_transpose2d(x[:, :cref], wd, bd, stride=2), w3, b3, padding=1))
    want = F.conv2d(mid, w2, b2, padding=1)
    if cout == 2:
        want = F.normalize(want, p=2, dim=1)
    fw, fshift = _pack_upconv(wd.cuda(), bd.cuda(), [(0, 0, cref)], cp, w3.cuda(), b3.cuda(), dt)
    xd = nhwc(x).to(dt).cuda().contiguous()
    w2p = w2.permute(0, 2, 3, 1).contiguous().cuda()
    got = ops.tail512(xd, cp, fw, fshift, w2p, b2.cuda(), cout, cout == 2, batch=b, h1=h1, w1=w1)
    assert tuple(got.shape) == (b, cout, 2 * h1, 2 * w1) and got.dtype == torch.float32
    # cout = 2: the normalised field is ill-conditioned where the un-normalised vector is tiny -> compare where it is not
    if cout == 2:
        raw = F.conv2d(mid, w2, b2, padding=1)
        ok = (raw.pow(2).sum(1, keepdim=True).sqrt() > (0.2 if bf16 else 1e-2)).expand_as(want)
        err = ((got.cpu() - want).abs() * ok).max().item()
        assert err <= (3e-2 if bf16 else 1e-4), "tail512 ori: %.3e" % err
        assert float((got.pow(2).sum(1).sqrt() - 1).abs().max()) < 1e-5
    else:
        close(got.cpu(), want, 2e-2 if bf16 else 1e-5, "tail512 cp=%d" % cp)
    if cout == 1:
        # the heat-map softmax (models.py:319-320) from the per-(tile, wave) partials the same launch leaves behind
        lg, smx = ops.tail512(xd, cp, fw, fshift, w2p, b2.cuda(), 1, False, batch=b, h1=h1, w1=w1, want_softmax=True)
        assert torch.equal(lg, got)
        heat = ops.softmax_apply(lg.reshape(b, -1), smx)
        want_h = torch.softmax(lg.reshape(b, -1).double().cpu(), dim=1)
        assert float((heat.cpu().double() - want_h).abs().max() / want_h.max()) < 1e-5
        assert float((heat.sum(1) - 1).abs().max()) < 1e-5
    # the unfused HIP pair on the same operands
    y = ops.upconv3x3(xd, cp, fw, fshift, 16, batch=b, h1=h1, w1=w1, act=ops.ACT_RELU)
    ref2 = ops.head_conv3x3(y, w2p, b2.cuda(), cout, cout == 2)
    if cout == 1:
        close(got.cpu(), ref2.cpu(), 2e-2 if bf16 else 1e-5, "tail512 vs upconv + head")


@pytest.mark.parametrize("cp,cref,h1,w1", [(48, 41, 16, 32), (40, 33, 32, 16), (32, 25, 16, 48)])
def test_tail512_split_bf16_planes_are_fp32_class(ops, cp, cref, h1, w1):
    """The fp32 tail of the bf16 STORAGE path: fp32 operands split into bf16 hi + lo planes, hi.hi + lo.hi + hi.lo on the bf16
    matrix cores.  Against the float64 result: <= 3e-5 of scale (measured ~1e-5) where plain bf16 operands give ~4e-3 — and the
    exact fp32 kernel on the same operands agrees to that level."""
    from ccvpe_amd.models import _pack_upconv
    b = 2
    x = synth.normal((b, cp, h1, w1), 900 + cp)
    x[:, cref:] = 0
    wd = synth.normal((cref, 16, 2, 2), 902, (1.0 / cref) ** 0.5)
    bd = synth.normal((16,), 903, 0.3)
    w3 = synth.normal((16, 16, 3, 3), 904, (1.0 / (9 * 16)) ** 0.5)
    b3 = synth.normal((16,), 905, 0.1)
    w2 = synth.normal((1, 16, 3, 3), 906, (1.0 / (9 * 16)) ** 0.5)
    b2 = synth.normal((1,), 907, 0.1)
    dd = lambda t: t.double()
    mid = F.relu(F.conv2d(F.conv_transpose2d(dd(x[:, :cref]), dd(wd), dd(bd), stride=2), dd(w3), dd(b3), padding=1))
    want = F.conv2d(mid, dd(w2), dd(b2), padding=1)
    fw, fshift = _pack_upconv(wd.cuda(), bd.cuda(), [(0, 0, cref)], cp, w3.cuda(), b3.cuda(), torch.float32)
    xd = nhwc(x).cuda().contiguous()
    w2p = w2.permute(0, 2, 3, 1).contiguous().cuda()
    exact = ops.tail512(xd, cp, fw, fshift, w2p, b2.cuda(), 1, False, batch=b, h1=h1, w1=w1)
    split = ops.tail512(xd, cp, fw, fshift, w2p, b2.cuda(), 1, False, batch=b, h1=h1, w1=w1, split=True)
    close(exact.cpu(), want, 1e-5, "tail512 fp32")
    close(split.cpu(), want, 3e-5, "tail512 split")
    lg, smx = ops.tail512(xd, cp, fw, fshift, w2p, b2.cuda(), 1, False, batch=b, h1=h1, w1=w1, split=True, want_softmax=True)
    heat = ops.softmax_apply(lg.reshape(b, -1), smx)
    want_h = torch.softmax(lg.reshape(b, -1).double().cpu(), dim=1)
    assert float((heat.cpu().double() - want_h).abs().max() / want_h.max()) < 1e-5


# ------------------------------------------------------------------------------------------
# EfficientNet pieces
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("circ", [True, False])
def test_stem_golden(ops, circ, synth_sd):
    from ccvpe_amd.models import _fold_bn
    want = G.load("effnet_modules_" + ("circ" if circ else "zero"))["stem"]
    sd = synth_sd("vigor", 0)
    pfx = "grd_efficientnet" if circ else "sat_efficientnet"
    x = synth.normal((2, 3, 32, 48), 4242)
    sc, sh = _fold_bn(sd, pfx + "._bn0")
    got = ops.stem_conv(dev(x), dev(sd[pfx + "._conv_stem.weight"].permute(2, 3, 1, 0)), dev(sc), dev(sh), circ)
    close(nchw(got), torch.from_numpy(want), 1e-4, "stem")


@pytest.mark.parametrize("b,h,w,circ,bf16", [(2, 70, 301, True, False), (3, 37, 130, False, False), (1, 320, 640, True, True),
                                             (2, 154, 231, False, True)])
def test_stem_tiles_borders_and_wrap(ops, b, h, w, circ, bf16):
    """The MFMA stem on shapes that are not multiples of its 4 x 64 output tile, odd sizes (Oxford's 154 x 231), zero and circular
    padding (static SAME from the 224 schedule: pad (0, 1) on both axes, utils.py:265-277,341-353), fp32 and bf16 output."""
    x = synth.normal((b, 3, h, w), 4300 + h)
    wt = synth.normal((32, 3, 3, 3), 4301, (1.0 / 27) ** 0.5)
    sc = synth.uniform((32,), 4302, 0.5, 1.5)
    sh = synth.normal((32,), 4303, 0.1)
    xp = F.pad(x, (0, 1, 0, 0), mode="circular") if circ else F.pad(x, (0, 1, 0, 0))
    xp = F.pad(xp, (0, 0, 0, 1))
    want = O.swish(F.conv2d(xp, wt, stride=2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    got = ops.stem_conv(dev(x), dev(wt.permute(2, 3, 1, 0)), dev(sc), dev(sh), circ,
                        out_dtype=torch.bfloat16 if bf16 else torch.float32)
    assert tuple(got.shape) == (b, (h - 2) // 2 + 1, (w - 2) // 2 + 1, 32)
    close(nchw(got).float(), want, 1e-2 if bf16 else 1e-5, "stem %dx%d" % (h, w))
    raw = ops.stem_conv_raw(dev(x), dev(wt.permute(2, 3, 1, 0)), circ)
    close(nchw(raw), F.conv2d(xp, wt, stride=2), 1e-5, "stem raw")


@pytest.mark.parametrize("b,h,w,circ,bf16", [(2, 70, 300, True, False), (3, 37, 131, False, False), (1, 320, 640, True, True),
                                             (2, 154, 231, False, True), (2, 16, 64, True, False), (1, 512, 512, False, False)])
def test_stem_and_block0_depthwise_in_one_launch(ops, b, h, w, circ, bf16):
    """csrc/stem_dw.hip: stem conv + BN + swish -> depthwise 3x3 + BN + swish + squeeze partials with the stem tensor in LDS.
    Against the oracle's two convolutions (zero / circular padding of BOTH: the depthwise conv pads the stem OUTPUT), on shapes
    that are not multiples of the 8 x 32 output tile; fp32 must also equal the two unfused launches bit for bit."""
    x = synth.normal((b, 3, h, w), 4400 + h)
    wt = synth.normal((32, 3, 3, 3), 4401, (1.0 / 27) ** 0.5)
    s0, b0 = synth.uniform((32,), 4402, 0.5, 1.5), synth.normal((32,), 4403, 0.1)
    wd = synth.normal((32, 1, 3, 3), 4404, 1.0 / 3)
    s1, b1 = synth.uniform((32,), 4405, 0.5, 1.5), synth.normal((32,), 4406, 0.1)
    xp = F.pad(x, (0, 1, 0, 0), mode="circular") if circ else F.pad(x, (0, 1, 0, 0))
    xp = F.pad(xp, (0, 0, 0, 1))
    t = O.swish(F.conv2d(xp, wt, stride=2) * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1))
    want = O.swish(O.same_conv(t, wd, 3, 1, 224, circ, groups=32) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    dt = torch.bfloat16 if bf16 else torch.float32
    assert ops.stem_dw_supported(h, w, circ) == -(-t.shape[2] // 8) * -(-t.shape[3] // 32)
    args = (dev(x), dev(wt.permute(2, 3, 1, 0)), dev(s0), dev(b0))
    dwa = (dev(wd.reshape(32, 3, 3).permute(1, 2, 0)), dev(s1), dev(b1))
    got, part = ops.stem_dw(*args, *dwa, circ, out_dtype=dt)
    assert tuple(got.shape) == (b, t.shape[2], t.shape[3], 32)
    close(nchw(got).float(), want, 1e-2 if bf16 else 1e-5, "stem + dw %dx%d" % (h, w))
    close(part.sum(1), want.sum(dim=(2, 3)), 1e-2 if bf16 else 1e-4, "squeeze partials")
    if not bf16:
        u, part2 = ops.dwconv(ops.stem_conv(*args, circ), *dwa, 3, 1, circ)
        assert torch.equal(got, u), "the fused launch differs from stem_conv + dwconv"
        close(part.sum(1), part2.sum(1), 1e-5, "squeeze partials vs the unfused launches")


def test_stem_dw_rejects_odd_width_with_circular_padding(ops):
    assert ops.stem_dw_supported(64, 131, True) == 0 and ops.stem_dw_supported(64, 131, False) > 0
    with pytest.raises(Exception):
        ops.stem_dw(dev(synth.normal((1, 3, 64, 131), 1)), dev(torch.zeros(3, 3, 3, 32)), dev(torch.ones(32)), dev(torch.zeros(32)),
                    dev(torch.zeros(3, 3, 32)), dev(torch.ones(32)), dev(torch.zeros(32)), True)


@pytest.mark.parametrize("k,s,c,h,w,circ", [(3, 1, 32, 9, 12, False), (3, 2, 96, 10, 14, True), (5, 2, 144, 8, 12, True),
                                            (5, 1, 480, 6, 7, False), (5, 1, 1152, 5, 9, True), (3, 2, 240, 7, 9, False),
                                            (5, 2, 672, 9, 11, False), (3, 1, 1152, 4, 6, True)])
def test_dwconv_and_squeeze(ops, k, s, c, h, w, circ):
    b = 2
    x = synth.normal((b, c, h, w), 40 + c)
    wt = synth.normal((c, 1, k, k), 41, 1.0 / k)
    sc = synth.uniform((c,), 42, 0.5, 1.5)
    sh = synth.normal((c,), 43, 0.1)
    sched = 224                                        # even schedule size: pads (k-1)/2 or (0,1)/(1,2)
    y = O.same_conv(x, wt, k, s, sched, circ, groups=c)
    want = O.swish(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    got, part = ops.dwconv(dev(nhwc(x)), dev(wt.reshape(c, k, k).permute(1, 2, 0)), dev(sc), dev(sh), k, s, circ)
    close(nchw(got), want, 1e-4, "dwconv k%d s%d" % (k, s))
    close(part.sum(1), want.sum(dim=(2, 3)), 1e-4, "squeeze partials")


@pytest.mark.parametrize("b,c,cs,nblk", [(3, 240, 10, 7), (64, 1152, 48, 1), (2, 1152, 48, 1), (64, 672, 28, 1), (5, 480, 20, 1),
                                          (64, 96, 4, 128), (7, 144, 6, 100), (300, 32, 8, 64), (4, 30, 7, 5), (1, 2048, 100, 3)])
def test_se_gate(ops, b, c, cs, nblk):
    part = synth.normal((b, nblk, c), 50)
    w1 = synth.normal((cs, c), 51, 0.1)
    b1 = synth.normal((cs,), 52, 0.1)
    w2 = synth.normal((c, cs), 53, 0.3)
    b2 = synth.normal((c,), 54, 0.1)
    mean = part.sum(1) / 35.0
    z = O.swish(mean @ w1.t() + b1)
    want = torch.sigmoid(z @ w2.t() + b2)
    got = ops.se_gate(dev(part), 35, dev(w1), dev(b1), dev(w2.t()), dev(b2))
    close(got, want, 1e-5, "se gate b%d c%d" % (b, c))


@pytest.mark.parametrize("circ", [True, False])
def test_mbconv_blocks_golden(ops, circ, synth_sd):
    """Every MBConv block of the reference (golden from its own modules) via the HIP ops."""
    from ccvpe_amd.models import _pack_encoder
    want = G.load("effnet_modules_" + ("circ" if circ else "zero"))
    sd = {k: v.cuda() for k, v in synth_sd("vigor", 0).items() if "efficientnet" in k}
    pfx = "grd_efficientnet" if circ else "sat_efficientnet"
    e = _pack_encoder(sd, pfx)
    for i, blk in enumerate(e.blocks):
        if "block%d" % i not in want:
            continue
        xin = synth.normal((2, blk.cin) + G.BLOCK_HW, 5000 + i)
        x = dev(nhwc(xin))
        b, h, w, _ = x.shape
        t = x
        if blk.expand:
            t = ops.conv_igemm(x, blk.cin, blk.w_exp, blk.mid, batch=b, in_h=h, in_w=w, scale=blk.s0, shift=blk.b0,
                               act=ops.ACT_SWISH)
        u, part = ops.dwconv(t, blk.w_dw, blk.s1, blk.b1, blk.k, blk.s, circ)
        gate = ops.se_gate(part, u.shape[1] * u.shape[2], blk.se_w1, blk.se_b1, blk.se_w2, blk.se_b2)
        y = ops.conv_igemm(u, blk.mid, blk.w_proj, blk.cout, batch=b, in_h=u.shape[1], in_w=u.shape[2], gate=gate,
                           scale=blk.s2, shift=blk.b2, residual=x if blk.skip else None)
        close(nchw(y), torch.from_numpy(want["block%d" % i]), 1e-4, "block%d" % i)


@pytest.mark.parametrize("k,s,cin,h,w,circ", [(3, 2, 16, 40, 72, False), (3, 1, 24, 36, 40, True), (5, 2, 24, 34, 48, True),
                                              (5, 1, 40, 20, 24, False), (3, 2, 40, 18, 80, True), (5, 2, 24, 33, 21, False),
                                              (5, 1, 48, 17, 16, True)])
def test_mbconv_front_fused(ops, k, s, cin, h, w, circ):
    """Fused expand+depthwise (expanded tensor in LDS) == unfused oracle; several row bands."""
    from ccvpe_amd.models import _pack_conv
    b, mid = 2, 6 * cin
    x = synth.normal((b, cin, h, w), 300 + cin)
    w_exp = synth.normal((mid, cin, 1, 1), 301, (2.0 / cin) ** 0.5)
    s0, b0 = synth.uniform((mid,), 302, 0.5, 1.5), synth.normal((mid,), 303, 0.2)
    w_dw = synth.normal((mid, 1, k, k), 304, 1.0 / k)
    s1, b1 = synth.uniform((mid,), 305, 0.5, 1.5), synth.normal((mid,), 306, 0.2)
    t = O.swish(F.conv2d(x, w_exp) * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1))
    want = O.swish(O.same_conv(t, w_dw, k, s, 224, circ, groups=mid) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    assert ops.mbconv_front_supported(h, w, cin, mid, k, s) > 0
    got, part = ops.mbconv_front(dev(nhwc(x)), dev(_pack_conv(w_exp)), dev(s0), dev(b0),
                                 dev(w_dw.reshape(mid, k, k).permute(1, 2, 0)), dev(s1), dev(b1), mid, k, s, circ)
    close(nchw(got), want, 1e-4, "fused front k%d s%d" % (k, s))
    close(part.sum(1), want.sum(dim=(2, 3)), 1e-4, "squeeze partials")


PLANE_SHAPES = [(3, 1, 80, 32, 32, False), (5, 1, 80, 32, 32, False), (5, 1, 112, 32, 32, False), (5, 2, 112, 32, 32, False),
                (5, 1, 192, 16, 16, False), (3, 1, 192, 16, 16, False), (3, 1, 80, 20, 40, True), (5, 1, 112, 20, 40, True),
                (5, 2, 112, 20, 40, True), (5, 1, 192, 10, 20, True), (3, 1, 192, 10, 20, True), (5, 1, 80, 7, 12, True),
                (5, 2, 112, 18, 16, False), (3, 1, 112, 5, 8, False)]


def _plane_case(k, s, cin, h, w, circ, b=3, rnd=lambda t: t):
    mid = 6 * cin
    x = rnd(synth.normal((b, cin, h, w), 700 + cin + h))
    w_exp = rnd(synth.normal((mid, cin, 1, 1), 701, (2.0 / cin) ** 0.5))
    s0, b0 = synth.uniform((mid,), 702, 0.5, 1.5), synth.normal((mid,), 703, 0.2)
    w_dw = synth.normal((mid, 1, k, k), 704, 1.0 / k)
    s1, b1 = synth.uniform((mid,), 705, 0.5, 1.5), synth.normal((mid,), 706, 0.2)
    t = O.swish(F.conv2d(x, w_exp) * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1))
    want = O.swish(O.same_conv(t, w_dw, k, s, 224, circ, groups=mid) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    return x, w_exp, s0, b0, w_dw, s1, b1, t, want


@pytest.mark.parametrize("k,s,cin,h,w,circ", PLANE_SHAPES)
def test_mbconv_plane_late_blocks(ops, k, s, cin, h, w, circ):
    """The late-block front (csrc/mbconv_plane.hip: channel slices x whole-row bands, expanded plane in LDS) behind
    ccvpe_mbconv_front_*, against the unfused oracle (efficientnet_pytorch/model.py:102-110,114); its depthwise-only
    form behind ccvpe_dwconv_* on the same case; and the round-5 chain (switch off) as a third opinion."""
    from ccvpe_amd.models import _pack_conv
    from ccvpe_amd import _lib
    lib = _lib.load()
    mid = 6 * cin
    x, w_exp, s0, b0, w_dw, s1, b1, t, want = _plane_case(k, s, cin, h, w, circ)
    nblk = ops.mbconv_front_supported(h, w, cin, mid, k, s)
    assert nblk > 0, "the plane kernel should take this shape"
    wd = dev(w_dw.reshape(mid, k, k).permute(1, 2, 0))
    got, part = ops.mbconv_front(dev(nhwc(x)), dev(_pack_conv(w_exp)), dev(s0), dev(b0), wd, dev(s1), dev(b1), mid, k, s, circ)
    assert part.shape[1] == nblk
    close(nchw(got), want, 1e-4, "plane front k%d s%d" % (k, s))
    close(part.sum(1), want.sum(dim=(2, 3)), 1e-4, "squeeze partials")
    got2, part2 = ops.dwconv(dev(nhwc(t)), wd, dev(s1), dev(b1), k, s, circ)
    close(nchw(got2), want, 1e-4, "plane depthwise k%d s%d" % (k, s))
    close(part2.sum(1), want.sum(dim=(2, 3)), 1e-4, "squeeze partials (depthwise form)")
    prev = lib.ccvpe_set_mbconv_plane_kernels(0)
    try:
        assert ops.mbconv_front_supported(h, w, cin, mid, k, s) == 0
        got3, part3 = ops.dwconv(dev(nhwc(t)), wd, dev(s1), dev(b1), k, s, circ)
    finally:
        lib.ccvpe_set_mbconv_plane_kernels(prev)
    assert part3.shape[1] == 1
    close(got2, got3, 2e-6, "plane depthwise vs dwconv_plane_kernel")


@pytest.mark.parametrize("circ", [True, False])
def test_mbconv_blocks_golden_fused_front(ops, circ, synth_sd):
    """Blocks 1-5 through the fused front kernel against the reference-module goldens."""
    from ccvpe_amd.models import _pack_encoder
    want = G.load("effnet_modules_" + ("circ" if circ else "zero"))
    sd = {k: v.cuda() for k, v in synth_sd("vigor", 0).items() if "efficientnet" in k}
    e = _pack_encoder(sd, "grd_efficientnet" if circ else "sat_efficientnet")
    n = 0
    for i, blk in enumerate(e.blocks):
        if "block%d" % i not in want or not blk.expand:
            continue
        h, w = G.BLOCK_HW
        if not ops.mbconv_front_supported(h, w, blk.cin, blk.mid, blk.k, blk.s):
            continue
        x = dev(nhwc(synth.normal((2, blk.cin) + G.BLOCK_HW, 5000 + i)))
        u, part = ops.mbconv_front(x, blk.w_exp, blk.s0, blk.b0, blk.w_dw, blk.s1, blk.b1, blk.mid, blk.k, blk.s, circ)
        gate = ops.se_gate(part, u.shape[1] * u.shape[2], blk.se_w1, blk.se_b1, blk.se_w2, blk.se_b2)
        y = ops.conv_igemm(u, blk.mid, blk.w_proj, blk.cout, batch=2, in_h=u.shape[1], in_w=u.shape[2], gate=gate,
                           scale=blk.s2, shift=blk.b2, residual=x if blk.skip else None)
        close(nchw(y), torch.from_numpy(want["block%d" % i]), 1e-4, "block%d" % i)
        n += 1
    assert n >= 3


@pytest.mark.parametrize("cin,cout,b,hw,with_res", [(32, 16, 1, 256, False), (96, 24, 4, 128, False), (144, 24, 5, 128, True), (144, 40, 17, 64, False)])
def test_narrow_projection_kernel_fp32(ops, cin, cout, b, hw, with_res):
    """fp32 storage of csrc/pwn.hip (route CCVPE_ROUTE_PWN behind ccvpe_conv_igemm_f32): the early-block MBConv projections
    (efficientnet_pytorch/model.py:118-131) with the weights x SE gate in registers, exact fp32 matrix instructions — against the
    oracle and against the generic kernel (switch off).  240 -> 40 keeps the generic kernel in fp32 (registers)."""
    from ccvpe_amd import _lib
    from ccvpe_amd.models import _pack_conv
    lib = _lib.load()
    x = synth.normal((b, cin, hw, hw), 910 + cin)
    wt = synth.normal((cout, cin, 1, 1), 911 + cout, (1.0 / cin) ** 0.5)
    sc, sh = synth.uniform((cout,), 912, 0.5, 1.5), synth.normal((cout,), 913, 0.1)
    gate = synth.uniform((b, cin), 914, 0.1, 1.0)
    res = synth.normal((b, cout, hw, hw), 915) if with_res else None
    want = F.conv2d(x * gate.view(b, cin, 1, 1), wt) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if with_res:
        want = want + res
    kw = dict(batch=b, in_h=hw, in_w=hw, scale=dev(sc), shift=dev(sh), gate=dev(gate), residual=dev(nhwc(res)) if with_res else None)
    xd, wd = dev(nhwc(x)), dev(_pack_conv(wt))
    assert ops.conv_igemm(xd, cin, wd, cout, route_only=True, **kw)[0] == "pwn"
    got = ops.conv_igemm(xd, cin, wd, cout, **kw)
    close(nchw(got), want, 1e-5, "narrow projection fp32 %d->%d" % (cin, cout))
    prev = lib.ccvpe_set_pwn_kernels(0)
    try:
        ref = ops.conv_igemm(xd, cin, wd, cout, **kw)
    finally:
        lib.ccvpe_set_pwn_kernels(prev)
    close(got, ref.cpu(), 2e-6, "streaming kernel vs generic kernel")
    d = dict(kw)
    d.pop("gate")
    assert ops.conv_igemm(dev(nhwc(synth.normal((4, 240, 64, 64), 1))), 240, dev(_pack_conv(synth.normal((40, 240, 1, 1), 2))), 40, route_only=True,
                          batch=4, in_h=64, in_w=64, scale=dev(synth.uniform((40,), 3, 0.5, 1.5)), shift=dev(synth.normal((40,), 4, 0.1)))[0] == "igemm"


# ------------------------------------------------------------------------------------------
# descriptors / matching / heads
# ------------------------------------------------------------------------------------------
def test_ground_descriptor(ops):
    b, h, w = 2, 10, 20
    cd = (64, 32, 16, 8, 4, 2)
    y1 = synth.normal((b, h, w, 128), 60)
    wh = synth.normal((6, h), 61, 0.3)
    bh = synth.normal((6,), 62, 0.1)
    got = ops.ground_descriptor(dev(y1), dev(wh), dev(bh), cd)
    off, outs = 0, []
    for l in range(6):
        d = torch.einsum("bhwc,h->bwc", y1[..., off:off + cd[l]], wh[l]) + bh[l]
        outs.append(d.reshape(b, -1))
        off += cd[l]
    close(got, torch.cat(outs, 1), 1e-5, "ground descriptor")


@pytest.mark.parametrize("C,L,stride,hw,shifts,n_max,n_tail", [
    (1280, 1280, 64, 8, list(range(20)), 20, 20),                    # VIGOR level 1 (train model)
    (1280, 1280, 64, 8, [0] + list(range(20)), 1, 20),               # ori_prior(0) level 1
    (1280, 640, 64, 8, list(range(-10, 11)) + list(range(20)), 21, 20),   # ori_prior(180), FoV 180
    (640, 640, 32, 16, list(range(20)), 20, 0),                      # level 2
    (40, 40, 2, 20, list(range(20)), 20, 0),                         # level 6 (stride 2 -> 8-byte path)
    (40, 20, 2, 20, list(range(-10, 11)), 21, 0),                    # level 6, FoV 180
    (160, 160, 8, 23, [0], 1, 0),                                    # N_rot = 1, ragged pixel count
    (2048, 512, 128, 8, list(range(16)), 16, 16),                    # KITTI level 1
    (128, 32, 8, 16, list(range(16)), 16, 0),                        # KITTI level 5 (wrapping duplicates)
    (32, 32, 8, 32, list(range(16)), 16, 0),                         # KITTI level 6
    (80, 80, 4, 16, list(range(-3, 4)), 7, 0),                       # ori_prior(54) level 5
])
def test_match_level(ops, C, L, stride, hw, shifts, n_max, n_tail):
    b = 2
    x = synth.normal((b, C, hw, hw), 70 + C)
    g = synth.normal((b, L + 5), 71 + L)          # wider row: exercises ldg
    ldo = (C + 1 + n_tail + 7) // 8 * 8
    sc, cat = ops.match_level(dev(nhwc(x)), dev(g)[:, :L], L, shifts, n_max, n_tail, stride, ldo)
    want = O.rotational_matching(x, g[:, :L], shifts, stride)
    close(sc, want, 2e-5, "scores")
    catc = nchw(cat).cpu()
    close(catc[:, :C], F.normalize(x, p=2, dim=1), 1e-5, "normalised features")
    close(catc[:, C], want[:, :n_max].max(dim=1)[0], 2e-5, "max over rotations")
    if n_tail:
        close(catc[:, C + 1:C + 1 + n_tail], want[:, len(shifts) - n_tail:], 2e-5, "tail scores")
    assert (catc[:, C + 1 + n_tail:] == 0).all()


@pytest.mark.parametrize("C,L,stride,hw,nsh,n_tail", [(80, 80, 4, 19, 20, 0), (160, 160, 8, 13, 20, 0), (320, 320, 16, 9, 20, 0),
                                                      (80, 40, 4, 12, 21, 0), (160, 80, 8, 11, 12, 0), (320, 160, 16, 7, 9, 9),
                                                      (640, 320, 32, 5, 21, 0), (40, 40, 2, 33, 32, 0), (40, 20, 2, 17, 16, 0),
                                                      (1280, 1280, 64, 3, 20, 20), (48, 48, 4, 21, 11, 0)])
def test_match_level_matrix_core_form(ops, C, L, stride, hw, nsh, n_tail):
    """9 <= N_rot <= 32: the circulant [N_rot x C] of the ground descriptor times the [C x pixels] tile on v_mfma_f32_16x16x4_f32
    (models.py:191-202 and the five blocks after it as one dense contraction; north_star's "MFMA for the descriptor x aerial-patch
    inner product").  Against the oracle, and against the vector-ALU form of the same kernel (switch off): pixel tiles per wave
    (C <= 80), K slices per wave (C >= 160), half-filled tiles (C >= 640), ragged pixel counts, partial windows, 1 and 2 row tiles."""
    from ccvpe_amd import _lib
    lib = _lib.load()
    b = 3
    shifts = list(range(-(nsh // 2), nsh - nsh // 2)) if L < C else list(range(nsh))
    x = synth.normal((b, C, hw, hw), 170 + C)
    g = synth.normal((b, L + 3), 171 + L)
    ldo = (C + 1 + n_tail + 7) // 8 * 8
    n_max = nsh
    sc, cat = ops.match_level(dev(nhwc(x)), dev(g)[:, :L], L, shifts, n_max, n_tail, stride, ldo)
    want = O.rotational_matching(x, g[:, :L], shifts, stride)
    close(sc, want, 2e-5, "scores (matrix cores)")
    catc = nchw(cat).cpu()
    close(catc[:, :C], F.normalize(x, p=2, dim=1), 1e-5, "normalised features")
    close(catc[:, C], want[:, :n_max].max(dim=1)[0], 2e-5, "max over rotations")
    if n_tail:
        close(catc[:, C + 1:C + 1 + n_tail], want[:, len(shifts) - n_tail:], 2e-5, "tail scores")
    assert (catc[:, C + 1 + n_tail:] == 0).all()
    prev = lib.ccvpe_set_match_mfma(0)
    try:
        sc2, cat2 = ops.match_level(dev(nhwc(x)), dev(g)[:, :L], L, shifts, n_max, n_tail, stride, ldo)
    finally:
        lib.ccvpe_set_match_mfma(prev)
    close(sc, sc2.cpu(), 4e-6, "matrix-core form vs vector-ALU form")
    close(cat, cat2.cpu(), 4e-6, "concat rows, both forms")


@pytest.mark.parametrize("C,L,stride,side,nsh,b,bf16", [(40, 40, 2, 256, 20, 1, False), (40, 20, 2, 256, 21, 1, False), (80, 80, 4, 128, 20, 4, False),
                                                          (80, 40, 4, 128, 16, 5, False), (48, 48, 4, 192, 11, 2, False), (64, 64, 4, 144, 9, 4, False),
                                                          (40, 40, 2, 256, 20, 2, True), (80, 40, 4, 128, 21, 4, True), (40, 40, 3, 256, 20, 1, False)])
def test_match_level_streaming_form(ops, C, L, stride, side, nsh, b, bf16):
    """The narrow levels (C <= 80, >= 65 536 pixels, no tail scores) as a streaming kernel: every wave walks its own run of 16-pixel
    tiles, x straight from global memory in MFMA layout, the circulant fragments of the sample's descriptor rebuilt in registers at
    sample boundaries (match_stream_kernel).  Against the oracle and against the tiled matrix-core form (switch = 1); odd offsets
    (stride 3) are served too — the fragments are built element by element."""
    from ccvpe_amd import _lib
    lib = _lib.load()
    shifts = list(range(-(nsh // 2), nsh - nsh // 2)) if L < C else list(range(nsh))
    x = synth.normal((b, C, side, side), 270 + C)
    g = synth.normal((b, L + 3), 271 + L)
    if bf16:
        x = x.to(torch.bfloat16).float()
    xd = dev(nhwc(x)).to(torch.bfloat16) if bf16 else dev(nhwc(x))
    ldo = (C + 1 + 7) // 8 * 8
    sc, cat = ops.match_level(xd, dev(g)[:, :L], L, shifts, nsh, 0, stride, ldo)
    want = O.rotational_matching(x, g[:, :L], shifts, stride)
    close(sc, want, 2e-5, "scores (streaming form)")
    catc = nchw(cat.float()).cpu()
    tol = 1e-2 if bf16 else 1e-5
    close(catc[:, :C], F.normalize(x, p=2, dim=1), tol, "normalised features")
    close(catc[:, C], want.max(dim=1)[0], 1e-2 if bf16 else 2e-5, "max over rotations")
    assert (catc[:, C + 1:] == 0).all()
    prev = lib.ccvpe_set_match_mfma(1)
    try:
        sc2, cat2 = ops.match_level(xd, dev(g)[:, :L], L, shifts, nsh, 0, stride, ldo)
    finally:
        lib.ccvpe_set_match_mfma(prev)
    close(sc, sc2.cpu(), 4e-6, "streaming form vs tiled form")
    close(cat.float(), cat2.float().cpu(), 1e-2 if bf16 else 4e-6, "concat rows, both forms")


def test_match_level_streaming_form_propagates_nan_like_torch_max(ops):
    b, C, side = 1, 40, 256
    x = synth.normal((b, C, side, side), 290)
    x[:, :, 100, 37] = 0.0
    g = synth.normal((b, C), 291)
    shifts = list(range(20))
    sc, cat = ops.match_level(dev(nhwc(x)), dev(g), C, shifts, 20, 0, 2, 48)
    assert not torch.isfinite(sc[0, :, 100, 37]).any() and not torch.isfinite(nchw(cat)[0, C, 100, 37])
    assert torch.isfinite(sc[0, :, 100, 36]).all() and (nchw(cat)[0, :C, 100, 37] == 0).all()


def test_match_level_matrix_core_form_propagates_nan_like_torch_max(ops):
    """A zero-norm window (models.py:196 has no eps) makes that hypothesis NaN; torch.max over the stack then returns NaN."""
    b, C, hw = 1, 40, 8
    x = synth.normal((b, C, hw, hw), 190)
    x[:, :, 2, 3] = 0.0
    g = synth.normal((b, C), 191)
    shifts = list(range(20))
    sc, cat = ops.match_level(dev(nhwc(x)), dev(g), C, shifts, 20, 0, 2, 48)
    want = O.rotational_matching(x, g, shifts, 2)
    assert not torch.isfinite(sc[0, :, 2, 3]).any() and not torch.isfinite(want[0, :, 2, 3]).any()
    assert not torch.isfinite(nchw(cat)[0, C, 2, 3])
    assert torch.isfinite(sc[0, :, 0, 0]).all() and (nchw(cat)[0, :C, 2, 3] == 0).all()


def test_match_level_zero_window_gives_nonfinite_like_reference(ops):
    """models.py:196 has no eps: a zero-norm window divides by zero.  Reproduced, not 'fixed'."""
    b, C, hw = 1, 40, 8
    x = synth.normal((b, C, hw, hw), 90)
    x[:, :, 0, 0] = 0.0
    g = synth.normal((b, C), 91)
    sc, cat = ops.match_level(dev(nhwc(x)), dev(g), C, [0, 1], 2, 0, 2, 48)
    want = O.rotational_matching(x, g, [0, 1], 2)
    assert not torch.isfinite(sc[0, :, 0, 0]).any() and not torch.isfinite(want[0, :, 0, 0]).any()
    assert (nchw(cat)[0, :C, 0, 0] == 0).all()          # F.normalize eps path


@pytest.mark.parametrize("cout", [1, 2])
def test_head_conv(ops, cout):
    b, hw = 2, 21
    x = synth.normal((b, 16, hw, hw), 80)
    wt = synth.normal((cout, 16, 3, 3), 81, 0.1)
    bias = synth.normal((cout,), 82, 0.1)
    want = F.conv2d(x, wt, bias, padding=1)
    if cout == 2:
        want = F.normalize(want, p=2, dim=1)
    got = ops.head_conv3x3(dev(nhwc(x)), dev(wt.permute(0, 2, 3, 1)), dev(bias), cout, cout == 2)
    close(got, want, 1e-5, "head conv")


def test_softmax_rows(ops):
    lg = synth.normal((3, 262144), 85, 2.0)
    got = ops.softmax_rows(dev(lg))
    want = torch.softmax(lg, dim=1)
    close(got, want, 1e-5, "softmax")
    assert torch.equal(got.argmax(1).cpu(), want.argmax(1))


def test_losses_golden(ops):
    want = G.load("losses")
    for n_cols in (1280, 20480):
        sc = synth.uniform((3, n_cols), 8000 + n_cols, -1.0, 1.0)
        lab = synth.uniform((3, n_cols), 8100 + n_cols) ** 6
        got = ops.infonce_loss(dev(sc), dev(lab)).item()
        G.assert_close(got, want["infonce_%d" % n_cols], 1e-4, 0, "infonce")
    lg = synth.normal((3, 262144), 8200, 2.0)
    lab = synth.uniform((3, 262144), 8201) ** 20
    lab = lab / lab.sum(1, keepdim=True)
    G.assert_close(ops.cross_entropy_loss(dev(lg), dev(lab)).item(), want["ce"], 1e-4, 0, "ce")
    ori = F.normalize(synth.normal((3, 2, 512, 512), 8300), dim=1)
    gto = F.normalize(synth.normal((3, 2, 512, 512), 8301), dim=1)
    got = ops.orientation_loss(dev(ori), dev(gto), dev(lab.reshape(3, 1, 512, 512))).item()
    G.assert_close(got, want["ori"], 1e-4, 0, "ori loss")


def test_eval_postprocess(ops):
    b = 4
    heat = torch.softmax(synth.normal((b, 512 * 512), 95, 3.0), dim=1).reshape(b, 1, 512, 512)
    heat[1, 0, 100, 7] = heat[1].max() + 0.1          # clear peak
    heat[2, 0, 5, 9] = 0.5; heat[2, 0, 300, 300] = 0.5   # tie -> first index wins (numpy.argmax)
    ori = F.normalize(synth.normal((b, 2, 512, 512), 96), dim=1)
    got = ops.eval_postprocess(dev(heat), dev(ori)).cpu()
    want = O.eval_postprocess(heat, ori)
    assert torch.equal(got[:, :2], want[:, :2])
    assert got[2, 0] == 5 and got[2, 1] == 9
    close(got[:, 2:4], want[:, 2:4], 1e-6, "cos/sin")
    assert (got[:, 4] - want[:, 4]).abs().max() < 1e-2   # degrees (acosf vs libm acos in fp64)
    close(got[:, 5], want[:, 5], 1e-6, "prob")


@pytest.mark.parametrize("C,L,stride,hw,woff", [(1280, 224, 64, 8, 528), (160, 28, 8, 16, 66), (80, 14, 4, 16, 33),
                                                (40, 7, 2, 24, 16)])
def test_match_level_centred_window(ops, C, L, stride, hw, woff):
    """CVM_OxfordRobotCar's window (models.py:1094): channels [woff, woff+L) of the rolled volume, odd offsets included."""
    shifts = list(range(20))
    x = synth.normal((2, C, hw, hw), 4100 + C)
    g = synth.normal((2, L), 4101)
    ldo = (C + 1 + 20 + 7) // 8 * 8
    sc, cat = ops.match_level(dev(nhwc(x)), dev(g), L, shifts, 20, 20, stride, ldo, window_offset=woff)
    want = O.rotational_matching(x, g, shifts, stride, woff)
    close(sc, want, 1e-5, "centred-window scores")
    close(cat[..., C], want.max(dim=1)[0], 1e-5, "max column")


@pytest.mark.parametrize("kind,c0,c1,n,hw,k", [("3x3", 1024, 320, 640, 16, 3), ("1x1", 1152, 0, 192, 10, 1),
                                               ("2x2s2", 1280, 0, 1280, 16, 2), ("deconv", 1288, 0, 4096, 8, 1)])
def test_conv_igemm_split_k_matches_one_pass(ops, kind, c0, c1, n, hw, k):
    """Small-batch GEMMs take the split-K path (K slices + deterministic second pass): same result as the one-pass
    kernels (bias, ReLU, residual, two sources, pixel-shuffle output), and the planner engages on these shapes."""
    import ctypes
    from ccvpe_amd import _lib
    from ccvpe_amd.models import _pack_conv, _pack_deconv
    b = 1
    x0 = dev(synth.normal((b, hw, hw, c0), 5000 + c0))
    x1 = dev(synth.normal((b, hw, hw, c1), 5001)) if c1 else None
    bias = dev(synth.normal((n if kind != "deconv" else n // 4,), 5002, 0.1))
    kw = dict(batch=b, in_h=hw, in_w=hw)
    if kind == "deconv":
        w = synth.normal((c0, n // 4, 2, 2), 5003, c0 ** -0.5)
        wp, bp = _pack_deconv(dev(w), bias, [(0, 0, c0)], c0)
        args = dict(shift=bp, out_mode=ops.OUT_DECONV2X, **kw)
    else:
        w = synth.normal((n, c0 + c1, k, k), 5003, ((c0 + c1) * k * k) ** -0.5)
        wp = _pack_conv(dev(w))
        st, pad = (2, 0) if kind == "2x2s2" else (1, k // 2)
        args = dict(kh=k, kw=k, stride=st, pad=pad, src1=x1, c1=c1, shift=bias, act=ops.ACT_RELU if k == 3 else ops.ACT_NONE,
                    **kw)
        if kind == "1x1":
            args["residual"] = dev(synth.normal((b, hw, hw, n), 5004))
    old = ops.SPLIT_K
    try:
        ops.SPLIT_K = False
        ref = ops.conv_igemm(x0, c0, wp, n, **args)
        ops.SPLIT_K = True
        calls = ops.SPLIT_K_CALLS
        got = ops.conv_igemm(x0, c0, wp, n, **args)
        assert ops.SPLIT_K_CALLS == calls + 1, "the split-K planner did not engage on a small-batch shape"
    finally:
        ops.SPLIT_K = old
    close(got, ref, 2e-5, "split-K vs one pass")
    got2 = ops.conv_igemm(x0, c0, wp, n, **args)
    assert torch.equal(got, got2), "split-K result must be deterministic"
