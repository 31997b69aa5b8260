"""Backward building blocks (ccvpe_amd/backward.py) against torch autograd on the CPU (fp32).
Tolerance 2e-4 of the gradient scale (the weight gradient reduces over thousands of pixels)."""
import pytest
import torch
import torch.nn.functional as F

from ccvpe_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bw():
    from ccvpe_amd import backward, _lib
    _lib.load()
    return backward


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def close(got, want, tol, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("c0,c1,n,k,stride,pad,h,w", [(40, 16, 40, 3, 1, 1, 12, 14), (96, 0, 24, 1, 1, 0, 9, 13),
                                                       (16, 0, 96, 1, 1, 0, 17, 11), (64, 0, 48, 2, 2, 0, 8, 8),
                                                       (320, 112, 320, 3, 1, 1, 6, 6), (16, 0, 16, 3, 1, 1, 33, 20),
                                                       (1024, 320, 640, 3, 1, 1, 4, 4), (1280, 0, 126, 1, 1, 0, 10, 20),
                                                       (104, 0, 80, 3, 1, 1, 9, 11), (200, 0, 160, 3, 1, 1, 7, 9),
                                                       (432, 0, 320, 3, 1, 1, 5, 6), (56, 0, 40, 3, 1, 1, 13, 9)])
def test_conv_wgrad_and_dgrad_vs_autograd(bw, c0, c1, n, k, stride, pad, h, w):
    b = 3
    x = synth.normal((b, c0 + c1, h, w), 800 + c0).requires_grad_(True)
    wt = synth.normal((n, c0 + c1, k, k), 801, (1.0 / ((c0 + c1) * k * k)) ** 0.5).requires_grad_(True)
    bias = synth.normal((n,), 802, 0.1).requires_grad_(True)
    y = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    dy = synth.normal(tuple(y.shape), 803)
    y.backward(dy)
    xs = nhwc(x.detach())
    x0 = xs[..., :c0].contiguous().cuda()
    x1 = xs[..., c0:].contiguous().cuda() if c1 else None
    dyd = nhwc(dy).cuda()
    if n % 4:                                   # pixel stride must be a multiple of 4 floats: pad like the model
        dyd = F.pad(dyd, (0, 4 - n % 4)).contiguous()
    close(bw.conv_wgrad(x0, dyd, n, k, k, stride, pad, x1), wt.grad, 2e-4, "wgrad")
    close(bw.bias_grad(dyd)[:n], bias.grad, 2e-4, "bias grad")
    dw2, db2 = bw.conv_wgrad(x0, dyd, n, k, k, stride, pad, x1, want_bias=True)      # bias gradient from the same launch
    close(dw2, wt.grad, 2e-4, "wgrad (+bias launch)")
    close(db2, bias.grad, 2e-4, "bias grad fused into the weight-gradient launch")
    dyd = dyd[..., :n].contiguous() if n % 4 == 0 else None
    if dyd is None:
        return                                  # dgrad of the padded case is the plain 1x1 path tested above
    wd = wt.detach().cuda()
    if k == 1:
        dx = bw.conv1x1_dgrad(dyd, wd)
    elif k == 3:
        dx = bw.conv3x3_dgrad(dyd, wd)
    else:
        dx = bw.conv2x2s2_dgrad(dyd, wd)
    close(dx.permute(0, 3, 1, 2), x.grad, 2e-4, "dgrad")
    if k == 3:                                  # gradient in front of a ReLU that produced x: same launch, masked store
        keep = torch.relu(synth.normal((b, h, w, c0 + c1), 804)).cuda()
        dxm = bw.conv3x3_dgrad(dyd, wd, relu_out=keep)
        assert torch.equal(dxm, dx * (keep > 0)), "fused ReLU mask differs from relu_bwd of the plain dgrad"


@pytest.mark.parametrize("cin,cout,h", [(48, 16, 9), (648, 320, 3), (168, 40, 5)])
def test_deconv_backward_vs_autograd(bw, cin, cout, h):
    b = 2
    x = synth.normal((b, cin, h, h), 810 + cin).requires_grad_(True)
    wt = synth.normal((cin, cout, 2, 2), 811, (1.0 / cin) ** 0.5).requires_grad_(True)
    y = F.conv_transpose2d(x, wt, None, stride=2)
    dy = synth.normal(tuple(y.shape), 812)
    y.backward(dy)
    dyd = nhwc(dy).cuda()
    close(bw.deconv_wgrad(nhwc(x.detach()).cuda(), dyd), wt.grad, 2e-4, "deconv wgrad")
    close(bw.deconv_dgrad(dyd, wt.detach().cuda()).permute(0, 3, 1, 2), x.grad, 2e-4, "deconv dgrad")


# ---- EfficientNet train-mode backward pieces --------------------------------------------------------------

def _act(z, act):
    return z * torch.sigmoid(z) if act == 2 else (torch.relu(z) if act == 1 else z)


@pytest.mark.parametrize("c,h,w,act,with_se,with_dcs", [(32, 20, 24, 2, True, False), (96, 9, 7, 2, False, True),
                                                        (40, 5, 6, 0, False, True), (1152, 3, 4, 2, True, False),
                                                        (16, 33, 17, 1, False, False)])
def test_bn_act_bwd_vs_autograd(bw, c, h, w, act, with_se, with_dcs):
    from ccvpe_amd import ops
    b, eps = 4, 1e-3
    x = synth.normal((b, h, w, c), 900 + c).double().requires_grad_(True)
    gamma = (1.0 + 0.3 * synth.normal((c,), 901)).double().requires_grad_(True)
    beta = (0.2 * synth.normal((c,), 902)).double().requires_grad_(True)
    gate = torch.sigmoid(synth.normal((b, c), 903)).double().requires_grad_(True) if with_se else None
    dcs = (synth.uniform((b,), 904) > 0.3).double() / 0.7 if with_dcs else None
    mean = x.mean(dim=(0, 1, 2))
    var = x.var(dim=(0, 1, 2), unbiased=False)
    u = _act((x - mean) / torch.sqrt(var + eps) * gamma + beta, act)
    if dcs is not None:
        u = u * dcs.view(b, 1, 1, 1)
    if with_se:
        pooled = u.mean(dim=(1, 2))                       # the squeeze branch: loss also depends on the mean
        wm = synth.normal((b, c), 905).double()
        v = u * gate.view(b, 1, 1, c)
    else:
        v = u
    dv = synth.normal((b, h, w, c), 906).double()
    loss = (v * dv).sum()
    if with_se:
        loss = loss + (pooled * wm).sum()
    loss.backward()
    xd = x.detach().float().cuda()
    md, vd = mean.detach().float().cuda(), var.detach().float().cuda()
    gd, bd = gamma.detach().float().cuda(), beta.detach().float().cuda()
    dvd = dv.float().cuda()
    gated = gate.detach().float().cuda() if with_se else None
    dmean = (wm / (h * w)).float().cuda() if with_se else None
    dcsd = dcs.float().cuda() if with_dcs else None
    dx, dgamma, dbeta = bw.bn_act_bwd(xd, dvd, md, vd, gd, bd, eps, act, gated, dmean, dcsd)
    close(dx, x.grad, 2e-4, "bn dx")
    close(dgamma, gamma.grad, 2e-4, "dgamma")
    close(dbeta, beta.grad, 2e-4, "dbeta")
    if with_se:
        part = bw.se_dgate_partials(xd, dvd, md, vd, gd, bd, eps, act)
        close(part.sum(dim=1), gate.grad, 2e-4, "dgate")


@pytest.mark.parametrize("c,k,stride,h,w,circ", [(32, 3, 1, 12, 16, True), (96, 3, 2, 16, 20, True),
                                                 (144, 5, 2, 10, 14, False), (240, 5, 1, 7, 9, True),
                                                 (1152, 3, 1, 4, 4, False), (672, 5, 2, 8, 8, True),
                                                 # planes above 1 024 pixels: the all-taps weight-gradient kernel (the rows above run dw_wgrad_rows_kernel)
                                                 (32, 3, 1, 40, 48, True), (96, 3, 2, 34, 38, True), (144, 5, 2, 36, 40, False), (240, 5, 1, 33, 36, True)])
def test_dwconv_backward_vs_autograd(bw, c, k, stride, h, w, circ):
    from oracle import ccvpe_oracle as orc
    b = 3
    x = synth.normal((b, c, h, w), 910 + c).requires_grad_(True)
    wt = synth.normal((c, 1, k, k), 911, 0.3).requires_grad_(True)
    y = orc.same_conv(x, wt, k, stride, 224, circ, groups=c)
    dy = synth.normal(tuple(y.shape), 912)
    y.backward(dy)
    wp = wt.detach().reshape(c, k * k).t().contiguous().cuda()         # [k*k][C]
    dyd, xd = nhwc(dy).cuda(), nhwc(x.detach()).cuda()
    close(bw.dwconv_dgrad(dyd, wp, h, w, k, stride, circ), nhwc(x.grad), 2e-4, "dw dgrad")
    close(bw.dwconv_wgrad(xd, dyd, k, stride, circ), wt.grad.reshape(c, k * k).t(), 2e-4, "dw wgrad")


@pytest.mark.parametrize("c,cs,hw", [(32, 8, 100), (1152, 48, 12), (96, 4, 35)])
def test_se_bwd_vs_autograd(bw, c, cs, hw):
    b = 5
    u = synth.normal((b, hw, c), 920 + c).double()
    w1 = synth.normal((cs, c), 921, c ** -0.5).double().requires_grad_(True)
    b1 = synth.normal((cs,), 922, 0.1).double().requires_grad_(True)
    w2 = synth.normal((c, cs), 923, cs ** -0.5).double().requires_grad_(True)
    b2 = synth.normal((c,), 924, 0.1).double().requires_grad_(True)
    m = u.mean(dim=1).requires_grad_(True)
    z1 = m @ w1.t() + b1
    gate = torch.sigmoid((z1 * torch.sigmoid(z1)) @ w2.t() + b2)
    dgate = synth.normal((b, c), 925).double()
    (gate * dgate).sum().backward()
    # forward squeeze partials in 3 chunks, dgate partials in 2
    chunks = [u[:, i::3].sum(dim=1) for i in range(3)]
    sep = torch.stack(chunks, dim=1).float().cuda()
    dgp = torch.stack([0.25 * dgate, 0.75 * dgate], dim=1).float().cuda()
    dmean, dw1, db1, dw2, db2 = bw.se_bwd(sep, hw, dgp, w1.detach().float().cuda(), b1.detach().float().cuda(),
                                          w2.detach().t().contiguous().float().cuda(), b2.detach().float().cuda())
    close(dmean, m.grad / hw, 2e-4, "dmean")
    close(dw1, w1.grad, 2e-4, "dw1")
    close(db1, b1.grad, 2e-4, "db1")
    close(dw2, w2.grad, 2e-4, "dw2")
    close(db2, b2.grad, 2e-4, "db2")


@pytest.mark.parametrize("b,rows,c", [(3, 20 * 24, 96), (2, 33 * 17, 240), (5, 7 * 9, 1152)])
def test_se_bn_bwd_two_pass_equals_three_pass(bw, b, rows, c):
    """BatchNorm + squeeze-excite backward: the two-pass form (five per-(sample, channel) sums, then apply: what the training
    step runs) against the three-pass form (se_dgate partials + bn_act_bwd) it replaces — same dgate, dx, dgamma, dbeta."""
    from ccvpe_amd import ops
    x = (synth.normal((b, rows, c), 950 + c) * 1.3 + 0.2).cuda()
    dv = synth.normal((b, rows, c), 951).cuda()
    mean, var = x.mean(dim=(0, 1)).contiguous(), x.var(dim=(0, 1), unbiased=False).contiguous()
    gamma, beta = synth.uniform((c,), 952, 0.5, 1.5).cuda(), synth.normal((c,), 953, 0.3).cuda()
    gate, dmean = synth.uniform((b, c), 954, 0.1, 0.9).cuda(), (synth.normal((b, c), 955) / rows).cuda()
    dgp = bw.se_dgate_partials(x, dv, mean, var, gamma, beta, 1e-3, ops.ACT_SWISH)
    dx0, dg0, db0 = bw.bn_act_bwd(x, dv, mean, var, gamma, beta, 1e-3, ops.ACT_SWISH, gate=gate, dmean=dmean)
    sums = bw.se_bn_bwd_reduce(x, dv, mean, var, gamma, beta, 1e-3, ops.ACT_SWISH)
    dx1, dg1, db1 = bw.se_bn_bwd_apply(x, dv, mean, var, gamma, beta, 1e-3, ops.ACT_SWISH, gate, dmean, sums)
    close(sums[0], dgp.sum(1), 2e-5, "dgate")
    close(dg1, dg0, 5e-5, "dgamma")
    close(db1, db0, 5e-5, "dbeta")
    close(dx1, dx0, 5e-5, "dx")


@pytest.mark.parametrize("c,n,h,w", [(96, 16, 19, 23), (144, 24, 12, 30), (240, 40, 9, 11), (480, 80, 8, 8), (672, 112, 6, 7),
                                     (1152, 192, 5, 4), (1152, 320, 4, 4)])
def test_gated_1x1_wgrad_equals_wgrad_of_the_materialised_product(bw, c, n, h, w):
    """MBConv projection: weight gradient with the SE gate applied while x is staged == weight gradient of u * gate (every
    row-tile family the projection layers use)."""
    b = 3
    u = synth.normal((b, h, w, c), 960 + c).cuda()
    gate = synth.uniform((b, c), 961, 0.05, 0.95).cuda()
    dy = synth.normal((b, h, w, n), 962).cuda()
    want = bw.conv_wgrad(bw.gate_mul(u, gate), dy, n, 1, 1, 1, 0)
    got = bw.conv1x1_wgrad_gated(u, gate, dy, n)
    close(got, want, 1e-5, "gated 1x1 wgrad")
    ref = torch.einsum("bhwn,bhwc->nc", dy.double().cpu(), (u * gate.view(b, 1, 1, c)).double().cpu())
    close(got.reshape(n, c), ref, 2e-4, "gated 1x1 wgrad vs float64")


def test_relu_bwd(bw):
    y = torch.relu(synth.normal((2, 5, 6, 16), 930))
    dy = synth.normal((2, 5, 6, 16), 931)
    close(bw.relu_bwd(y.cuda(), dy.cuda()), dy * (y > 0), 1e-7, "relu bwd")


# ---- heads / glue backward ----------------------------------------------------------------------------------

def test_softmax_bwd(bw):
    logits = (3.0 * synth.normal((3, 4096), 940)).double().requires_grad_(True)
    h = torch.softmax(logits, dim=1)
    dh = synth.normal((3, 4096), 941).double()
    dl = synth.normal((3, 4096), 942).double() * 1e-3
    ((h * dh).sum() + (logits * dl).sum()).backward()
    got = bw.softmax_bwd(h.detach().float().cuda(), dh.float().cuda(), dl.float().cuda())
    close(got, logits.grad, 2e-5, "softmax bwd")


@pytest.mark.parametrize("cout", [1, 2])
def test_head_conv_bwd_vs_autograd(bw, cout):
    b, h, w = 2, 37, 70
    x = synth.normal((b, 16, h, w), 950).requires_grad_(True)
    wt = synth.normal((cout, 16, 3, 3), 951, 0.1).requires_grad_(True)
    bias = synth.normal((cout,), 952, 0.1).requires_grad_(True)
    r = F.conv2d(x, wt, bias, padding=1)
    out = F.normalize(r, p=2, dim=1) if cout == 2 else r
    dout = synth.normal(tuple(out.shape), 953)
    out.backward(dout)
    xd = nhwc(x.detach()).cuda()
    wp = wt.detach().permute(0, 2, 3, 1).contiguous().cuda()
    dr = dout.cuda()
    if cout == 2:
        dr = bw.l2norm2_bwd(r.detach().cuda(), dr)
    dx, dw, db = bw.head_conv3x3_bwd(xd, wp, dr)
    dxm, dwm, dbm = bw.head_conv3x3_bwd(xd, wp, dr, relu_mask_x=True)
    assert torch.equal(dxm, dx * (xd > 0)) and torch.equal(dwm, dw) and torch.equal(dbm, db)
    close(dx, nhwc(x.grad), 2e-4, "head dx")
    close(dw, wt.grad.permute(0, 2, 3, 1), 2e-4, "head dw")
    close(db, bias.grad, 2e-4, "head dbias")


def test_ground_descriptor_bwd(bw):
    from ccvpe_amd import ops
    cd = (8, 4, 4, 4, 2, 2)
    b, h, w, ld = 2, 5, 10, 24
    y1 = synth.normal((b, h, w, ld), 960).double().requires_grad_(True)
    wh = synth.normal((6, h), 961).double().requires_grad_(True)
    bh = synth.normal((6,), 962).double().requires_grad_(True)
    outs, off = [], 0
    for l in range(6):
        d = torch.einsum("y,byxc->bxc", wh[l], y1[..., off:off + cd[l]]) + bh[l]
        outs.append(d.reshape(b, -1))
        off += cd[l]
    out = torch.cat(outs, 1)
    dout = synth.normal(tuple(out.shape), 963).double()
    (out * dout).sum().backward()
    got = ops.ground_descriptor(y1.detach().float().cuda(), wh.detach().float().cuda(), bh.detach().float().cuda(), cd)
    close(got, out, 1e-5, "gdesc fwd")
    dy1, dwh, dbh = bw.ground_descriptor_bwd(y1.detach().float().cuda(), wh.detach().float().cuda(), cd, dout.float().cuda())
    close(dy1, y1.grad, 1e-5, "gdesc dy1")
    close(dwh, wh.grad, 1e-5, "gdesc dwh")
    close(dbh, bh.grad, 1e-5, "gdesc dbh")


def test_add_cols(bw):
    src = synth.normal((2, 3, 5, 24), 970)
    dst = synth.normal((2, 3, 5, 8), 971)
    want = dst + src[..., 12:20]
    close(bw.add_cols(src.cuda(), 12, 8, dst.clone().cuda()), want, 1e-7, "add_cols")
    close(bw.add_cols(src.cuda(), 4, 8, dst.clone().cuda(), accumulate=False), src[..., 4:12], 1e-7, "copy_cols")


@pytest.mark.parametrize("circ", [True, False])
def test_stem_wgrad(bw, circ):
    from oracle import ccvpe_oracle as orc
    x = synth.normal((2, 3, 32, 48), 980)
    wt = synth.normal((32, 3, 3, 3), 981, 0.2).requires_grad_(True)
    y = orc.same_conv(x, wt, 3, 2, 224, circ)
    dy = synth.normal(tuple(y.shape), 982)
    y.backward(dy)
    got = bw.stem_conv_wgrad(x.cuda(), nhwc(dy).cuda(), circ)
    close(got, wt.grad.permute(2, 3, 1, 0), 2e-4, "stem wgrad")


@pytest.mark.parametrize("c,L,hw,shifts,n_max,n_tail,stride", [
    (64, 64, 8, list(range(20)), 20, 20, 2),                 # level-6 style: full window, ori tail
    (64, 64, 12, list(range(20)), 20, 0, 2),
    (32, 20, 16, [-1, 0, 1], 3, 0, 4),                       # partial window (FoV < 360), ori_prior shifts
    (128, 128, 5, [0] + list(range(20)), 1, 20, 4),          # ori_prior level 6: 1 loc shift + 20 recomputed
    (16, 10, 40, list(range(16)), 16, 0, 8)])                # kitti-like, hw > 256 pixels per sample
def test_match_level_bwd_vs_autograd(bw, c, L, hw, shifts, n_max, n_tail, stride):
    from ccvpe_amd import ops
    from oracle import ccvpe_oracle as orc
    b = 2
    n = len(shifts)
    x = synth.normal((b, c, hw, hw), 990 + c).double().requires_grad_(True)
    g = synth.normal((b, L), 991).double().requires_grad_(True)
    sc = orc.rotational_matching(x, g, shifts, stride)                       # [B,n,H,W]
    mx = sc[:, :n_max].max(dim=1, keepdim=True)[0]
    xn = F.normalize(x, p=2, dim=1)
    parts = [xn, mx] + ([sc[:, n - n_tail:]] if n_tail else [])
    dst = torch.cat(parts, dim=1)                                            # our column order [X, max, tail]
    ldo = ((c + 1 + n_tail + 7) // 8) * 8
    dsc = synth.normal(tuple(sc.shape), 992).double()
    ddst = synth.normal((b, hw, hw, ldo), 993).double()
    ((sc * dsc).sum() + (nhwc(dst) * ddst[..., :dst.shape[1]]).sum()).backward()
    xd = nhwc(x.detach().float()).cuda()
    gd = g.detach().float().cuda()
    scores, dstx = ops.match_level(xd, gd, L, shifts, n_max, n_tail, stride, ldo, channels=c)
    close(scores, sc, 1e-5, "match fwd scores")
    dg = torch.zeros((b, L + 3), device="cuda")
    dx = bw.match_level_bwd(xd, gd, L, shifts, n_max, n_tail, stride, scores, dsc.float().cuda(), ddst.float().cuda(), c, dg)
    close(dx, nhwc(x.grad), 2e-4, "match dx")
    close(dg[:, :L], g.grad, 2e-4, "match dg")


def test_loss_gradients_vs_oracle_autograd():
    """ccvpe_amd.losses (the reference's losses.py API, HIP forward + backward) against autograd through the oracle."""
    from ccvpe_amd import losses
    from oracle import ccvpe_oracle as orc
    sc = synth.uniform((3, 5120), 1001, -1.0, 1.0).requires_grad_(True)
    lab = synth.uniform((3, 5120), 1002) ** 6
    (2.5 * orc.infonce_loss(sc, lab)).backward()
    scd = sc.detach().cuda().requires_grad_(True)
    lo = losses.infoNCELoss(scd, lab.cuda())
    (2.5 * lo).backward()
    close(lo, orc.infonce_loss(sc.detach(), lab), 1e-5, "infonce value")
    close(scd.grad, sc.grad, 1e-4, "infonce grad")

    lg = synth.normal((2, 65536), 1003, 2.0).requires_grad_(True)
    lb = synth.uniform((2, 65536), 1004) ** 20
    lb = lb / lb.sum(1, keepdim=True)
    orc.cross_entropy_loss(lg, lb).backward()
    lgd = lg.detach().cuda().requires_grad_(True)
    losses.cross_entropy_loss(lgd, lb.cuda()).backward()
    close(lgd.grad, lg.grad, 1e-4, "ce grad")

    ori = F.normalize(synth.normal((2, 2, 64, 64), 1005), dim=1).requires_grad_(True)
    gto = F.normalize(synth.normal((2, 2, 64, 64), 1006), dim=1)
    gt = synth.uniform((2, 1, 64, 64), 1007) ** 8
    orc.orientation_loss(ori, gto, gt).backward()
    od = ori.detach().cuda().requires_grad_(True)
    losses.orientation_loss(od, gto.cuda(), gt.cuda()).backward()
    close(od.grad, ori.grad, 1e-5, "ori grad")


def test_match_level_bwd_centred_window(bw):
    """Backward with CVM_OxfordRobotCar's centred window (odd channel offset)."""
    from ccvpe_amd import ops
    from oracle import ccvpe_oracle as orc
    b, c, L, hw, stride, woff = 2, 80, 14, 16, 4, 33
    shifts = list(range(20))
    x = synth.normal((b, c, hw, hw), 1100).double().requires_grad_(True)
    g = synth.normal((b, L), 1101).double().requires_grad_(True)
    sc = orc.rotational_matching(x, g, shifts, stride, woff)
    dst = torch.cat([F.normalize(x, p=2, dim=1), sc.max(dim=1, keepdim=True)[0]], dim=1)
    ldo = ((c + 1 + 7) // 8) * 8
    dsc = synth.normal(tuple(sc.shape), 1102).double()
    ddst = synth.normal((b, hw, hw, ldo), 1103).double()
    ((sc * dsc).sum() + (nhwc(dst) * ddst[..., :c + 1]).sum()).backward()
    xd, gd = nhwc(x.detach().float()).cuda(), g.detach().float().cuda()
    scores, _ = ops.match_level(xd, gd, L, shifts, 20, 0, stride, ldo, channels=c, window_offset=woff)
    dg = torch.zeros((b, L), device="cuda")
    dx = bw.match_level_bwd(xd, gd, L, shifts, 20, 0, stride, scores, dsc.float().cuda(), ddst.float().cuda(), c, dg,
                            window_offset=woff)
    close(dx, nhwc(x.grad), 2e-4, "centred match dx")
    close(dg, g.grad, 2e-4, "centred match dg")
