"""Train-mode forward + backward of the path, every arithmetic step a libccvpe_hip.so call.

Reference behaviour being reproduced: `model.train()` + `loss.backward()` on CVM_VIGOR / CVM_VIGOR_ori_prior /
CVM_KITTI (train_VIGOR.py:193-229, train_KITTI.py): BatchNorm with batch statistics and running-stat
updates (efficientnet_pytorch/model.py:63,73,87,182,210), drop_connect (utils.py:129-154), gradients for
every parameter the forward touches.

Design: ONE torch.autograd.Function for the whole model.  forward() runs the HIP kernels and keeps a tape of
the tensors the backward needs (raw conv outputs + batch statistics, not the normalised tensors: BN +
activation are recomputed inside the backward kernels); backward() walks the tape in reverse with the
backward kernels (ccvpe_amd/backward.py) and returns the parameter gradients in the reference's layouts.
The caller's loss stays ordinary torch code on the returned tensors (losses.py), exactly as in the
reference's training scripts.  The decoders run unfused in train mode (ConvTranspose2d as its own GEMM),
so that the upsampled tensor exists for the backward.

Determinism: no atomics anywhere; every reduction is partials + fixed-order merge.
"""
import torch

from . import backward as bw
from . import ops
from .synth import MODEL_SPECS

BN_EPS = 1e-3
SKIP_BLOCKS = (15, 10, 4, 2, 0)
# train mode: ground and aerial encoder (forward AND backward) on two HIP streams; CCVPE_TRAIN_TWO_STREAMS=0 for A/B runs
TWO_STREAMS = __import__("os").environ.get("CCVPE_TRAIN_TWO_STREAMS", "1") != "0"
# the decoders' weight gradients deferred to a third stream that runs beside the encoders' backward; =0 for A/B runs
DEFER_WGRAD = __import__("os").environ.get("CCVPE_TRAIN_DEFER_WGRAD", "1") != "0"
# localisation and orientation decoders on two streams (forward and backward); CCVPE_TRAIN_DECODER_STREAMS=0 for A/B runs
DECODER_STREAMS = TWO_STREAMS and __import__("os").environ.get("CCVPE_TRAIN_DECODER_STREAMS", "1") != "0"
# (running the critical chain on HIGH-priority streams so that the deferred weight gradients only fill idle capacity measured
# slower: 192.6 vs 173.3 ms per step — removed; so did confining them to half / a quarter of the CUs with
# hipExtStreamCreateWithCUMask: 188.9 / 186.9 vs 175.1 ms)


def _round_up(v, m):
    return (v + m - 1) // m * m


# ------------------------------------------------------------------------------------------------------
# forward
# ------------------------------------------------------------------------------------------------------
def _bn(model, live, name, x_raw, act, residual=None, dc=None, want_se=False):
    mean, var = ops.bn_stats(x_raw, live[name + ".running_mean"], live[name + ".running_var"], model.BN_MOMENTUM)
    model._nbt_pending.append(live[name + ".num_batches_tracked"])      # bumped once per forward with one foreach add
    out = ops.bn_act(x_raw, mean, var, live[name + ".weight"].detach(), live[name + ".bias"].detach(), BN_EPS, act,
                     residual=residual, dc_scale=dc, want_se=want_se)
    return out, mean, var


def encoder_forward(model, e, live, prefix, img, circular, multiscale, drop_masks, rec):
    """EfficientNet.extract_features[_multiscale] with self.training == True (model.py:278-326).
    Returns (features [B,h,w,1280], per-block outputs, tape or None)."""
    tape = {"blocks": []} if rec else None
    stem_raw = ops.stem_conv_raw(img, e.stem_w, circular)
    x, m, v = _bn(model, live, prefix + "._bn0", stem_raw, ops.ACT_SWISH)
    if rec:
        tape["stem"] = (img, stem_raw, m, v)
    feats = []
    for i, blk in enumerate(e.blocks):
        bp = "%s._blocks.%d" % (prefix, i)
        b, h, w, _ = x.shape
        s = {"x_in": x, "hw": (h, w)}
        t = x
        if blk.expand:
            e_raw = ops.conv_igemm(x, blk.cin, blk.w_exp, blk.mid, batch=b, in_h=h, in_w=w)
            t, m0, v0 = _bn(model, live, bp + "._bn0", e_raw, ops.ACT_SWISH)
            s.update(e_raw=e_raw, m0=m0, v0=v0)
        u_raw = ops.dwconv_raw(t, blk.w_dw, blk.k, blk.s, circular)
        (u, part), m1, v1 = _bn(model, live, bp + "._bn1", u_raw, ops.ACT_SWISH, want_se=True)
        ho, wo = u.shape[1], u.shape[2]
        gate = ops.se_gate(part, ho * wo, blk.se_w1, blk.se_b1, blk.se_w2, blk.se_b2)
        # the SE gate is applied inside the projection GEMM's operand load (as in eval) and, in the backward, inside the weight
        # gradient's staging (bw.conv1x1_wgrad_gated): the gated tensor u * gate is never materialised
        p_raw = ops.conv_igemm(u, blk.mid, blk.w_proj, blk.cout, batch=b, in_h=ho, in_w=wo, gate=gate)
        if rec:
            s.update(t=t, u_raw=u_raw, m1=m1, v1=v1, part=part, gate=gate, u=u)
        dc = None
        rate = model.drop_connect_rate * float(i) / len(e.blocks)               # model.py:293-295
        if blk.skip and rate:
            keep = 1.0 - rate
            if drop_masks is not None:
                mask = drop_masks[(prefix, i)].to(x.device, torch.float32)
            else:                                                                # utils.py:145-150
                mask = torch.floor(keep + torch.rand((b,), device=x.device, dtype=torch.float32))
            dc = (mask / keep).contiguous()
            model._last_drop_masks[(prefix, i)] = mask          # diagnostics / tests/test_drop_connect_gpu.py
        x, m2, v2 = _bn(model, live, bp + "._bn2", p_raw, ops.ACT_NONE, residual=x if blk.skip else None, dc=dc)
        if rec:
            s.update(p_raw=p_raw, m2=m2, v2=v2, dc=dc)
            tape["blocks"].append(s)
        if multiscale:
            feats.append(x)
    b, h, w, _ = x.shape
    h_raw = ops.conv_igemm(x, 320, e.w_head, 1280, batch=b, in_h=h, in_w=w)
    f, m, v = _bn(model, live, prefix + "._bn1", h_raw, ops.ACT_SWISH)
    if rec:
        tape["head"] = (x, h_raw, m, v)
    return f, feats, tape


def _decoder_level(lv, cat, k, k_algo, skip, batch, hw, last, cout_last, rec):
    """deconv -> [cat skip] -> 3x3 + ReLU -> 3x3 (models.py:42-47,208-209); the last level ends in the 16 -> cout head."""
    up = ops.conv_igemm(cat, k, lv.up_w, lv.up_n, batch=batch, in_h=hw, in_w=hw, shift=lv.up_b,
                        out_mode=ops.OUT_DECONV2X, algo_k=k_algo)
    y = ops.conv_igemm(up, lv.c0, lv.w_a, lv.n_a, batch=batch, in_h=2 * hw, in_w=2 * hw, kh=3, kw=3, pad=1,
                       src1=skip, c1=lv.c1, shift=lv.b_a, act=ops.ACT_RELU)
    if last:
        out = ops.head_conv3x3(y, lv.w_b, lv.b_b, cout_last, cout_last == 2)
    else:
        out = ops.conv_igemm(y, lv.n_a, lv.w_b, lv.n_b, batch=batch, in_h=2 * hw, in_w=2 * hw, kh=3, kw=3, pad=1,
                             shift=lv.b_b)
    t = {"cat": cat, "k": k, "up": up, "skip": skip, "y": y, "hw": hw} if rec else None
    return out, t


def forward_train(model, grd, sat, drop_masks=None, rec=False):
    """Train-mode forward.  Returns (outputs, tape): outputs = (logits, heatmap, x_ori, sc_1..sc_6) with the FULL
    score volume of every level (the model slices level 6 for ori_prior afterwards)."""
    spec = MODEL_SPECS[model.kind]
    n_rot = spec["n_rot"]
    from .models import MATCH_STRIDES, window_offset
    strides = MATCH_STRIDES[model.kind]
    circular = bool(model.circular_padding) and model.kind == "vigor"
    pk = model._packed()
    live = model.state_dict(keep_vars=True)
    grd = grd.contiguous().float()
    sat = sat.contiguous().float()
    batch = grd.shape[0]
    tape = {} if rec else None
    model._nbt_pending = []
    model._last_drop_masks = {}

    # The two encoders are independent until the first matching block, and in train mode they are long chains of small
    # HBM- / latency-bound launches (BatchNorm statistics and apply passes, depthwise convs, SE): the ground encoder + descriptor
    # heads run on the model's side stream next to the aerial encoder, as in the eval forward (fork / join by events).
    main = torch.cuda.current_stream()
    side = model._side_stream() if TWO_STREAMS else main
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gfeat, _, gt = encoder_forward(model, pk.grd, live, "grd_efficientnet", grd, circular, False, drop_masks, rec)
        _, gh, gw, _ = gfeat.shape
        if gh != spec["grd_h"]:
            raise ValueError("ground feature height %d != %d expected by the descriptor heads" % (gh, spec["grd_h"]))
        y1 = ops.conv_igemm(gfeat, 1280, pk.gd_w, pk.gd_n, batch=batch, in_h=gh, in_w=gw, shift=pk.gd_bias,
                            ldd=_round_up(pk.gd_n, 4))
        gdesc = ops.ground_descriptor(y1, pk.gd_wh, pk.gd_bh, spec["cd"])
    svol, sfeats, st = encoder_forward(model, pk.sat, live, "sat_efficientnet", sat, False, True, drop_masks, rec)
    main.wait_stream(side)
    if side is not main:
        gdesc.record_stream(main)                 # (read by the matching kernels on the main stream)
    model._stats_epoch = getattr(model, "_stats_epoch", 0) + 1      # the folded (eval) pack is stale now
    torch._foreach_add_(model._nbt_pending, 1)                      # num_batches_tracked += 1 for all 98 BatchNorms
    model._nbt_pending = []
    sdesc = ops.conv_igemm(svol, 1280, pk.sd_w, pk.sd_n, batch=batch, in_h=svol.shape[1], in_w=svol.shape[2],
                           kh=2, kw=2, stride=2, shift=pk.sd_bias)
    if rec:
        tape.update(grd=gt, sat=st, gfeat=gfeat, y1=y1, gdesc=gdesc, svol=svol, gw=gw, circular=circular,
                    match=[], loc=[], ori=[], pk=pk, live=live)

    loc_shifts = model._loc_shifts()
    scores_out = []
    x = sdesc
    cat6 = None
    goff = 0
    for j in range(6):
        lv = pk.loc[j]
        hw = x.shape[1]
        cd = spec["cd"][j]
        L = gw * cd
        g = gdesc[:, goff:goff + L]
        if j == 0:
            if model.ori_noise is None:
                shifts, n_max = list(range(n_rot)), n_rot
            else:                                               # models.py:501-511
                shifts, n_max = loc_shifts + list(range(n_rot)), len(loc_shifts)
            n_tail = n_rot
        else:
            shifts, n_max, n_tail = loc_shifts, len(loc_shifts), 0
        woff = window_offset(model.kind, lv.c, L)
        sc, cat = ops.match_level(x, g, L, shifts, n_max, n_tail, strides[j], lv.ldo, channels=lv.c, window_offset=woff)
        if rec:
            tape["match"].append(dict(x=x, goff=goff, L=L, shifts=shifts, n_max=n_max, n_tail=n_tail, stride=strides[j],
                                      sc=sc, c=lv.c, woff=woff))
        goff += L
        if j == 0:
            cat6 = cat
            if DECODER_STREAMS:
                # the orientation decoder needs only cat6 + the skips: it runs on the side stream beside the localisation
                # decoder (whose narrow high-resolution levels leave matrix-core time the other's wide levels can use)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    xo = _ori_decoder_forward(pk, cat6, sfeats, batch, rec, tape)
        scores_out.append(sc)
        skip = sfeats[SKIP_BLOCKS[j]] if j < 5 else None
        x, t = _decoder_level(lv, cat, lv.ldo, lv.c + 1, skip, batch, hw, j == 5, 1, rec)
        if rec:
            tape["loc"].append(t)
    logits_map = x                                                                # [B,1,512,512]
    logits = logits_map.reshape(batch, -1)                                        # models.py:319
    heatmap = ops.softmax_rows(logits).reshape(logits_map.shape)                  # models.py:320

    if DECODER_STREAMS:
        main.wait_stream(side)
        if side is not main:
            xo.record_stream(main)
    else:
        xo = _ori_decoder_forward(pk, cat6, sfeats, batch, rec, tape)
    if rec:
        tape["heatmap"] = heatmap
    return (logits, heatmap, xo) + tuple(scores_out), tape


def _ori_decoder_forward(pk, cat6, sfeats, batch, rec, tape):
    xo = cat6
    for j in range(6):
        ov = pk.ori[j]
        hw = xo.shape[1]
        skip = sfeats[SKIP_BLOCKS[j]] if j < 5 else None
        xo, t = _decoder_level(ov, xo, ov.k, ov.k_algo, skip, batch, hw, j == 5, 2, rec)
        if rec:
            tape["ori"].append(t)
    return xo


# ------------------------------------------------------------------------------------------------------
# backward
# ------------------------------------------------------------------------------------------------------
def _p(live, name):
    return live[name].detach()


def _bn_bwd(live, name, grads, x_raw, dv, mean, var, act, **kw):
    dx, dgamma, dbeta = bw.bn_act_bwd(x_raw, dv, mean, var, _p(live, name + ".weight"), _p(live, name + ".bias"), BN_EPS,
                                      act, **kw)
    grads[name + ".weight"] = dgamma
    grads[name + ".bias"] = dbeta
    return dx


def encoder_backward(e, live, prefix, tape, dfeat, dfeats, circular, grads, bwd=None):
    """dfeat: gradient w.r.t. the 1280-channel output; dfeats {block index: gradient w.r.t. that block's output}.
    bwd: {parameter name: weight in the backward layout} from the train pack (models._pack_backward)."""
    bwd = bwd or {}
    x_last, h_raw, m, v = tape["head"]
    dh = _bn_bwd(live, prefix + "._bn1", grads, h_raw, dfeat, m, v, ops.ACT_SWISH)
    grads[prefix + "._conv_head.weight"] = bw.conv_wgrad(x_last, dh, 1280, 1, 1, 1, 0)
    dx = bw.conv1x1_dgrad(dh, _p(live, prefix + "._conv_head.weight"), wp=bwd.get(prefix + "._conv_head.weight"))
    for i in reversed(range(len(e.blocks))):
        blk, s = e.blocks[i], tape["blocks"][i]
        bp = "%s._blocks.%d" % (prefix, i)
        if i in dfeats:
            bw.add_cols(dfeats[i], 0, blk.cout, dx)
        h, w = s["hw"]
        # project conv + bn2 (+ residual, drop_connect)
        dp = _bn_bwd(live, bp + "._bn2", grads, s["p_raw"], dx, s["m2"], s["v2"], ops.ACT_NONE, dc_scale=s["dc"])
        grads[bp + "._project_conv.weight"] = bw.conv1x1_wgrad_gated(s["u"], s["gate"], dp, blk.cout)
        dv = bw.conv1x1_dgrad(dp, _p(live, bp + "._project_conv.weight"), wp=bwd.get(bp + "._project_conv.weight"))
        # squeeze-excite
        g1, b1 = _p(live, bp + "._bn1.weight"), _p(live, bp + "._bn1.bias")
        # BN1 + SE backward in two passes over (u_raw, dv): five per-(sample, channel) sums first (A[0] = the gate gradient),
        # the small SE backward on them, then dbeta / dgamma finished from the sums and dx written (csrc/train_bwd.hip)
        sums = bw.se_bn_bwd_reduce(s["u_raw"], dv, s["m1"], s["v1"], g1, b1, BN_EPS, ops.ACT_SWISH)
        ho, wo = s["u_raw"].shape[1], s["u_raw"].shape[2]
        dmean, dw1, db1, dw2, db2 = bw.se_bwd(s["part"], ho * wo, sums[0].unsqueeze(1), blk.se_w1, blk.se_b1, blk.se_w2, blk.se_b2)
        grads[bp + "._se_reduce.weight"], grads[bp + "._se_reduce.bias"] = dw1, db1
        grads[bp + "._se_expand.weight"], grads[bp + "._se_expand.bias"] = dw2, db2
        # depthwise conv + bn1
        du, dg1, db1n = bw.se_bn_bwd_apply(s["u_raw"], dv, s["m1"], s["v1"], g1, b1, BN_EPS, ops.ACT_SWISH, s["gate"], dmean, sums)
        grads[bp + "._bn1.weight"], grads[bp + "._bn1.bias"] = dg1, db1n
        grads[bp + "._depthwise_conv.weight"] = bw.dwconv_wgrad(s["t"], du, blk.k, blk.s, circular).t()
        dt = bw.dwconv_dgrad(du, blk.w_dw, h, w, blk.k, blk.s, circular, w_flipped=bwd.get(bp + "._depthwise_conv.weight"))
        # expand conv + bn0
        if blk.expand:
            de = _bn_bwd(live, bp + "._bn0", grads, s["e_raw"], dt, s["m0"], s["v0"], ops.ACT_SWISH)
            grads[bp + "._expand_conv.weight"] = bw.conv_wgrad(s["x_in"], de, blk.mid, 1, 1, 1, 0)
            dxin = bw.conv1x1_dgrad(de, _p(live, bp + "._expand_conv.weight"), wp=bwd.get(bp + "._expand_conv.weight"))
        else:
            dxin = dt
        if blk.skip:
            bw.add_cols(dx, 0, blk.cin, dxin)
        dx = dxin
    img, stem_raw, m, v = tape["stem"]
    ds = _bn_bwd(live, prefix + "._bn0", grads, stem_raw, dx, m, v, ops.ACT_SWISH)
    grads[prefix + "._conv_stem.weight"] = bw.stem_conv_wgrad(img, ds, circular).permute(3, 2, 0, 1)


def _scatter_rows(dst, src, col_map):
    """dst (reference input-channel order) <- src (this implementation's order) along dim 0."""
    for d0, s0, n in col_map:
        dst[s0:s0 + n] = src[d0:d0 + n]
    return dst


def _decoder_level_backward(live, lv, t, dout, last, names, col_map, dfeats, skip_block, grads, bwd=None, defer=None):
    """names = (deconv, conv); dout: gradient w.r.t. the level's output (NHWC, or NCHW head output when last).
    Returns the gradient w.r.t. the level's input `cat` [B,hw,hw,k] in this implementation's column order."""
    deconv, conv = names
    bwd = bwd or {}
    if defer is None:                  # weight gradients right here (tests of a single level)
        def defer(fn):
            fn()
    y, up, skip, cat, k = t["y"], t["up"], t["skip"], t["cat"], t["k"]
    w_b = _p(live, conv + ".2.weight")
    if last:
        dy, dwb, dbb = bw.head_conv3x3_bwd(y, lv.w_b, dout, relu_mask_x=True)      # dy: in front of conv.1's ReLU
        grads[conv + ".2.weight"] = dwb.permute(0, 3, 1, 2)
        grads[conv + ".2.bias"] = dbb
    else:
        def wg2(y=y, dout=dout):
            grads[conv + ".2.weight"], grads[conv + ".2.bias"] = bw.conv_wgrad(y, dout, lv.n_b, 3, 3, 1, 1, want_bias=True)
        defer(wg2)
        dy = bw.conv3x3_dgrad(dout, w_b, relu_out=y, wp=bwd.get(conv + ".2.weight"))   # ReLU backward fused in the store

    def wg0(dy=dy):
        grads[conv + ".0.weight"], grads[conv + ".0.bias"] = bw.conv_wgrad(up, dy, lv.n_a, 3, 3, 1, 1, x1=skip, want_bias=True)
    defer(wg0)
    dcat2 = bw.conv3x3_dgrad(dy, _p(live, conv + ".0.weight"), wp=bwd.get(conv + ".0.weight"))   # [B,2hw,2hw,c0+c1]
    if skip is not None:
        if skip_block in dfeats:
            bw.add_cols(dcat2, lv.c0, lv.c1, dfeats[skip_block])
        else:
            d = torch.empty(skip.shape, device=skip.device, dtype=torch.float32)
            dfeats[skip_block] = bw.add_cols(dcat2, lv.c0, lv.c1, d, accumulate=False)
    # ConvTranspose2d(k2,s2): weight [Cin,Cout,2,2] (Cin in reference order), bias [Cout]
    w_ref = _p(live, deconv + ".weight")
    cout = w_ref.shape[1]

    def wgd(dcat2=dcat2):
        grads[deconv + ".bias"] = bw.bias_grad(dcat2, cout)
        dw_ours = bw.conv_wgrad(dcat2, cat, k, 2, 2, 2, 0, c0=cout)             # [k, Cout, 2, 2], our row order
        grads[deconv + ".weight"] = _scatter_rows(torch.empty_like(w_ref), dw_ours, col_map)
    defer(wgd)
    wp = bwd.get(deconv + ".weight")
    if wp is None:
        w_ours = w_ref.new_zeros((k,) + tuple(w_ref.shape[1:]))
        for d0, s0, n in col_map:
            w_ours[d0:d0 + n] = w_ref[s0:s0 + n]
        wp = bw._pack_conv(w_ours)
    b, h2 = dcat2.shape[0], dcat2.shape[1]
    return ops.conv_igemm(dcat2, cout, wp, k, batch=b, in_h=h2, in_w=h2, kh=2, kw=2, stride=2, ld0=dcat2.shape[-1])


def backward_train(model, tape, gout, on_ready=None):
    """gout = gradients w.r.t. (logits, heatmap, x_ori, sc_1..sc_6) (None where unused).
    Returns {parameter name: gradient in the parameter's own layout}.
    on_ready(grads): called three times with the dict of gradients finished so far — after the decoders / matching /
    aerial descriptor (~80 % of the parameter bytes), after the aerial encoder, and at the end — so that a
    data-parallel caller can start reducing them while the encoders' backward is still running."""
    spec = MODEL_SPECS[model.kind]
    n_rot = spec["n_rot"]
    pk, live = tape["pk"], tape["live"]          # the forward's packed weights / parameter table (nothing changed since)
    bwd = getattr(pk, "bwd", None) or {}         # backward-layout weights, built inside the per-step re-pack graph
    grads = {}
    dfeats = {}
    g_logits, g_heat, g_ori = gout[0], gout[1], gout[2]
    g_scores = gout[3:]
    heatmap = tape["heatmap"]
    batch = heatmap.shape[0]
    dev = heatmap.device

    def _c(t):
        return None if t is None else t.contiguous().float()

    # ---- localisation head: softmax over the flattened map (models.py:319-320) -----------------------
    hm = heatmap.reshape(batch, -1)
    if g_heat is not None:
        dlog = bw.softmax_bwd(hm, _c(g_heat).reshape(batch, -1), _c(g_logits))
    elif g_logits is not None:
        dlog = _c(g_logits)
    else:
        dlog = torch.zeros_like(hm)
    dloc = dlog.reshape(heatmap.shape)
    # ---- orientation head: F.normalize (models.py:341) ---------------------------------------------------
    t5 = tape["ori"][5]
    if g_ori is not None:
        raw = ops.head_conv3x3(t5["y"], pk.ori[5].w_b, pk.ori[5].b_b, 2, False)
        dori = bw.l2norm2_bwd(raw, _c(g_ori))
    else:
        dori = torch.zeros((batch, 2) + tuple(heatmap.shape[2:]), device=dev, dtype=torch.float32)

    # The decoders' WEIGHT gradients (MFMA-bound, ~20 % of the step) are off the critical path: nothing in the backward
    # reads them.  They are collected here and launched on a third stream when the encoders' backward starts — a phase of
    # HBM- / latency-bound kernels (BatchNorm, depthwise, SE) that leaves the matrix cores idle — instead of sitting between
    # the input-gradient GEMMs of the decoder chain.
    deferred = []
    defer = deferred.append if DEFER_WGRAD else None

    # ---- orientation decoder, level 1 -> 6: on the side stream beside the localisation decoder + matching (its skip
    # gradients go to a dict of its own and are added to the localisation decoder's afterwards: two streams must not
    # read-modify-write one tensor) ------------------------------------------------------------------------------------
    main0 = torch.cuda.current_stream()
    dside = model._side_stream() if DECODER_STREAMS else main0
    dfeats_ori = {} if DECODER_STREAMS else dfeats
    dside.wait_stream(main0)
    with torch.cuda.stream(dside):
        d = dori
        for j in reversed(range(6)):
            lvl = 6 - j
            ov = pk.ori[j]
            if j == 0:
                c6 = spec["ori"][0][0] - n_rot
                cmap = [(0, n_rot, c6), (c6 + 1, 0, n_rot)]
            else:
                cmap = [(0, 0, spec["ori"][j][0])]
            d = _decoder_level_backward(live, ov, tape["ori"][j], d, j == 5, ("deconv%d_ori" % lvl, "conv%d_ori" % lvl), cmap,
                                        dfeats_ori, SKIP_BLOCKS[j] if j < 5 else None, grads, bwd, defer)
        dcat6_ori = d

    # ---- localisation decoder + matching, level 1 -> 6 -------------------------------------------------
    dgdesc = torch.zeros_like(tape["gdesc"])
    d = dloc
    for j in reversed(range(6)):
        lvl = 6 - j
        lv = pk.loc[j]
        c = lv.c
        dcat = _decoder_level_backward(live, lv, tape["loc"][j], d, j == 5, ("deconv%d" % lvl, "conv%d" % lvl),
                                       [(0, 1, c), (c, 0, 1)], dfeats, SKIP_BLOCKS[j] if j < 5 else None, grads, bwd, defer)
        if j == 0:
            if DECODER_STREAMS:                # join the orientation decoder's backward
                main0.wait_stream(dside)
                dside.wait_stream(main0)
                for blk, t in dfeats_ori.items():
                    if blk in dfeats:
                        bw.add_cols(t, 0, t.shape[-1], dfeats[blk])
                    else:
                        dfeats[blk] = t
            bw.add_cols(dcat6_ori, 0, lv.ldo, dcat)
        mt = tape["match"][j]
        g = tape["gdesc"][:, mt["goff"]:mt["goff"] + mt["L"]]
        dg = dgdesc[:, mt["goff"]:mt["goff"] + mt["L"]]
        d = bw.match_level_bwd(mt["x"], g, mt["L"], mt["shifts"], mt["n_max"], mt["n_tail"], mt["stride"], mt["sc"],
                               _c(g_scores[j]), dcat, c, dg, window_offset=mt["woff"])
    dsdesc = d

    # ---- aerial descriptor: Linear(5120 -> N) == conv 2x2 stride 2 (models.py:102-104,173-184) ------------
    w_lin = _p(live, "sat_feature_to_descriptors.1.weight")
    w4 = w_lin.view(w_lin.shape[0], 1280, 2, 2)
    svol = tape["svol"]
    def wg_lin(dsdesc=dsdesc):
        dw_lin, grads["sat_feature_to_descriptors.1.bias"] = bw.conv_wgrad(svol, dsdesc, pk.sd_n, 2, 2, 2, 0, want_bias=True)
        grads["sat_feature_to_descriptors.1.weight"] = dw_lin.reshape(w_lin.shape)
    (defer or (lambda fn: fn()))(wg_lin)
    dsvol = bw.conv2x2s2_dgrad(dsdesc, w4, wp=bwd.get("sat_feature_to_descriptors.1.weight"))
    main = torch.cuda.current_stream()
    wst = model._side_stream(2) if deferred else main
    if deferred:
        wst.wait_stream(main)
        g_dec = grads
        with torch.cuda.stream(wst):
            for fn in deferred:
                fn()
        # (the closures keep the decoder's gradient tensors — allocated on the main stream, read on this one — alive until
        # the join below: freed earlier, the allocator would hand their blocks to the encoders' backward while these kernels
        # are still queued)
        grads = {}                         # the encoders' gradients are collected apart until the weight-gradient stream joins
    elif on_ready is not None:
        on_ready(grads)

    # ---- the two encoders' backward passes are independent: ground descriptors + ground encoder on the side stream, aerial
    # encoder on the main stream (both are chains of small HBM- / latency-bound launches).  Their gradients are collected in
    # separate dicts so that on_ready never sees a gradient whose kernels are still queued on the other stream.
    side = model._side_stream() if TWO_STREAMS else main
    side.wait_stream(main)
    g_grd = {}
    with torch.cuda.stream(side):
        # ---- ground descriptors (models.py:57-97,153-165) ---------------------------------------------------------
        y1, gfeat = tape["y1"], tape["gfeat"]
        dy1, dwh, dbh = bw.ground_descriptor_bwd(y1, pk.gd_wh, spec["cd"], dgdesc)
        n = pk.gd_n
        dwg = bw.conv_wgrad(gfeat, dy1, n, 1, 1, 1, 0).reshape(n, 1280)
        dbg = bw.bias_grad(dy1)
        off = 0
        for l in range(1, 7):
            p = "grd_feature_to_descriptor%d" % l
            cd = spec["cd"][l - 1]
            g_grd[p + ".0.weight"] = dwg[off:off + cd].reshape(live[p + ".0.weight"].shape)
            g_grd[p + ".0.bias"] = dbg[off:off + cd]
            g_grd[p + ".2.weight"] = dwh[l - 1].reshape(live[p + ".2.weight"].shape)
            g_grd[p + ".2.bias"] = dbh[l - 1:l].reshape(live[p + ".2.bias"].shape)
            off += cd
        b, gh, gw, ld = dy1.shape
        # dy1's pad columns (n..ld) are zero; the K dimension of the dgrad GEMM is padded to ld with zero weights
        wgd = bwd.get("grd_feature_to_descriptor.0.weight")
        if wgd is None:
            wcat = torch.cat([_p(live, "grd_feature_to_descriptor%d.0.weight" % l) for l in range(1, 7)], 0)
            wpad = wcat.new_zeros((ld,) + tuple(wcat.shape[1:]))
            wpad[:n] = wcat
            wgd = bw._pack_conv(wpad.permute(1, 0, 2, 3))
        dgfeat = ops.conv_igemm(dy1, ld, wgd, 1280, batch=b, in_h=gh, in_w=gw)
        encoder_backward(pk.grd, live, "grd_efficientnet", tape["grd"], dgfeat, {}, tape["circular"], g_grd, bwd)
    encoder_backward(pk.sat, live, "sat_efficientnet", tape["sat"], dsvol, dfeats, False, grads, bwd)
    if on_ready is not None:
        on_ready(grads)                    # (aerial encoder: gradient group 1)
    if wst is not main:
        main.wait_stream(wst)              # join the decoders' weight gradients (group 0)
        wst.wait_stream(main)
        deferred = None
        for t in g_dec.values():
            if t is not None and t.is_cuda:
                t.record_stream(main)      # allocated on the weight-gradient stream, consumed on the main stream
        grads.update(g_dec)
        if on_ready is not None:
            on_ready(grads)
    main.wait_stream(side)                 # join: the ground gradients are complete on the main stream's timeline from here
    side.wait_stream(main)                 # ... and later side-stream work cannot overtake the main stream's readers
    grads.update(g_grd)
    if on_ready is not None:
        on_ready(grads)
    return grads


class CVMFunction(torch.autograd.Function):
    """(grd, sat, *parameters) -> (logits, heatmap, x_ori, sc_1..sc_6); gradients for the parameters only."""

    @staticmethod
    def forward(ctx, model, drop_masks, names, grd, sat, *params):
        outs, tape = forward_train(model, grd, sat, drop_masks, rec=True)
        ctx.model, ctx.tape, ctx.names = model, tape, names
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *gout):
        model, tape, names = ctx.model, ctx.tape, ctx.names
        if tape is None:
            raise RuntimeError("ccvpe_amd: backward through the same forward twice is not supported")
        sync = getattr(model, "_grad_sync", None)          # harness.GradientAllReducer.attach(model)
        with torch.no_grad():
            if sync is not None and sync.arena_ok():
                # flat gradient arena: every finished group is copied into its slots and (with several ranks) all-reduced
                # in place while the rest of the backward runs; the parameters' .grad become views of the arena
                sync.begin()
                grads = backward_train(model, tape, gout, on_ready=sync.ready)
                sync.finish(grads)
            else:
                if sync is not None:
                    sync.begin_fallback()       # this step was NOT reduced in the backward: the caller's reducer() must do it
                grads = backward_train(model, tape, gout)
        live = tape["live"]
        ctx.tape = None
        out = []
        for nm in names:
            g = grads.get(nm)
            if g is not None:
                g = g.reshape(live[nm].shape).contiguous()
            out.append(g)
        return (None, None, None, None, None) + tuple(out)


def apply(model, grd, sat, drop_masks=None):
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    names = [n for n, _ in named]
    return CVMFunction.apply(model, drop_masks, names, grd, sat, *[p for _, p in named])
