"""Drop-in model classes for the CCVPE dense cross-view matching path on MI355X.

Mirrors the reference's module API for this path:
    CVM_VIGOR(device, circular_padding)                       /root/reference/models.py:49-50
    CVM_VIGOR_ori_prior(device, ori_noise, circular_padding)  /root/reference/models.py:346-347
    CVM_KITTI(device)                                         /root/reference/models.py:655-656
    forward(grd, sat) -> 9-tuple                              /root/reference/models.py:343,652,950
with the same `state_dict()` keys, shapes and order (818 tensors), so checkpoints written by
the reference's training scripts load strictly and vice versa.

The modules only HOLD parameters (reference NCHW/OIHW fp32 layout).  On the first eval forward
(and again whenever a parameter changed) they are re-packed once into the library's layouts —
BN folded to scale/shift, conv weights to [N][tap][C] K-major rows, deconv weights to a
[4*Cout][K] GEMM with the concat order this implementation uses — and every arithmetic step of
forward() is a libccvpe_hip.so call (ccvpe_amd/ops.py).  There is no eager/CPU fallback.

Eval mode is the inference path (fp32 or bf16).  Train mode (fp32) lives in ccvpe_amd/train.py: batch-statistic
BatchNorm, drop_connect and a full backward through the HIP backward kernels behind one autograd.Function.
"""
import torch
import torch.nn as nn

from . import ops
from .synth import B0_BLOCKS, MODEL_SPECS, state_dict_spec

BN_EPS = 1e-3                      # efficientnet_pytorch/utils.py:666
SKIP_BLOCKS = (15, 10, 4, 2, 0)    # models.py:167-171 (decoder levels 6..2)
MATCH_STRIDES = {"vigor": (64, 32, 16, 8, 4, 2),       # models.py:192,217,239,261,283,305
                 "kitti": (128, 64, 32, 16, 8, 8),     # models.py:794,818,841,864,887,910
                 "oxford": (64, 32, 16, 8, 4, 2)}      # models.py:1093,1116,1140,1163,1186,1209


def window_offset(kind, channels, length):
    """First channel of the matching window inside the rolled volume: 0 for VIGOR / KITTI (models.py:193), the centred
    int(C/2 - L/2) of CVM_OxfordRobotCar (models.py:1094)."""
    return int(channels / 2 - length / 2) if kind == "oxford" else 0


def _round_up(v, m):
    return (v + m - 1) // m * m


class _Holder(nn.Module):
    """Parameter container; gives the reference's dotted state_dict names."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder is not callable; use the owning CVM_* module")


def _populate(root, kind, init):
    for key, shape, role in state_dict_spec(kind):
        parts = key.split(".")
        node = root
        for name in parts[:-1]:
            child = node._modules.get(name)
            if child is None:
                child = _Holder()
                node.add_module(name, child)
            node = child
        t = init[key].clone()
        if role in ("bn_m", "bn_v", "bn_n"):
            node.register_buffer(parts[-1], t)
        else:
            node.register_parameter(parts[-1], nn.Parameter(t))


# ----------------------------------------------------------------------------------------
# weight re-packing (runs once per weight version, on the device, with torch ops)
# ----------------------------------------------------------------------------------------
def _pad2(w, dtype=torch.float32, n_mult=16):
    """Zero-pad [N,K] to N%16==0 and K to one 64-byte stage row (16 fp32 / 32 bf16), cast to dtype."""
    k_mult = 16 if dtype == torch.float32 else 32
    n, k = w.shape
    if n % n_mult == 0 and k % k_mult == 0:          # nothing to pad (most conv weights): one copy kernel instead of three
        out = w.to(dtype).contiguous()
        # always a COPY, never an alias of the live parameter: every packed weight then goes stale together when a
        # parameter is edited behind torch's version counter (see _CVMBase.invalidate)
        return out.clone() if out.data_ptr() == w.data_ptr() else out
    out = w.new_zeros((_round_up(n, n_mult), _round_up(k, k_mult)))
    out[:n, :k] = w
    return out.to(dtype).contiguous()


def _pack_conv(w, dtype=torch.float32):
    """OIHW -> [O][kh][kw][I] rows, zero-padded (N to 16, K to a 64-byte multiple)."""
    o = w.shape[0]
    return _pad2(w.permute(0, 2, 3, 1).reshape(o, -1), dtype)


def _fold_bn(sd, p):
    scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return scale.contiguous(), shift.contiguous()


class _Obj(object):
    pass


def _pack_encoder(sd, p, dtype=torch.float32, fold_bn=True):
    """fold_bn=False (train mode): BatchNorm uses batch statistics, the folded scale/shift would be ~800 tiny launches
    per step for nothing."""
    fold = _fold_bn if fold_bn else (lambda _sd, _p: (None, None))
    e = _Obj()
    e.stem_w = sd[p + "._conv_stem.weight"].permute(2, 3, 1, 0).contiguous()      # [ky][kx][ci][co]
    e.stem_scale, e.stem_shift = fold(sd, p + "._bn0")
    e.blocks = []
    for i, (k, s, ex, cin, cout) in enumerate(B0_BLOCKS):
        b = _Obj()
        bp = "%s._blocks.%d" % (p, i)
        b.k, b.s, b.expand, b.cin, b.cout, b.mid = k, s, ex != 1, cin, cout, cin * ex
        if b.expand:
            b.w_exp = _pack_conv(sd[bp + "._expand_conv.weight"], dtype)
            b.s0, b.b0 = fold(sd, bp + "._bn0")
        b.w_dw = sd[bp + "._depthwise_conv.weight"].reshape(b.mid, k, k).permute(1, 2, 0).contiguous()
        b.s1, b.b1 = fold(sd, bp + "._bn1")
        b.se_w1 = sd[bp + "._se_reduce.weight"].reshape(-1, b.mid).contiguous()
        b.se_b1 = sd[bp + "._se_reduce.bias"].contiguous()
        b.se_w2 = sd[bp + "._se_expand.weight"].reshape(b.mid, -1).t().contiguous()      # [Cs][C]
        b.se_b2 = sd[bp + "._se_expand.bias"].contiguous()
        b.w_proj = _pack_conv(sd[bp + "._project_conv.weight"], dtype)
        b.s2, b.b2 = fold(sd, bp + "._bn2")
        b.skip = (s == 1 and cin == cout)                                          # model.py:126
        e.blocks.append(b)
    e.w_head = _pack_conv(sd[p + "._conv_head.weight"], dtype)
    e.head_scale, e.head_shift = fold(sd, p + "._bn1")
    return e


def _pack_deconv(w, bias, col_map, ldo, dtype=torch.float32):
    """ConvTranspose2d weight [Cin,Cout,2,2] -> GEMM rows n=(dy*2+dx)*Cout+co over this
    implementation's K order.  col_map: list of (dst_start, src_start, length)."""
    cin, cout = w.shape[0], w.shape[1]
    wg = w.permute(2, 3, 1, 0).reshape(4 * cout, cin)
    out = w.new_zeros((4 * cout, ldo))
    for d0, s0, n in col_map:
        out[:, d0:d0 + n] = wg[:, s0:s0 + n]
    return _pad2(out, dtype), bias.repeat(4).contiguous()


# which (3x3 tap k, deconv tap a) pairs land on low-res offset d for output parity p:  _UP_S[p][d]
_UP_S = {0: {0: ((0, 1),), 1: ((1, 0), (2, 1))}, 1: {0: ((0, 0), (1, 1)), 1: ((2, 0),)}}
_UP_R = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}      # 3x3 taps inside the image for border class 0/1/2
# Decoder levels (index j = 0..5 <-> level 6..1) whose ConvTranspose2d is folded into the following 3x3
# conv (csrc/conv_igemm.hip: upconv kernels).  Measured at B=64 fp32 (ms/step): none 46.8, levels 6-3 42.6,
# 6-2 41.95, all six 41.7 -> all levels by default (CCVPE_FOLD_LEVELS overrides, e.g. "" to disable).
# orientation decoder on the side stream next to the localisation decoder: "auto" (default) = in bf16 storage only, where the
# decoder kernels are short and far from the MFMA roof (C2, B = 32: 8.48 -> 7.80 ms; C1 model bf16 13.18 -> 13.05); in fp32
# both decoders are MFMA-bound (35.59 -> 35.34 ms) and stay serial so that per-kernel timings are those of a kernel alone
_OVERLAP_ENV = __import__("os").environ.get("CCVPE_OVERLAP_DECODERS", "auto")


# Round 6, same-box A/B (bf16, ms per forward, decoders on two streams / on one): B = 32 (C2) 4.97 / 5.25, B = 64 (C1) 8.47 / 8.54, but
# B = 256 (C4, hipGraph replay) 34.5 / 32.8 and eager 34.1 / 32.8 — at that size every decoder kernel fills the chip by itself and
# two of them only fight over L2 / the Infinity Cache.  (The two ENCODERS on two streams pay at every size measured.)
OVERLAP_DECODERS_MAX_BATCH = 128


def _overlap_decoders(precision, batch=1):
    return _OVERLAP_ENV == "1" or (_OVERLAP_ENV == "auto" and precision == "bf16" and batch <= OVERLAP_DECODERS_MAX_BATCH)
FOLD_LEVELS = tuple(int(c) for c in __import__("os").environ.get("CCVPE_FOLD_LEVELS", "012345"))
# Below this many low-res pixels (batch * h * w) the folded GEMM has too few output tiles to fill the chip and walks
# K = 4*c0 + 9*c1 serially (B = 8, level 6: 2 x 1.05 ms at 20 TF/s); the unfused pair goes through the split-K igemm
# instead.  Round 4 sweep (CVM_VIGOR, ms per forward at thresholds 1024 / 2048 / 4096): B = 32 bf16 6.53 / 6.49 / 6.69, fp32 18.97 /
# 19.02 / 19.92; B = 16 bf16 4.03 / 4.02 / 4.02, fp32 11.40 / 11.25 / 11.24 -> 2048 (B >= 32 folds level 6, B <= 16 does not).
FOLD_MIN_PIXELS = int(__import__("os").environ.get("CCVPE_FOLD_MIN_PIXELS", "2048"))
# level 1 (512 x 512) of both decoders as ONE launch (csrc/tail512.hip); CCVPE_FUSE_TAIL=0 restores upconv + head conv (A/B runs)
FUSE_TAIL = __import__("os").environ.get("CCVPE_FUSE_TAIL", "1") != "0"
# bf16 storage path: the fp32 tail's matrix arithmetic on bf16 hi + lo planes (csrc/tail512.hip SPLIT); CCVPE_SPLIT_TAIL=0 = exact fp32
SPLIT_TAIL = __import__("os").environ.get("CCVPE_SPLIT_TAIL", "1") != "0"
# train mode: replay the per-step weight re-pack as one hipGraph (see _CVMBase._packed); CCVPE_PACK_GRAPH=0 keeps it eager
PACK_GRAPH = __import__("os").environ.get("CCVPE_PACK_GRAPH", "1") != "0"
# train mode, fp32: the per-step re-pack as ONE gather launch (ccvpe_amd/repack.py); CCVPE_PACK_GATHER=0 keeps the graph replay
PACK_GATHER = __import__("os").environ.get("CCVPE_PACK_GATHER", "1") != "0"
# one-hypothesis matching of level j + 1 inside level j's last conv (bf16 narrow levels); CCVPE_FUSE_MATCH=0 = separate launches (A/B runs)
FUSE_MATCH = __import__("os").environ.get("CCVPE_FUSE_MATCH", "1") != "0"
# stem conv + block-0 depthwise conv as one launch (csrc/stem_dw.hip); CCVPE_FUSE_STEM=0 = the two unfused launches (A/B runs)
FUSE_STEM = __import__("os").environ.get("CCVPE_FUSE_STEM", "1") != "0"
# eval forward: CCVPE_EVAL_TWO_STREAMS=0 runs the ground encoder on the main stream too (for per-kernel profiles in which no
# two kernels share the chip; the default overlaps the two encoders)
EVAL_TWO_STREAMS = __import__("os").environ.get("CCVPE_EVAL_TWO_STREAMS", "1") != "0"


def _pack_upconv(wd, bd, col_map, cp, w3, b3, dtype=torch.float32):
    """Effective weights of  conv3x3(cat[deconv2x2s2(x), skip])  per output parity.

    wd [Cin_ref,Cd,2,2], bd [Cd]: the ConvTranspose2d; col_map re-orders its input channels into this
    implementation's concat order (cp channels incl. padding); w3 [Co,Cd+C1,3,3], b3 [Co]: the conv.
    Returns (w [4][Co^16][K^], shift9 [9][Co]) with K = 4*cp + 9*C1; products are accumulated in fp64."""
    cd, co = wd.shape[1], w3.shape[0]
    c1 = w3.shape[1] - cd
    wd_my = wd.new_zeros((cp, cd, 2, 2), dtype=torch.float64)
    for d0, s0, n in col_map:
        wd_my[d0:d0 + n] = wd[s0:s0 + n].double()
    w3d = w3[:, :cd].double()
    k = 4 * cp + 9 * c1
    kmult = 16 if dtype == torch.float32 else 32
    out = wd.new_zeros((4, _round_up(co, 16), _round_up(k, kmult)), dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            par = py * 2 + px
            for du in range(2):
                for dv in range(2):
                    weff = wd.new_zeros((co, cp), dtype=torch.float64)
                    for ky, a in _UP_S[py][du]:
                        for kx, b in _UP_S[px][dv]:
                            weff += w3d[:, :, ky, kx] @ wd_my[:, :, a, b].t()
                    tap = du * 2 + dv
                    out[par, :co, tap * cp:(tap + 1) * cp] = weff
            if c1:
                out[par, :co, 4 * cp:k] = w3[:, cd:].double().permute(0, 2, 3, 1).reshape(co, 9 * c1)
    shift9 = wd.new_zeros((9, co), dtype=torch.float64)
    for rc in range(3):
        for cc in range(3):
            v = b3.double().clone()
            for ky in _UP_R[rc]:
                for kx in _UP_R[cc]:
                    v += w3d[:, :, ky, kx] @ bd.double()
            shift9[rc * 3 + cc] = v
    return out.to(dtype).contiguous(), shift9.float().contiguous()


def _pack_backward(sd, kind, pk):
    """Train mode: the weights in the layouts the BACKWARD GEMMs read (ccvpe_amd/backward.py: 1x1 -> W^T; 3x3 -> taps flipped,
    in / out swapped; ConvTranspose2d <-> conv 2x2 stride 2; stride-1 depthwise -> taps reversed), keyed by parameter name.
    Built together with the forward pack, i.e. inside the ONE hipGraph that replays the per-step re-pack: the backward used to
    re-lay them out with ~250 eager torch launches in the middle of its kernel stream."""
    spec = MODEL_SPECS[kind]
    n_rot = spec["n_rot"]
    bwd = {}
    for enc in ("grd_efficientnet", "sat_efficientnet"):
        for i, (k, s_, ex, cin, cout) in enumerate(B0_BLOCKS):
            bp = "%s._blocks.%d" % (enc, i)
            if ex != 1:
                bwd[bp + "._expand_conv.weight"] = _pack_conv(sd[bp + "._expand_conv.weight"].permute(1, 0, 2, 3))
            bwd[bp + "._project_conv.weight"] = _pack_conv(sd[bp + "._project_conv.weight"].permute(1, 0, 2, 3))
            if s_ == 1:
                mid = cin * ex
                w_dw = sd[bp + "._depthwise_conv.weight"].reshape(mid, k * k).t()            # [k*k, C] as the forward packs it
                bwd[bp + "._depthwise_conv.weight"] = w_dw.flip(0).contiguous()
        bwd[enc + "._conv_head.weight"] = _pack_conv(sd[enc + "._conv_head.weight"].permute(1, 0, 2, 3))
    for j in range(6):
        lvl = 6 - j
        for branch, lv in (("", pk.loc[j]), ("_ori", pk.ori[j])):
            conv, deconv = "conv%d%s" % (lvl, branch), "deconv%d%s" % (lvl, branch)
            if lvl != 1:
                bwd[conv + ".2.weight"] = _pack_conv(sd[conv + ".2.weight"].flip(2, 3).permute(1, 0, 2, 3))
            bwd[conv + ".0.weight"] = _pack_conv(sd[conv + ".0.weight"].flip(2, 3).permute(1, 0, 2, 3))
            # ConvTranspose2d weight [Cin_ref, Cout, 2, 2] -> rows in this implementation's input-channel order, as a conv 2x2 s2
            w_ref = sd[deconv + ".weight"]
            if branch == "":
                c = lv.c
                k_rows, cmap = lv.ldo, [(0, 1, c), (c, 0, 1)]
            elif j == 0:
                c6 = spec["ori"][0][0] - n_rot
                k_rows, cmap = lv.k, [(0, n_rot, c6), (c6 + 1, 0, n_rot)]
            else:
                k_rows, cmap = lv.k, [(0, 0, spec["ori"][j][0])]
            w_ours = w_ref.new_zeros((k_rows,) + tuple(w_ref.shape[1:]))
            for d0, s0, n in cmap:
                w_ours[d0:d0 + n] = w_ref[s0:s0 + n]
            bwd[deconv + ".weight"] = _pack_conv(w_ours)
    # aerial descriptor Linear == conv 2x2 s2: its input gradient is the pixel-shuffle GEMM
    w = sd["sat_feature_to_descriptors.1.weight"]
    w4 = w.view(w.shape[0], 1280, 2, 2)
    bwd["sat_feature_to_descriptors.1.weight"] = _pack_deconv(w4, w4.new_zeros((1280,)), [(0, 0, w.shape[0])], w.shape[0])[0]
    # six ground-descriptor 1x1 convs as one GEMM over the padded column count of the fused forward output
    wcat = torch.cat([sd["grd_feature_to_descriptor%d.0.weight" % l] for l in range(1, 7)], 0)
    ld = _round_up(pk.gd_n, 4)
    wpad = wcat.new_zeros((ld,) + tuple(wcat.shape[1:]))
    wpad[:pk.gd_n] = wcat
    bwd["grd_feature_to_descriptor.0.weight"] = _pack_conv(wpad.permute(1, 0, 2, 3))
    return bwd


def _pack_model(sd, kind, n_tail, dtype=torch.float32, fold=True, f32_tail=0):
    """f32_tail (bf16 packs only): the last `f32_tail` levels of the LOCALISATION decoder (and the heat-map head) get fp32
    weights — the forward widens the activations entering them, so the arg-max of the heat-map is decided in fp32."""
    spec = MODEL_SPECS[kind]
    pk = _Obj()
    pk.dtype = dtype
    pk.f32_from = 6 - f32_tail if dtype != torch.float32 else 6      # first loc level index (0..5) that runs in fp32
    base_dtype = dtype
    pk.grd = _pack_encoder(sd, "grd_efficientnet", dtype, fold_bn=fold)
    pk.sat = _pack_encoder(sd, "sat_efficientnet", dtype, fold_bn=fold)
    # six ground-descriptor heads fused into one 1x1 GEMM (N = sum Cd) + height collapse
    ws, bs, wh, bh = [], [], [], []
    for l in range(1, 7):
        p = "grd_feature_to_descriptor%d" % l
        ws.append(sd[p + ".0.weight"].reshape(-1, 1280))
        bs.append(sd[p + ".0.bias"])
        wh.append(sd[p + ".2.weight"].reshape(-1))
        bh.append(sd[p + ".2.bias"].reshape(-1))
    pk.gd_w = _pad2(torch.cat(ws, 0), dtype)
    pk.gd_bias = torch.cat(bs, 0).contiguous()
    pk.gd_n = pk.gd_bias.shape[0]
    pk.gd_wh = torch.stack(wh, 0).contiguous()
    pk.gd_bh = torch.cat(bh, 0).contiguous()
    # aerial descriptor Linear(5120 -> N) == conv 2x2 stride 2 (models.py:102-104,173-184)
    w = sd["sat_feature_to_descriptors.1.weight"]
    pk.sd_w = _pack_conv(w.view(w.shape[0], 1280, 2, 2), dtype)
    pk.sd_bias = sd["sat_feature_to_descriptors.1.bias"].contiguous()
    pk.sd_n = w.shape[0]

    pk.loc, pk.ori = [], []
    n_rot = spec["n_rot"]
    for j in range(6):
        lvl = 6 - j
        # ---- loc branch -------------------------------------------------------------------
        dc_in, dc_out, c_in, c_out = spec["loc"][j]
        dtype = torch.float32 if j >= pk.f32_from else base_dtype
        c = dc_in - 1                                  # feature channels entering the level
        ldo = _round_up(c + 1 + (n_tail if j == 0 else 0), 8)
        lv = _Obj()
        lv.c, lv.ldo, lv.up_n = c, ldo, 4 * dc_out
        # reference concat order is [max, X]; ours is [X, max, (tail), pad]
        lv.up_w, lv.up_b = _pack_deconv(sd["deconv%d.weight" % lvl], sd["deconv%d.bias" % lvl],
                                        [(0, 1, c), (c, 0, 1)], ldo, dtype)
        lv.c0, lv.c1 = dc_out, c_in - dc_out
        lv.n_a = c_out
        if fold and j in FOLD_LEVELS:
            lv.fw, lv.fshift = _pack_upconv(sd["deconv%d.weight" % lvl], sd["deconv%d.bias" % lvl],
                                            [(0, 1, c), (c, 0, 1)], ldo, sd["conv%d.0.weight" % lvl],
                                            sd["conv%d.0.bias" % lvl], dtype)
        lv.w_a = _pack_conv(sd["conv%d.0.weight" % lvl], dtype)
        lv.b_a = sd["conv%d.0.bias" % lvl].contiguous()
        if lvl != 1:
            lv.w_b = _pack_conv(sd["conv%d.2.weight" % lvl], dtype)
            lv.n_b = c_out
        else:
            lv.w_b = sd["conv1.2.weight"].permute(0, 2, 3, 1).contiguous()      # [1][3][3][16]
        lv.b_b = sd["conv%d.2.bias" % lvl].contiguous()
        pk.loc.append(lv)
        # ---- ori branch -------------------------------------------------------------------
        dtype = base_dtype
        dc_in, dc_out, c_in, c_out = spec["ori"][j]
        ov = _Obj()
        if j == 0:
            c6 = dc_in - n_rot
            ov.k = pk.loc[0].ldo
            ov.k_algo = dc_in
            # reference order [scores(n_rot), X]; ours [X, max, scores, pad]
            ov.up_w, ov.up_b = _pack_deconv(sd["deconv6_ori.weight"], sd["deconv6_ori.bias"],
                                            [(0, n_rot, c6), (c6 + 1, 0, n_rot)], pk.loc[0].ldo, dtype)
        else:
            ov.k = ov.k_algo = dc_in
            ov.up_w, ov.up_b = _pack_deconv(sd["deconv%d_ori.weight" % lvl], sd["deconv%d_ori.bias" % lvl],
                                            [(0, 0, dc_in)], dc_in, dtype)
        ov.up_n = 4 * dc_out
        ov.c0, ov.c1 = dc_out, c_in - dc_out
        if fold and j in FOLD_LEVELS:
            cmap = [(0, n_rot, c6), (c6 + 1, 0, n_rot)] if j == 0 else [(0, 0, dc_in)]
            ov.fw, ov.fshift = _pack_upconv(sd["deconv%d_ori.weight" % lvl], sd["deconv%d_ori.bias" % lvl], cmap,
                                            ov.k, sd["conv%d_ori.0.weight" % lvl], sd["conv%d_ori.0.bias" % lvl],
                                            dtype)
        ov.w_a = _pack_conv(sd["conv%d_ori.0.weight" % lvl], dtype)
        ov.b_a = sd["conv%d_ori.0.bias" % lvl].contiguous()
        ov.n_a = c_out
        if lvl != 1:
            ov.w_b = _pack_conv(sd["conv%d_ori.2.weight" % lvl], dtype)
            ov.n_b = c_out
        else:
            ov.w_b = sd["conv1_ori.2.weight"].permute(0, 2, 3, 1).contiguous()  # [2][3][3][16]
        ov.b_b = sd["conv%d_ori.2.bias" % lvl].contiguous()
        pk.ori.append(ov)
    pk.bwd = _pack_backward(sd, kind, pk) if (not fold and base_dtype == torch.float32) else None
    return pk


# ----------------------------------------------------------------------------------------
# forward building blocks (every line below enqueues HIP kernels through the C ABI)
# ----------------------------------------------------------------------------------------
def _run_encoder(e, img, circular, multiscale, dtype=torch.float32):
    """EfficientNet.extract_features[_multiscale] (efficientnet_pytorch/model.py:278-326)."""
    blk0 = e.blocks[0]
    stem_fused = (FUSE_STEM and not blk0.expand and blk0.k == 3 and blk0.s == 1 and blk0.mid == 32
                  and ops.stem_dw_supported(img.shape[2], img.shape[3], circular) > 0)
    x = None if stem_fused else ops.stem_conv(img, e.stem_w, e.stem_scale, e.stem_shift, circular, out_dtype=dtype)
    feats = []
    for blk in e.blocks:
        if x is not None:
            b, h, w, _ = x.shape
        if x is None:
            # block 0 has no expand conv: the stem tensor only feeds its depthwise conv and stays in LDS (csrc/stem_dw.hip)
            u, part = ops.stem_dw(img, e.stem_w, e.stem_scale, e.stem_shift, blk.w_dw, blk.s1, blk.b1, circular,
                                  out_dtype=dtype)
            b = u.shape[0]
        elif blk.expand and ops.mbconv_front_supported(h, w, blk.cin, blk.mid, blk.k, blk.s):
            # early blocks: the 6x-expanded tensor stays in LDS (csrc/mbconv_front.hip)
            u, part = ops.mbconv_front(x, blk.w_exp, blk.s0, blk.b0, blk.w_dw, blk.s1, blk.b1, blk.mid,
                                       blk.k, blk.s, circular)
        else:
            t = x
            if blk.expand:
                t = ops.conv_igemm(x, blk.cin, blk.w_exp, blk.mid, batch=b, in_h=h, in_w=w,
                                   scale=blk.s0, shift=blk.b0, act=ops.ACT_SWISH)
            u, part = ops.dwconv(t, blk.w_dw, blk.s1, blk.b1, blk.k, blk.s, circular)
        ho, wo = u.shape[1], u.shape[2]
        gate = ops.se_gate(part, ho * wo, blk.se_w1, blk.se_b1, blk.se_w2, blk.se_b2)
        x = ops.conv_igemm(u, blk.mid, blk.w_proj, blk.cout, batch=b, in_h=ho, in_w=wo, gate=gate,
                           scale=blk.s2, shift=blk.b2, residual=x if blk.skip else None)
        if multiscale:
            feats.append(x)
    b, h, w, _ = x.shape
    f = ops.conv_igemm(x, 320, e.w_head, 1280, batch=b, in_h=h, in_w=w, scale=e.head_scale,
                       shift=e.head_shift, act=ops.ACT_SWISH)
    return f, feats


def _double_conv(lv, up, skip, batch, hw):
    """deconv output (+ skip, read as a second source: the cat is never materialised) ->
    3x3 + ReLU -> 3x3   (models.py:42-47,208-209)."""
    y = ops.conv_igemm(up, lv.c0, lv.w_a, lv.n_a, batch=batch, in_h=hw, in_w=hw, kh=3, kw=3, pad=1,
                       src1=skip, c1=lv.c1, shift=lv.b_a, act=ops.ACT_RELU)
    return y


class _CVMBase(nn.Module):
    kind = "vigor"

    def __init__(self, device, circular_padding, ori_noise=None, efficientnet_weights=None):
        super().__init__()
        self.device = device                     # stored, never used (models.py:52)
        self.circular_padding = circular_padding
        self.ori_noise = ori_noise
        # Parameters start from the reference's construction-time distributions (torch default Conv2d / ConvTranspose2d /
        # Linear / BatchNorm initialisation: ccvpe_amd/init.py, models.py:55-148).  The reference then loads ImageNet
        # EfficientNet-B0 weights from the network (utils.py:747): here that is a LOCAL checkpoint, given as
        # `efficientnet_weights=` (path or state_dict, lukemelas key names) or through CCVPE_EFFICIENTNET_B0.
        from . import init as _init
        _populate(self, self.kind, _init.reference_init_state_dict(self.kind))
        weights = efficientnet_weights if efficientnet_weights is not None else __import__("os").environ.get("CCVPE_EFFICIENTNET_B0")
        if weights:
            _init.load_efficientnet_b0(self, weights)
        self._pack_cache = None
        self._pack_key = None
        self.precision = "fp32"
        self.fp32_tail_levels = 1

    def set_precision(self, precision, fp32_tail_levels=1):
        """'fp32' (default: exact fp32 everywhere) or 'bf16' (BASELINE C2/C4: bf16 NHWC activations and
        packed weights, fp32 accumulation/BN/SE/softmax; scores, logits, heat-map and orientation are
        returned in fp32 either way).
        fp32_tail_levels (bf16 only, default 1): the last levels of the localisation decoder — level 1 and the heat-map
        head by default — run through the fp32 kernels on widened activations (the arg-max of the
        heat-map, models.py:319-320, is a logits-ORDERING question; bf16 storage of the 512^2 tensors that directly form the
        logits reorders near-ties).  Measured over 64 seeded pairs against the fp32 path (tools/bf16_argmax_rate.py, CVM_VIGOR
        N_rot = 20; fp32 top-1/top-2 margins down to 1.5e-3 of the logit range): tail 0 -> 63/64 arg-max pixels equal (logit
        error 5.3e-3 of range), tail 1 -> 64/64 (4.5e-3), tail 2 -> 64/64 (3.5e-3), tail 3 -> 64/64 (3.4e-3): the floor
        comes from the bf16 encoders / upper decoder levels, so more than one fp32 level buys little for ~2.5 ms per level
        at B = 64.  0 = bf16 storage everywhere (fastest; the arg-max may move to a near-tie pixel)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if not 0 <= int(fp32_tail_levels) <= 5:
            raise ValueError("fp32_tail_levels must be in 0..5 (level 6 feeds the bf16 orientation decoder)")
        self.precision = precision
        self.fp32_tail_levels = int(fp32_tail_levels)
        return self

    # -- weight version tracking -------------------------------------------------------------
    def invalidate(self):
        """Drop the packed (re-laid-out, BN-folded, bf16) weights so that the next forward re-packs them from the live
        parameters.  Needed only after editing parameters or buffers through a path that does not bump torch's version
        counter — `p.data.copy_()/mul_()` (EMA, manual init): optimizer steps, load_state_dict(), in-place ops on the
        parameter itself and ccvpe_amd.optim.Adam are detected automatically (_weights_key)."""
        self._pack_cache = None
        self._pack_key = None
        object.__setattr__(self, "_flat_tensors", None)
        return self

    def _weights_key(self):
        """(storage, version) of every parameter / buffer: what the packed weights were derived from.  The module tree is
        walked ONCE (0.86 ms per forward otherwise: 7 % of a B = 8 forward); `to()` / `load_state_dict` swap or rewrite the
        tensors, which the per-tensor (data_ptr, _version) pairs of the cached list catch — a parameter OBJECT that is replaced
        (`module.weight = nn.Parameter(...)`) needs invalidate()."""
        from . import _lib
        flat = getattr(self, "_flat_tensors", None)
        if flat is None:
            flat = list(self.parameters()) + list(self.buffers())
            object.__setattr__(self, "_flat_tensors", flat)
        return (_lib.weights_epoch,) + tuple((t.data_ptr(), t._version) for t in flat)

    def _packed(self):
        # running statistics are updated in place by the HIP kernels (no torch version bump): the eval pack (BN folded
        # from them) additionally depends on how many train-mode forwards have run; the train pack does not use them
        epoch = 0 if self.training else getattr(self, "_stats_epoch", 0)
        key = (self.precision, self.fp32_tail_levels, self.training, epoch) + self._weights_key()
        if self._pack_cache is not None and key == self._pack_key:
            return self._pack_cache
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("ccvpe_amd: move the model to the MI355X first (.to('cuda'))")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        n_tail = MODEL_SPECS[self.kind]["n_rot"]
        dtype = torch.float32 if self.precision == "fp32" else torch.bfloat16

        def pack(src=sd):
            with torch.no_grad():
                # train mode runs the decoders unfused (ccvpe_amd/train.py): skip the fp64 fold of deconv into conv
                return _pack_model(src, self.kind, n_tail, dtype, fold=not self.training, f32_tail=self.fp32_tail_levels)

        if self.training and PACK_GATHER and dtype == torch.float32 and not getattr(self, "_pack_plan_failed", False):
            # The train-mode pack is pure data movement, so it is ONE gather launch per step (repack.py: the plan is derived by
            # running the pack code on index-carrying tensors and verified bit for bit against an eager pack when it is built).
            from . import repack
            where = tuple(t.data_ptr() for t in list(self.parameters()) + list(self.buffers()))
            gp = getattr(self, "_pack_plan", None)
            if gp is not None and gp[0] == where:
                self._pack_cache = gp[1].run()
            else:
                plan = repack.build(sd, pack)
                if plan is None:
                    self._pack_plan, self._pack_plan_failed = None, True
                    return self._packed()
                self._pack_plan = (where, plan)
                self._pack_cache = plan.pk
        elif self.training and PACK_GRAPH:
            # Every optimizer step changes every weight, so a training step re-packs all of them: ~280 tiny torch launches
            # (permute / pad / copy), 8.8 ms of wall time of which 1.9 ms is kernel time.  The launches read the live
            # parameters in place and never change shape, so from the second pack on (the first one runs eagerly and
            # warms the allocator) they are one hipGraph launch, re-captured only if a parameter's storage moves.
            where = (self.precision, self.fp32_tail_levels) + tuple(
                t.data_ptr() for t in list(self.parameters()) + list(self.buffers()))
            g = getattr(self, "_pack_graph", None)
            if g is not None and g[0] == where:
                g[1].replay()
                self._pack_cache = g[2]
            elif getattr(self, "_pack_where", None) == where and not getattr(self, "_pack_graph_failed", False):
                # thread_local capture mode: the RCCL watchdog thread of a data-parallel job keeps polling events while this
                # thread captures; a capture that fails for any reason leaves the eager path in charge for good
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        pk = pack()
                    graph.replay()
                    self._pack_graph = (where, graph, pk)
                    self._pack_cache = pk
                except Exception:                                   # noqa: BLE001
                    self._pack_graph, self._pack_graph_failed = None, True
                    self._pack_cache = pack()
            else:
                self._pack_graph = None
                self._pack_where = where
                self._pack_cache = pack()
        else:
            self._pack_cache = pack()
        self._pack_key = key
        return self._pack_cache

    def _side_stream(self, which=1):
        """Extra HIP streams of this model: 1 = the ground encoder (eval and train), 2 = the decoders' weight gradients in
        the training backward."""
        dev = next(self.parameters()).device
        streams = self.__dict__.setdefault("_side_streams", {})
        st = streams.get(which)
        if st is None or st.device != dev:
            st = torch.cuda.Stream(device=dev)
            streams[which] = st
        return st

    def _ori_decoder(self, pk, cat6, sfeats, batch):
        """Orientation decoder (models.py:323-341): six (deconv -> cat skip -> double_conv) levels and the
        final 3x3 conv to (cos, sin) + F.normalize."""
        xo = cat6
        # test hook (tests/test_bf16_gpu.py, tools/ori_norm_scan.py): the UN-normalised conv1_ori output, i.e. models.py:341's input
        normalize = not getattr(self, "ori_raw_output", False)
        for j in range(6):
            ov = pk.ori[j]
            hw = xo.shape[1]
            skip = sfeats[SKIP_BLOCKS[j]] if j < 5 else None
            if j == 5 and FUSE_TAIL and j in FOLD_LEVELS and ops.tail512_ok(hw, hw, ov.n_a):
                # the whole 512 x 512 level in one launch: deconv1_ori + conv1_ori + F.normalize (models.py:145-148,341)
                return ops.tail512(xo, ov.k, ov.fw, ov.fshift, ov.w_b, ov.b_b, 2, normalize, batch=batch, h1=hw, w1=hw)
            if j in FOLD_LEVELS and batch * hw * hw >= FOLD_MIN_PIXELS:
                y = ops.upconv3x3(xo, ov.k, ov.fw, ov.fshift, ov.n_a, batch=batch, h1=hw, w1=hw,
                                  src1=skip, c1=ov.c1, act=ops.ACT_RELU)
            else:
                up = ops.conv_igemm(xo, ov.k, ov.up_w, ov.up_n, batch=batch, in_h=hw, in_w=hw,
                                    shift=ov.up_b, out_mode=ops.OUT_DECONV2X, algo_k=ov.k_algo)
                y = _double_conv(ov, up, skip, batch, 2 * hw)
            if j < 5:
                xo = ops.conv_igemm(y, ov.n_a, ov.w_b, ov.n_b, batch=batch, in_h=2 * hw, in_w=2 * hw,
                                    kh=3, kw=3, pad=1, shift=ov.b_b)
            else:
                return ops.head_conv3x3(y, ov.w_b, ov.b_b, 2, normalize)       # + F.normalize (:341)

    def _loc_shifts(self):
        n_rot = MODEL_SPECS[self.kind]["n_rot"]
        if self.ori_noise is None:
            return list(range(n_rot))                                   # models.py:191
        k = int(self.ori_noise / 18)                                    # models.py:489
        return list(range(-k, k + 1))

    # -- train mode ------------------------------------------------------------------------------
    drop_connect_rate = 0.2          # efficientnet_pytorch/utils.py:639 (GlobalParams default)
    BN_MOMENTUM = 0.01               # model.py:52

    def forward(self, grd, sat, drop_masks=None):
        """Eval mode: the full inference path.  Train mode (self.training): the reference's training semantics
        (batch-statistic BatchNorm with running-stat updates, drop_connect); when autograd is enabled the returned
        tensors carry a graph whose backward runs the HIP backward kernels (ccvpe_amd/train.py) and fills the
        parameters' .grad — there is deliberately no eager fallback.  drop_masks: optional {(encoder prefix, block):
        [B] 0/1 tensor} to inject the drop_connect draws (parity tests); None draws them like utils.py:145-150."""
        if self.training and self.precision != "fp32":
            raise NotImplementedError("ccvpe_amd: train mode is fp32 only")
        if not (grd.is_cuda and sat.is_cuda):
            raise RuntimeError("ccvpe_amd runs on the MI355X only: inputs must be device tensors")
        if self.training:
            from . import train
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                outs = train.apply(self, grd, sat, drop_masks)
            else:
                with torch.no_grad():
                    outs, _ = train.forward_train(self, grd, sat, drop_masks, rec=False)
            outs = list(outs)
            if self.ori_noise is not None:          # ori_prior returns the recomputed 20-shift volume (models.py:501-511)
                outs[3] = outs[3][:, len(self._loc_shifts()):]
            return tuple(outs)
        main = torch.cuda.current_stream()
        return self._forward_eval(grd, sat, main, self._side_stream() if EVAL_TWO_STREAMS else main)

    def _forward_eval(self, grd, sat, main, side):
        """One eval forward on the stream pair (main, side); main must be the current stream.
        (Round 5 measured a software pipeline over sub-batches on top of this — the encoders of half-batch 2 beside the decoders
        of half-batch 1, four streams — and removed it: bf16 C1 at B = 64 9.45 -> 11.5 ms with two halves, 12.8 with four; the
        kernels lose more at the smaller batch than the extra concurrency returns.)"""
        spec = MODEL_SPECS[self.kind]
        n_rot = spec["n_rot"]
        strides = MATCH_STRIDES[self.kind]
        circular = bool(self.circular_padding) and self.kind == "vigor"   # models.py:660 (KITTI), :959 (Oxford)
        with torch.no_grad():
            pk = self._packed()
            grd = grd.contiguous().float()
            sat = sat.contiguous().float()
            batch = grd.shape[0]

            # encoders + descriptors (models.py:151-184).  The two encoders are independent until the
            # first matching block and their late MBConv blocks are small launches (<= 1 workgroup per
            # CU), so the ground encoder + descriptor heads run on a second HIP stream and overlap with
            # the aerial encoder (fork/join by events: also legal inside hipGraph capture).
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gfeat, _ = _run_encoder(pk.grd, grd, circular, False, pk.dtype)
                _, gh, gw, _ = gfeat.shape
                if gh != spec["grd_h"]:
                    raise ValueError("ground feature height %d != %d expected by the descriptor heads"
                                     % (gh, spec["grd_h"]))
                y1 = ops.conv_igemm(gfeat, 1280, pk.gd_w, pk.gd_n, batch=batch, in_h=gh, in_w=gw,
                                    shift=pk.gd_bias, ldd=_round_up(pk.gd_n, 4), out_f32=True)
                gdesc = ops.ground_descriptor(y1, pk.gd_wh, pk.gd_bh, spec["cd"])
            svol, sfeats = _run_encoder(pk.sat, sat, False, True, pk.dtype)
            main.wait_stream(side)
            gdesc.record_stream(main)
            sdesc = ops.conv_igemm(svol, 1280, pk.sd_w, pk.sd_n, batch=batch, in_h=svol.shape[1],
                                   in_w=svol.shape[2], kh=2, kw=2, stride=2, shift=pk.sd_bias)

            overlap = _overlap_decoders(self.precision, batch)
            loc_shifts = self._loc_shifts()
            fused = None
            scores_out = []
            smx = None
            x = sdesc
            cat6 = None
            goff = 0
            sfeats_loc = sfeats
            for j in range(6):
                lv = pk.loc[j]
                if j == pk.f32_from:
                    # bf16 storage path: from here on the localisation decoder runs in fp32 (ops dispatch on the
                    # activation dtype); widen its input and the skips it still needs — the orientation decoder keeps
                    # reading the bf16 originals
                    x = ops.cast_f32(x)
                    sfeats_loc = dict((SKIP_BLOCKS[q], ops.cast_f32(sfeats[SKIP_BLOCKS[q]])) for q in range(j, 5))
                hw = x.shape[1]
                cd = spec["cd"][j]
                L = gw * cd
                g = gdesc[:, goff:goff + L]
                goff += L
                if j == 0:
                    if self.ori_noise is None:
                        shifts, n_max = list(range(n_rot)), n_rot
                    else:     # ori_prior recomputes the full volume for the ori decoder (:501-511)
                        shifts, n_max = loc_shifts + list(range(n_rot)), len(loc_shifts)
                    n_tail = n_rot
                else:
                    shifts, n_max, n_tail = loc_shifts, len(loc_shifts), 0
                if fused is not None:                # the previous level's last conv already produced this level's (scores, cat)
                    sc, cat = fused
                    fused = None
                else:
                    sc, cat = ops.match_level(x, g, L, shifts, n_max, n_tail, strides[j], lv.ldo, channels=lv.c,
                                              window_offset=window_offset(self.kind, lv.c, L))
                if j == 0:
                    cat6 = cat
                    scores_out.append(sc if self.ori_noise is None else sc[:, n_max:])
                    # the orientation decoder (models.py:323-341) needs only cat6 + the skips.  Optionally it
                    # runs on the side stream concurrently with the localisation decoder (measured: only
                    # 39.29 -> 39.09 ms, both are MFMA-bound; off by default so that per-kernel timings
                    # of the dominant GEMMs are not blurred by a co-running kernel)
                    if overlap:
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            x_ori = self._ori_decoder(pk, cat6, sfeats, batch)
                else:
                    scores_out.append(sc)
                skip = sfeats_loc[SKIP_BLOCKS[j]] if j < 5 else None
                if j == 5 and FUSE_TAIL and j in FOLD_LEVELS and ops.tail512_ok(hw, hw, lv.n_a):
                    # the whole 512 x 512 level in one launch: deconv1 + conv1 -> logits (models.py:124-127,319)
                    # (bf16 storage path: its fp32 tail multiplies on the bf16 matrix cores with hi + lo operand planes — fp32-class
                    # accuracy at a quarter of the matrix cycles; the fp32 path is exact fp32)
                    logits_map, smx = ops.tail512(cat, lv.ldo, lv.fw, lv.fshift, lv.w_b, lv.b_b, 1, False, batch=batch, h1=hw, w1=hw,
                                                  split=(self.precision == "bf16" and cat.dtype == torch.float32 and SPLIT_TAIL),
                                                  want_softmax=True)
                    break
                if j in FOLD_LEVELS and batch * hw * hw >= FOLD_MIN_PIXELS:   # deconv folded into conv.0: one GEMM per output parity
                    y = ops.upconv3x3(cat, lv.ldo, lv.fw, lv.fshift, lv.n_a, batch=batch, h1=hw, w1=hw,
                                      src1=skip, c1=lv.c1, act=ops.ACT_RELU)
                else:
                    up = ops.conv_igemm(cat, lv.ldo, lv.up_w, lv.up_n, batch=batch, in_h=hw, in_w=hw,
                                        shift=lv.up_b, out_mode=ops.OUT_DECONV2X, algo_k=lv.c + 1)
                    y = _double_conv(lv, up, skip, batch, 2 * hw)
                if j < 5:
                    # the level feeding the fp32 tail writes its output in fp32 directly (no separate widening pass)
                    to_f32 = j + 1 == pk.f32_from and y.dtype != torch.float32
                    nxt = pk.loc[j + 1]
                    L_n = gw * spec["cd"][j + 1]
                    if (FUSE_MATCH and len(loc_shifts) == 1 and
                            ops.conv3x3_match1(y, lv.n_a, lv.w_b, lv.n_b, None, L_n, loc_shifts[0], strides[j + 1], nxt.ldo, batch=batch,
                                               in_h=2 * hw, in_w=2 * hw, bias=lv.b_b, out_f32=to_f32, query_only=True)):
                        # one rotation hypothesis (ori_prior(0)): the next level's matching rides in this conv's epilogue
                        # (bf16 narrow levels: csrc/narrow_impl.h c3n_kernel<MATCH>); x itself is never written
                        fused = ops.conv3x3_match1(y, lv.n_a, lv.w_b, lv.n_b, gdesc[:, goff:goff + L_n], L_n, loc_shifts[0],
                                                   strides[j + 1], nxt.ldo, batch=batch, in_h=2 * hw, in_w=2 * hw, bias=lv.b_b,
                                                   window_offset=window_offset(self.kind, nxt.c, L_n), out_f32=to_f32)
                        x = fused[1]                 # (only its shape is read at the top of the next level)
                    else:
                        x = ops.conv_igemm(y, lv.n_a, lv.w_b, lv.n_b, batch=batch, in_h=2 * hw, in_w=2 * hw,
                                           kh=3, kw=3, pad=1, shift=lv.b_b, out_f32=to_f32)
                else:
                    logits_map = ops.head_conv3x3(y, lv.w_b, lv.b_b, 1, False)      # [B,1,512,512]
            logits = logits_map.reshape(batch, -1)                                   # models.py:319
            if smx is not None:      # the fused 512 x 512 level left per-tile softmax partials: one pass instead of three sweeps
                heatmap = ops.softmax_apply(logits, smx).reshape(logits_map.shape)   # models.py:320
            else:
                heatmap = ops.softmax_rows(logits).reshape(logits_map.shape)

            if overlap:
                main.wait_stream(side)          # join the orientation decoder
                x_ori.record_stream(main)
            else:
                x_ori = self._ori_decoder(pk, cat6, sfeats, batch)
        return (logits, heatmap, x_ori) + tuple(scores_out)


class CVM_VIGOR(_CVMBase):
    """models.py:49-343 — training-time VIGOR model, 20 rotation hypotheses at every level."""
    kind = "vigor"

    def __init__(self, device, circular_padding, efficientnet_weights=None):
        super().__init__(device, circular_padding, None, efficientnet_weights)


class CVM_VIGOR_ori_prior(_CVMBase):
    """models.py:346-652 — test-time VIGOR model with an orientation prior of +-ori_noise deg."""
    kind = "vigor"

    def __init__(self, device, ori_noise, circular_padding=True, efficientnet_weights=None):
        super().__init__(device, circular_padding, ori_noise, efficientnet_weights)


class CVM_OxfordRobotCar(_CVMBase):
    """models.py:954-1246 — the VIGOR network on 154x231 ground images (4 x 7 ground feature map, descriptor lengths
    224 ... 7), zero padding, and a matching window CENTRED in the rolled aerial volume (models.py:1094)."""
    kind = "oxford"

    def __init__(self, device, efficientnet_weights=None):
        super().__init__(device, False, None, efficientnet_weights)


class CVM_KITTI(_CVMBase):
    """models.py:655-950 — KITTI model: 16 hypotheses, 2048-d aerial descriptor, no circular pad."""
    kind = "kitti"

    def __init__(self, device, efficientnet_weights=None):
        super().__init__(device, False, None, efficientnet_weights)
