"""CPU ORACLE for the CCVPE dense cross-view matching forward path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product (`ccvpe_amd/`) never does and fails loudly without its HIP library.

What it is: a functional, fp32, torch-CPU restatement of the reference's algorithm for
`CVM_VIGOR` / `CVM_VIGOR_ori_prior` / `CVM_KITTI` / `CVM_OxfordRobotCar` `forward(grd, sat)` (eval mode, and train
mode with batch-statistic BatchNorm + injected drop_connect draws; differentiable, so autograd through it gives
reference gradients), the three losses, the training ground truth of the datasets and the PIL/torchvision input
transform, written from the reference's behaviour and restructured (no nn.Module tree, BN
applied from running statistics, one fused ground-descriptor contraction, the aerial
descriptor as a 2x2/s2 convolution, rotational matching in closed circulant form).  All
arithmetic is delegated to torch CPU kernels exactly as the reference delegates to torch
(the reference pins no torch version and holds no tests or golden vectors
of its own).

Parity pinning: this oracle is pinned against the reference ITSELF, imported in the build
container by `tools/make_golden.py`; the resulting vectors are committed under
`tests/golden/` (5 eval forwards, a train-mode forward with running statistics, the reference's autograd gradients
for 520 parameter tensors, per-module outputs, losses) and re-checked by `tests/test_oracle_golden.py` everywhere, and
`tests/test_oracle_vs_reference.py` compares live whenever /root/reference is present.

Every function cites the reference file:line it follows (paths under /root/reference).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # efficientnet_pytorch/utils.py:666 (batch_norm_epsilon)
STATIC_IMAGE_SIZE = 224  # efficientnet_pytorch/utils.py:613: padding schedule is built for 224

# (kernel, stride, expand, cin, cout) — efficientnet_pytorch/utils.py:647-655, repeats unrolled
# the way EfficientNet.__init__ does it (model.py:187-202).
B0_BLOCKS = (
    (3, 1, 1, 32, 16),
    (3, 2, 6, 16, 24), (3, 1, 6, 24, 24),
    (5, 2, 6, 24, 40), (5, 1, 6, 40, 40),
    (3, 2, 6, 40, 80), (3, 1, 6, 80, 80), (3, 1, 6, 80, 80),
    (5, 1, 6, 80, 112), (5, 1, 6, 112, 112), (5, 1, 6, 112, 112),
    (5, 2, 6, 112, 192), (5, 1, 6, 192, 192), (5, 1, 6, 192, 192), (5, 1, 6, 192, 192),
    (3, 1, 6, 192, 320),
)
SKIP_BLOCKS = (15, 10, 4, 2, 0)   # models.py:167-171, ordered for decoder levels 6..2


# ----------------------------------------------------------------------------------------
# EfficientNet-B0 feature extractor
# ----------------------------------------------------------------------------------------
def static_same_pad(size, k, s):
    """(before, after) TF-'SAME' padding computed for the static schedule size.
    efficientnet_pytorch/utils.py:265-277 (zero) and :341-353 (circular)."""
    out = math.ceil(size / s)
    total = max((out - 1) * s + (k - 1) + 1 - size, 0)
    return total // 2, total - total // 2


def same_conv(x, w, k, s, sched, circular, groups=1):
    """Conv2dStaticSamePadding / Conv2dStaticCircularPadding forward
    (utils.py:279-282, :355-358): pad from the 224 schedule, then conv with padding 0.
    Circular variant: circular along W, zeros along H (utils.py:350-351)."""
    pb, pa = static_same_pad(sched, k, s)
    if pb or pa:
        if circular:
            x = F.pad(x, [pb, pa, 0, 0], mode="circular")
            x = F.pad(x, [0, 0, pb, pa])
        else:
            x = F.pad(x, [pb, pa, pb, pa])
    return F.conv2d(x, w, None, s, 0, 1, groups)


BN_MOMENTUM = 0.01     # model.py:52: 1 - batch_norm_momentum(0.99)


def bn_eval(x, sd, p, train=None):
    """nn.BatchNorm2d (model.py:63,73,87,182,210).  Eval mode: running statistics.  Train mode
    (`train` = dict collecting the updated running statistics): batch statistics, running statistics
    updated with momentum 0.01 and the unbiased variance, exactly as torch does."""
    if train is None:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                            sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)
    rm, rv = sd[p + ".running_mean"].clone(), sd[p + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, BN_MOMENTUM, BN_EPS)
    train[p + ".running_mean"], train[p + ".running_var"] = rm, rv
    return y


def swish(x):
    """SwishImplementation.forward (utils.py:64-69)."""
    return x * torch.sigmoid(x)


def mbconv(x, sd, p, k, s, e, cin, cout, sched, circular, train=None, drop_scale=None):
    """MBConvBlock.forward (model.py:90-131).  Eval: drop_connect is the identity (utils.py:141-142).
    Train: `drop_scale` [B] = binary_mask / keep_prob multiplies the block output before the skip
    (utils.py:145-153; the mask itself is random upstream, so parity tests inject it)."""
    inp = x
    if e != 1:
        x = swish(bn_eval(same_conv(x, sd[p + "._expand_conv.weight"], 1, 1, sched, circular),
                          sd, p + "._bn0", train))
    mid = cin * e
    x = same_conv(x, sd[p + "._depthwise_conv.weight"], k, s, sched, circular, groups=mid)
    x = swish(bn_eval(x, sd, p + "._bn1", train))
    # squeeze-excite (model.py:113-118)
    z = x.mean(dim=(2, 3), keepdim=True)
    z = swish(F.conv2d(z, sd[p + "._se_reduce.weight"], sd[p + "._se_reduce.bias"]))
    z = F.conv2d(z, sd[p + "._se_expand.weight"], sd[p + "._se_expand.bias"])
    x = torch.sigmoid(z) * x
    x = bn_eval(F.conv2d(x, sd[p + "._project_conv.weight"]), sd, p + "._bn2", train)
    if s == 1 and cin == cout:          # model.py:126-130
        if drop_scale is not None:
            x = x * drop_scale.view(-1, 1, 1, 1)
        x = x + inp
    return x


def efficientnet_features(x, sd, prefix, circular, train=None, drop_scales=None):
    """EfficientNet.extract_features_multiscale (model.py:303-326); extract_features
    (model.py:278-301) is the same computation without the list.
    Returns (head_out [B,1280,h,w], [16 block outputs]).  drop_scales: {block index: [B]} (train)."""
    sched = STATIC_IMAGE_SIZE
    x = swish(bn_eval(same_conv(x, sd[prefix + "._conv_stem.weight"], 3, 2, sched, circular),
                      sd, prefix + "._bn0", train))
    sched = math.ceil(sched / 2)
    feats = []
    for i, (k, s, e, cin, cout) in enumerate(B0_BLOCKS):
        ds = None if drop_scales is None else drop_scales.get(i)
        x = mbconv(x, sd, "%s._blocks.%d" % (prefix, i), k, s, e, cin, cout, sched, circular, train, ds)
        sched = math.ceil(sched / s)
        feats.append(x)
    x = swish(bn_eval(F.conv2d(x, sd[prefix + "._conv_head.weight"]), sd, prefix + "._bn1", train))
    return x, feats


# ----------------------------------------------------------------------------------------
# descriptors
# ----------------------------------------------------------------------------------------
def ground_descriptor(feat, sd, level):
    """grd_feature_to_descriptorK (models.py:57-97, KITTI :662-699):
    1x1 conv 1280->Cd, permute (0,2,3,1), 1x1 conv over the height axis, flatten.
    Closed form: D[b, w*Cd + c] = sum_h wH[h] * (W1[c,:].F[b,:,h,w] + b1[c]) + bH."""
    p = "grd_feature_to_descriptor%d" % level
    y = F.conv2d(feat, sd[p + ".0.weight"], sd[p + ".0.bias"])          # [B,Cd,h,w]
    wh = sd[p + ".2.weight"].reshape(-1)                                 # [h]
    d = torch.einsum("bchw,h->bwc", y, wh) + sd[p + ".2.bias"]           # [B,w,Cd]
    return d.reshape(d.shape[0], -1)


def aerial_descriptor(vol, sd):
    """models.py:173-184 with sat_feature_to_descriptors (:102-104): each non-overlapping
    2x2 patch, flattened (c,h,w), goes through one Linear == conv 2x2 stride 2 with
    W.view(N,1280,2,2)."""
    w = sd["sat_feature_to_descriptors.1.weight"]
    return F.conv2d(vol, w.view(w.shape[0], vol.shape[1], 2, 2),
                    sd["sat_feature_to_descriptors.1.bias"], stride=2)


# ----------------------------------------------------------------------------------------
# rotational matching
# ----------------------------------------------------------------------------------------
def rotational_matching(x, g, shifts, stride, win_off=0):
    """One matching block (models.py:186-202 and the five that follow; ori_prior :484-511;
    KITTI :788-804): for shift i, window[c] = x[(c + i*stride) mod C] for c < L
    (= roll(x, -i*stride, 1)[:, :L]); score_i = <g, window> / (||window|| * ||g||), no eps.
    x [B,C,H,W], g [B,L]; returns scores [B,len(shifts),H,W]."""
    B, C, H, W = x.shape
    L = g.shape[1]
    gn = g.norm(dim=1).view(B, 1, 1)
    base = torch.arange(L)
    out = []
    for i in shifts:
        idx = (base + win_off + i * stride) % C      # win_off: CVM_OxfordRobotCar's centred window (models.py:1094)
        win = x.index_select(1, idx)                                     # [B,L,H,W]
        num = torch.einsum("bl,blhw->bhw", g, win)
        out.append(num / (win.norm(dim=1) * gn))
    return torch.stack(out, dim=1)


def lmu_input(x, scores):
    """cat[max over rotations, L2-normalised features] (models.py:202-205; F.normalize eps
    1e-12 at :40)."""
    return torch.cat([scores.max(dim=1, keepdim=True)[0], F.normalize(x, p=2, dim=1)], dim=1)


def double_conv(x, sd, p):
    """models.py:42-47."""
    x = F.relu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1))
    return F.conv2d(x, sd[p + ".2.weight"], sd[p + ".2.bias"], padding=1)


def up(x, sd, p):
    """nn.ConvTranspose2d(k=2, s=2) (models.py:109...145)."""
    return F.conv_transpose2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=2)


# ----------------------------------------------------------------------------------------
# full forward
# ----------------------------------------------------------------------------------------
MODEL_CFG = {
    # strides per level (models.py:192,217,239,261,283,305) and hypotheses
    "vigor": dict(strides=(64, 32, 16, 8, 4, 2), n_rot=20),
    # models.py:794,818,841,864,887,910
    "kitti": dict(strides=(128, 64, 32, 16, 8, 8), n_rot=16),
    # models.py:1093,1116,1140,1163,1186,1209 (CVM_OxfordRobotCar)
    "oxford": dict(strides=(64, 32, 16, 8, 4, 2), n_rot=20),
}


def forward(sd, grd, sat, kind="vigor", circular_padding=True, ori_noise=None,
            return_intermediates=False, train_stats=None, drop_scales=None):
    """CVM_VIGOR.forward (models.py:150-343) when kind='vigor' and ori_noise is None;
    CVM_VIGOR_ori_prior.forward (models.py:448-652) when ori_noise is a number;
    CVM_KITTI.forward (models.py:752-950) when kind='kitti'; CVM_OxfordRobotCar.forward (models.py:1049-1246)
    when kind='oxford' (VIGOR's network with a centred matching window and a 4 x 7 ground feature map).
    Returns the reference's 9-tuple."""
    cfg = MODEL_CFG[kind]
    n_rot = cfg["n_rot"]
    if kind in ("kitti", "oxford"):
        circular_padding = False                       # models.py:660, :959
    if ori_noise is None:
        loc_shifts = list(range(n_rot))                # models.py:191
    else:
        k = int(ori_noise / 18)                        # models.py:489
        loc_shifts = list(range(-k, k + 1))
    inter = {}

    # train_stats: pass a dict to run the encoders in TRAIN mode (batch-stat BN); it receives the updated
    # running statistics.  drop_scales: {"grd_efficientnet"|"sat_efficientnet": {block: [B] scale}}.
    dsg = None if drop_scales is None else drop_scales.get("grd_efficientnet")
    dss = None if drop_scales is None else drop_scales.get("sat_efficientnet")
    gfeat, _ = efficientnet_features(grd, sd, "grd_efficientnet", circular_padding, train_stats, dsg)
    gdesc = [ground_descriptor(gfeat, sd, l) for l in range(1, 7)]
    svol, sfeats = efficientnet_features(sat, sd, "sat_efficientnet", False, train_stats, dss)
    sdesc = aerial_descriptor(svol, sd)
    inter.update(grd_feature=gfeat, grd_descriptors=gdesc, sat_feature=svol,
                 sat_descriptor=sdesc, sat_skips=[sfeats[i] for i in SKIP_BLOCKS])

    scores = []
    x = sdesc
    def woff(xv, gv):       # models.py:1094: int(sat_des_len/2 - grd_des_len/2), window centred in the rolled volume
        return int(xv.shape[1] / 2 - gv.shape[1] / 2) if kind == "oxford" else 0

    for lvl in range(6):                               # level index 0..5 == reference 1..6
        sc = rotational_matching(x, gdesc[lvl], loc_shifts, cfg["strides"][lvl], woff(x, gdesc[lvl]))
        scores.append(sc)
        x = up(lmu_input(x, sc), sd, "deconv%d" % (6 - lvl))
        if lvl < 5:
            x = torch.cat([x, sfeats[SKIP_BLOCKS[lvl]]], dim=1)
        x = double_conv(x, sd, "conv%d" % (6 - lvl))
    logits = x.flatten(1)                              # models.py:319
    heatmap = torch.softmax(logits, dim=-1).reshape(x.shape)   # models.py:320

    # orientation decoder (models.py:323-341); ori_prior recomputes the full n_rot volume
    # at level 1 for it (models.py:501-511) and returns THAT as score1.
    if ori_noise is None:
        score1 = scores[0]
    else:
        score1 = rotational_matching(sdesc, gdesc[0], list(range(n_rot)), cfg["strides"][0], woff(sdesc, gdesc[0]))
        scores[0] = score1
    xo = torch.cat([score1, F.normalize(sdesc, p=2, dim=1)], dim=1)
    for lvl in range(6):
        xo = up(xo, sd, "deconv%d_ori" % (6 - lvl))
        if lvl < 5:
            xo = torch.cat([xo, sfeats[SKIP_BLOCKS[lvl]]], dim=1)
        xo = double_conv(xo, sd, "conv%d_ori" % (6 - lvl))
    inter["ori_raw"] = xo                              # conv1_ori's output before models.py:341 (tests: where |raw| is small F.normalize amplifies any error)
    xo = F.normalize(xo, p=2, dim=1)                   # models.py:341

    out = (logits, heatmap, xo) + tuple(scores)
    if return_intermediates:
        return out, inter
    return out


# ----------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------
def infonce_loss(scores, labels, temperature=0.1):
    """losses.py:4-20.  masked_select == masked sum; the normaliser is joint over the batch."""
    e = torch.exp(scores / temperature)
    mask = labels > 1e-2
    p = e / e.sum(dim=1, keepdim=True)
    lab = torch.where(mask, labels, torch.zeros_like(labels))
    logp = torch.where(mask, torch.log(p), torch.zeros_like(p))
    return -(logp * lab).sum() / lab.sum()


def cross_entropy_loss(logits, labels):
    """losses.py:23-24."""
    return -(labels * torch.log_softmax(logits, dim=1)).sum() / logits.shape[0]


def orientation_loss(ori, gt_orientation, gt):
    """losses.py:28-29."""
    return (((gt_orientation - ori) ** 2).sum(dim=1, keepdim=True) * gt).sum() / ori.shape[0]


# ----------------------------------------------------------------------------------------
# training ground truth (what the datasets build per sample and the training loop max-pools)
# ----------------------------------------------------------------------------------------
def train_targets(center_xy, angle_deg, n_bins, height=512, width=512, sigma=4.0, ascending=False):
    """datasets.py:145-166 (VIGOR: 20 bins of 18 deg; cx = col_offset, cy = -row_offset) and :470-501 (KITTI: 16 bins of
    22.5 deg; cx = x_offset, cy = y_offset), then train_VIGOR.py:120-128: gt, gt / sum(gt), the (cos, sin) map and
    MaxPool2d(k, k)(gt_with_ori) for k = 64..2.  numpy float64 -> float32 exactly as the reference does.
    ascending=True: the Oxford RobotCar loader's bin order (datasets.py:340-347; cx / cy = col / row_offset_resized)."""
    import numpy as np
    bw = 360.0 / n_bins
    gts, oris, gwo = [], [], []
    for (cx, cy), ang in zip(np.asarray(center_xy, dtype=np.float64), np.asarray(angle_deg, dtype=np.float64)):
        x, y = np.meshgrid(np.linspace(-width / 2 + cx, width / 2 + cx, width),
                           np.linspace(-height / 2 + cy, height / 2 + cy, height))
        d = np.sqrt(x * x + y * y)
        e = np.exp(-(d ** 2 / (2.0 * sigma ** 2)))            # float64, rounded to float32 once per use like the reference
        g = e.astype(np.float32)
        w = np.zeros([n_bins, height, width], dtype=np.float32)
        index = int(ang // bw)
        ratio = (ang % bw) / bw
        if ascending:
            w[index] = e * (1 - ratio)
            w[0 if index == n_bins - 1 else index + 1] = e * ratio
        elif index == 0:
            w[0] = e * (1 - ratio)
            w[n_bins - 1] = e * ratio
        else:
            w[n_bins - index] = e * (1 - ratio)
            w[n_bins - index - 1] = e * ratio
        gts.append(g[None])
        gwo.append(w)
        o = np.empty([2, height, width], dtype=np.float32)
        o[0] = np.cos(ang * np.pi / 180)
        o[1] = np.sin(ang * np.pi / 180)
        oris.append(o)
    gt = torch.tensor(np.stack(gts))
    gt_with_ori = torch.tensor(np.stack(gwo))
    flat = torch.flatten(gt, start_dim=1)
    flat = flat / torch.sum(flat, dim=1, keepdim=True)
    labs = [F.max_pool2d(gt_with_ori, k, stride=k) for k in (64, 32, 16, 8, 4, 2)]
    return gt, flat, torch.tensor(np.stack(oris)), labs


# ----------------------------------------------------------------------------------------
# input pipeline after JPEG decoding
# ----------------------------------------------------------------------------------------
def preprocess_reference(img_u8, out_hw, roll=0, keep_w=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """train_VIGOR.py:57-70 transform (Resize -> ToTensor -> Normalize) with the arithmetic torchvision performs —
    Resize on a PIL image IS PIL.Image.resize(BILINEAR) (third-party dependency: Pillow, 12.2.0 in this image; its
    resampler is the algorithm ccvpe_amd/preprocess.py + csrc/preprocess.hip restate) — then the panorama roll of
    datasets.py:121 and the FoV crop of train_VIGOR.py:177-178.  img_u8: numpy [H,W,3] uint8."""
    import numpy as np
    from PIL import Image
    h, w = out_hw
    im = Image.fromarray(np.ascontiguousarray(img_u8), "RGB").resize((w, h), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255)                     # ToTensor
    t = (t - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)                      # Normalize
    t = torch.roll(t, int(roll), dims=2)
    return t if keep_w is None else t[:, :, :keep_w]


# ----------------------------------------------------------------------------------------
# evaluation post-processing
# ----------------------------------------------------------------------------------------
def eval_postprocess(heatmap, ori):
    """train_VIGOR.py:294-324 (train_KITTI.py:304-343 is the same): per sample the arg-max pixel
    of the heat-map (numpy argmax = first maximum), (cos, sin) there and the acos-based angle.
    Returns [B,6] = (y, x, cos, sin, angle_deg or NaN, prob)."""
    B, _, H, W = heatmap.shape
    out = torch.empty((B, 6), dtype=torch.float32)
    for b in range(B):
        flat = heatmap[b].reshape(-1)
        idx = int(flat.numpy().argmax())
        y, x = idx // W, idx % W
        c, s = float(ori[b, 0, y, x]), float(ori[b, 1, y, x])
        ang = float("nan")
        if abs(c) <= 1 and abs(s) <= 1:
            a = math.acos(c)
            ang = math.degrees(-a) % 360 if s < 0 else math.degrees(a)
        out[b] = torch.tensor([y, x, c, s, ang, float(flat[idx])])
    return out
