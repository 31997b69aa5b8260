"""Deterministic synthetic inputs and reference-layout weights.

There is no network (no VIGOR/KITTI images, no trained checkpoints), so tests, fixtures and
bench.py all use synthetic data.  Everything here is produced by an integer hash evaluated
with int64 torch ops on the CPU, so the very same bits come out in the build container (where
the golden vectors are made from the reference) and on the GPU box (where they are checked):
no dependence on any library RNG stream.

`synthetic_state_dict(kind, seed)` returns a state_dict with the reference's exact keys and
shapes (/root/reference/models.py:50-148, :656-749 and
efficientnet_pytorch/model.py:162-216).  BN running statistics are randomised and the
weights are variance-scaled so that activations keep O(1) magnitude through the 16 MBConv
blocks and the decoder, which gives the heat-map logits a usable dynamic range (with the
reference's default init the logits are ~flat and arg-max parity would be meaningless).
"""
import math
from collections import OrderedDict

import torch

_M32 = 0xFFFFFFFF


def _mul32(h, c):
    # (h * c) mod 2^32 without overflowing int64: h < 2^32 split in 16-bit halves.
    lo = h & 0xFFFF
    hi = h >> 16
    return (lo * c + (((hi * c) & 0xFFFF) << 16)) & _M32


def _fmix32(h):
    # murmur3 finaliser on 32-bit lanes held in int64
    h = h ^ (h >> 16)
    h = _mul32(h, 0x85EBCA6B)
    h = h ^ (h >> 13)
    h = _mul32(h, 0xC2B2AE35)
    h = h ^ (h >> 16)
    return h


def hash_u32(n, seed, device=None):
    """n hashed 32-bit words (int64 tensor) for stream `seed`.  `device`: where the integer ops run — the hash is exact integer
    arithmetic and the float conversions below are single IEEE operations, so a tensor generated on the MI355X is BIT-identical
    to the CPU one (tests/test_synth_device_gpu.py); large test batches are generated there (B = 64 pairs: 15 s on 8 host
    cores, milliseconds on the device)."""
    idx = torch.arange(n, dtype=torch.int64, device=device)
    s = (int(seed) * 0x9E3779B1 + 0x7F4A7C15) & _M32
    h = (idx & _M32) ^ s
    h = _fmix32(h)
    h = _fmix32((h + ((idx >> 32) & _M32) + 0x6A09E667) & _M32)
    return h


def uniform(shape, seed, lo=0.0, hi=1.0, device=None):
    n = int(math.prod(shape)) if len(shape) else 1
    u = (hash_u32(n, seed, device) >> 8).to(torch.float32) * (1.0 / 16777216.0)   # 24 bits, exact
    return (u * (hi - lo) + lo).reshape(shape)


def normal(shape, seed, std=1.0, mean=0.0, device=None):
    """Irwin-Hall(4) approximation of a unit normal: exact fp32 sums of 16-bit uniforms."""
    n = int(math.prod(shape)) if len(shape) else 1
    a = hash_u32(n, seed, device)
    b = hash_u32(n, seed ^ 0x5BD1E995, device)
    s = ((a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16)).to(torch.float32)
    z = (s * (1.0 / 65536.0) - 2.0) * math.sqrt(3.0)
    return (z * std + mean).reshape(shape)


# ----------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------
GRD_SHAPES = {"vigor": (320, 640), "vigor_fov180": (320, 320), "kitti": (256, 1024),
              "oxford": (154, 231)}          # train_OxfordRobotCar.py:50 (transforms.Resize([154, 231]))


def synthetic_pair(batch, kind="vigor", seed=1234, grd_hw=None, sat_hw=(512, 512), device=None):
    """(grd [B,3,h,w], sat [B,3,512,512]) fp32 NCHW, ~unit normal like ImageNet-normalised
    images (/root/reference/train_VIGOR.py:57-70).  device: generate there (same bits, see hash_u32)."""
    gh, gw = grd_hw if grd_hw is not None else GRD_SHAPES[kind]
    grd = normal((batch, 3, gh, gw), seed * 2 + 1, device=device)
    sat = normal((batch, 3, sat_hw[0], sat_hw[1]), seed * 2 + 2, device=device)
    return grd, sat


# ----------------------------------------------------------------------------------------
# reference-layout state_dict
# ----------------------------------------------------------------------------------------
# (kernel, stride, expand, cin, cout) for the 16 MBConv blocks of EfficientNet-B0
# (/root/reference/efficientnet_pytorch/utils.py:647-655, repeats unrolled).
B0_BLOCKS = (
    (3, 1, 1, 32, 16),
    (3, 2, 6, 16, 24), (3, 1, 6, 24, 24),
    (5, 2, 6, 24, 40), (5, 1, 6, 40, 40),
    (3, 2, 6, 40, 80), (3, 1, 6, 80, 80), (3, 1, 6, 80, 80),
    (5, 1, 6, 80, 112), (5, 1, 6, 112, 112), (5, 1, 6, 112, 112),
    (5, 2, 6, 112, 192), (5, 1, 6, 192, 192), (5, 1, 6, 192, 192), (5, 1, 6, 192, 192),
    (3, 1, 6, 192, 320),
)

# per model kind: ground-descriptor channels per column (Cd, levels 1..6), ground feature
# height, aerial descriptor length, decoder widths.
MODEL_SPECS = {
    # /root/reference/models.py:57-148
    "vigor": dict(
        cd=(64, 32, 16, 8, 4, 2), grd_h=10, sat_desc=1280, n_rot=20,
        loc=((1281, 1024, 1344, 640), (641, 320, 432, 320), (321, 160, 200, 160),
             (161, 80, 104, 80), (81, 40, 56, 40), (41, 16, 16, 16)),
        ori=((1300, 1024, 1344, 640), (640, 256, 368, 256), (256, 128, 168, 128),
             (128, 64, 88, 64), (64, 32, 48, 32), (32, 16, 16, 16)),
    ),
    # /root/reference/models.py:662-749
    "kitti": dict(
        cd=(16, 8, 4, 2, 1, 1), grd_h=8, sat_desc=2048, n_rot=16,
        loc=((2049, 1024, 1344, 512), (513, 256, 368, 256), (257, 128, 168, 128),
             (129, 64, 88, 128), (129, 32, 48, 32), (33, 16, 16, 16)),
        ori=((2064, 1024, 1344, 512), (512, 256, 368, 256), (256, 128, 168, 128),
             (128, 64, 88, 64), (64, 32, 48, 32), (32, 16, 16, 16)),
    ),
    # /root/reference/models.py:954-1047 (CVM_OxfordRobotCar: the VIGOR decoders, a 4 x 7 ground feature map)
    "oxford": dict(
        cd=(32, 16, 8, 4, 2, 1), grd_h=4, sat_desc=1280, n_rot=20,
        loc=((1281, 1024, 1344, 640), (641, 320, 432, 320), (321, 160, 200, 160),
             (161, 80, 104, 80), (81, 40, 56, 40), (41, 16, 16, 16)),
        ori=((1300, 1024, 1344, 640), (640, 256, 368, 256), (256, 128, 168, 128),
             (128, 64, 88, 64), (64, 32, 48, 32), (32, 16, 16, 16)),
    ),
}


def state_dict_spec(kind):
    """Ordered list of (key, shape, role) in the reference's registration order."""
    spec = MODEL_SPECS[kind]
    out = []

    def bn(prefix, c):
        out.append((prefix + ".weight", (c,), "bn_w"))
        out.append((prefix + ".bias", (c,), "bn_b"))
        out.append((prefix + ".running_mean", (c,), "bn_m"))
        out.append((prefix + ".running_var", (c,), "bn_v"))
        out.append((prefix + ".num_batches_tracked", (), "bn_n"))

    def effnet(p):
        out.append((p + "._conv_stem.weight", (32, 3, 3, 3), "conv"))
        bn(p + "._bn0", 32)
        for i, (k, s, e, cin, cout) in enumerate(B0_BLOCKS):
            b = "%s._blocks.%d" % (p, i)
            mid = cin * e
            if e != 1:
                out.append((b + "._expand_conv.weight", (mid, cin, 1, 1), "conv"))
                bn(b + "._bn0", mid)
            out.append((b + "._depthwise_conv.weight", (mid, 1, k, k), "conv"))
            bn(b + "._bn1", mid)
            sq = max(1, int(cin * 0.25))
            out.append((b + "._se_reduce.weight", (sq, mid, 1, 1), "conv"))
            out.append((b + "._se_reduce.bias", (sq,), "bias"))
            out.append((b + "._se_expand.weight", (mid, sq, 1, 1), "conv"))
            out.append((b + "._se_expand.bias", (mid,), "bias"))
            out.append((b + "._project_conv.weight", (cout, mid, 1, 1), "conv_lin"))
            bn(b + "._bn2", cout)
        out.append((p + "._conv_head.weight", (1280, 320, 1, 1), "conv"))
        bn(p + "._bn1", 1280)
        out.append((p + "._fc.weight", (1000, 1280), "conv_lin"))
        out.append((p + "._fc.bias", (1000,), "bias"))

    effnet("grd_efficientnet")
    for lvl, cd in enumerate(spec["cd"], 1):
        p = "grd_feature_to_descriptor%d" % lvl
        out.append((p + ".0.weight", (cd, 1280, 1, 1), "conv_lin"))
        out.append((p + ".0.bias", (cd,), "bias"))
        out.append((p + ".2.weight", (1, spec["grd_h"], 1, 1), "conv_lin"))
        out.append((p + ".2.bias", (1,), "bias"))
    effnet("sat_efficientnet")
    out.append(("sat_feature_to_descriptors.1.weight", (spec["sat_desc"], 5120), "conv_lin"))
    out.append(("sat_feature_to_descriptors.1.bias", (spec["sat_desc"],), "bias"))
    for branch, sfx in (("loc", ""), ("ori", "_ori")):
        for j, (dc_in, dc_out, c_in, c_out) in enumerate(spec[branch]):
            lvl = 6 - j
            out.append(("deconv%d%s.weight" % (lvl, sfx), (dc_in, dc_out, 2, 2), "deconv"))
            out.append(("deconv%d%s.bias" % (lvl, sfx), (dc_out,), "bias"))
            last = 1 if sfx == "" else 2
            c_out2 = c_out if lvl != 1 else last
            out.append(("conv%d%s.0.weight" % (lvl, sfx), (c_out, c_in, 3, 3), "conv"))
            out.append(("conv%d%s.0.bias" % (lvl, sfx), (c_out,), "bias"))
            out.append(("conv%d%s.2.weight" % (lvl, sfx), (c_out2, c_out, 3, 3), "conv_lin"))
            out.append(("conv%d%s.2.bias" % (lvl, sfx), (c_out2,), "bias"))
    return out


def synthetic_state_dict(kind="vigor", seed=0, head_gain=4.0):
    """Seeded reference-layout state_dict (fp32 CPU tensors).

    roles: "conv" is followed by a rectifier-like activation (He gain sqrt(2)),
    "conv_lin" is not (gain 1), "deconv" has fan_in = Cin (each output pixel sees one tap).
    The last loc conv (conv1.2) gets `head_gain` so the logits spread over several units.
    """
    sd = OrderedDict()
    for n, (key, shape, role) in enumerate(state_dict_spec(kind)):
        s = (seed * 1000003 + n * 7 + 11) & 0x7FFFFFFF
        if role == "bn_n":
            t = torch.zeros((), dtype=torch.int64)
        elif role == "bn_w":
            t = uniform(shape, s, 0.7, 1.3)
        elif role == "bn_v":
            t = uniform(shape, s, 0.6, 1.4)
        elif role in ("bn_b", "bn_m"):
            t = normal(shape, s, 0.1)
        elif role == "bias":
            t = normal(shape, s, 0.05)
        else:
            if role == "deconv":
                fan_in = shape[0]
                gain = 1.0
            else:
                fan_in = int(math.prod(shape[1:]))
                gain = math.sqrt(2.0) if role == "conv" else 1.0
            if key == "conv1.2.weight":
                gain = head_gain
            t = normal(shape, s, gain / math.sqrt(fan_in))
        sd[key] = t.contiguous()
    return sd
