"""Construction-time parameter initialisation equivalent to the reference's.

The reference builds its models from stock torch layers (/root/reference/models.py:55-148):
  * decoder / descriptor layers — nn.Conv2d, nn.ConvTranspose2d, nn.Linear with torch's DEFAULT initialisation:
      weight ~ kaiming_uniform_(a = sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)),  bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in));
      fan_in = Cin/groups * kh * kw for Conv2d and Linear, and weight.size(1) * kh * kw (= Cout * kh * kw) for ConvTranspose2d;
  * the two EfficientNet-B0 encoders — `EfficientNet.from_pretrained('efficientnet-b0', circular)`
    (efficientnet_pytorch/model.py:376-410): the same default layer initialisation (BatchNorm: weight 1, bias 0,
    running_mean 0, running_var 1), then the ImageNet checkpoint is loaded over it (utils.py:701-761).
There is no network here, so the checkpoint is OPTIONAL: `load_efficientnet_b0(model, path_or_state_dict)` copies a
local lukemelas `efficientnet-b0-*.pth` (key names `_conv_stem.weight`, `_bn0.*`, `_blocks.N.*`, `_conv_head.weight`,
`_bn1.*`, `_fc.*`) into BOTH encoders exactly as `from_pretrained(weights_path=...)` does; the CVM_* constructors call it
when given `efficientnet_weights=` or when the environment variable CCVPE_EFFICIENTNET_B0 names a file.

(`ccvpe_amd.synth` — the deterministic hash-based weights with randomised BatchNorm statistics — stays what tests and
bench.py load explicitly: parity fixtures need weights that give the heat-map a usable dynamic range.)"""
import math
import os

import torch

from .synth import state_dict_spec


def reference_init_state_dict(kind, generator=None):
    """Fresh reference-layout state_dict drawn from the reference's construction-time distributions (see module doc)."""
    sd = {}
    spec = state_dict_spec(kind)
    shapes = dict((k, s) for k, s, _ in spec)
    for key, shape, role in spec:
        if role == "bn_n":
            t = torch.zeros((), dtype=torch.int64)
        elif role in ("bn_w", "bn_v"):
            t = torch.ones(shape)
        elif role in ("bn_b", "bn_m"):
            t = torch.zeros(shape)
        else:
            wshape = shape if key.endswith(".weight") else shapes[key[:-len("bias")] + "weight"]
            recept = 1
            for d in wshape[2:]:
                recept *= d
            # torch.nn.init._calculate_fan_in_and_fan_out: fan_in = size(1) * receptive field — for a ConvTranspose2d
            # weight [Cin, Cout, kh, kw] that is Cout * kh * kw, for Conv2d [Cout, Cin/groups, kh, kw] and Linear [out, in]
            # the usual Cin/groups * kh * kw
            fan_in = wshape[1] * recept
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=generator) * 2.0 - 1.0) * bound
        sd[key] = t
    return sd


EFFNET_PREFIXES = ("grd_efficientnet", "sat_efficientnet")


def load_efficientnet_b0(model, weights, load_fc=True):
    """`load_pretrained_weights(model, 'efficientnet-b0', weights_path=...)` (efficientnet_pytorch/utils.py:729-761) for
    both encoders of a CVM_* model: `weights` is a path to (or the loaded state_dict of) a lukemelas EfficientNet-B0
    checkpoint.  Same checks as the reference: no missing keys (except `_fc.*` when load_fc is False), no unexpected keys."""
    sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, os.PathLike)) else dict(weights)
    if not load_fc:
        sd = {k: v for k, v in sd.items() if not k.startswith("_fc.")}
    own = model.state_dict()
    for prefix in EFFNET_PREFIXES:
        want = set(k[len(prefix) + 1:] for k in own if k.startswith(prefix + "."))
        missing = want - set(sd)
        if not load_fc:
            missing -= {"_fc.weight", "_fc.bias"}
        unexpected = set(sd) - want
        if missing:
            raise KeyError("Missing keys when loading pretrained weights: %s" % sorted(missing)[:8])
        if unexpected:
            raise KeyError("Unexpected keys when loading pretrained weights: %s" % sorted(unexpected)[:8])
        with torch.no_grad():
            for k, v in sd.items():
                dst = own[prefix + "." + k]
                if tuple(dst.shape) != tuple(v.shape):
                    raise ValueError("%s: checkpoint shape %s != model shape %s" % (k, tuple(v.shape), tuple(dst.shape)))
                dst.copy_(v)
    if hasattr(model, "invalidate"):
        model.invalidate()
    return model
