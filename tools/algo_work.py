#!/usr/bin/env python3
"""Algorithmic work per image pair of every benched configuration -> profiles/algo_work.json (read by bench.py).

BASELINE.md section 3 prices the whole step with two numbers per model — forward GFLOP per pair (2 x MACs of every conv /
deconv / linear executed) and activation bytes per pair (each such op's input read once + output written once; BatchNorm,
activations, SE scaling, concat and normalise assumed fused) — measured there for VIGOR FoV 360 and KITTI only.  The FoV-180
configuration (BASELINE C4: ground image 320 x 320, 21 shifts) and the N_rot = 1 localisation branch of C1 do less work, so
pricing them with the FoV-360 constants overstates their roofline fraction.  This script measures the same two numbers for
every benched configuration by running the CPU oracle (test infrastructure: oracle/ccvpe_oracle.py, B = 1) under a
TorchDispatchMode that sees every aten convolution / matmul, plus the matching contraction counted in closed form
(2 * n_shifts * L * pixels per level).

    python tools/algo_work.py            # writes profiles/algo_work.json (a few seconds per configuration, CPU)
"""
import json
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccvpe_amd import synth                     # noqa: E402
from oracle import ccvpe_oracle as O            # noqa: E402

CONFIGS = {
    # key: (oracle kind, ground shape key, circular_padding, ori_noise, what)
    "vigor_prior0": ("vigor", "vigor", True, 0, "C1: CVM_VIGOR_ori_prior(0), FoV 360: 1 shift in the loc branch, 20 for the ori input"),
    "vigor20": ("vigor", "vigor", True, None, "C2 / fwd+bwd: CVM_VIGOR, FoV 360, 20 shifts on all six levels"),
    "vigor_prior180_fov180": ("vigor", "vigor_fov180", False, 180, "C4: CVM_VIGOR_ori_prior(180, circular_padding=False), ground 320 x 320, 21 shifts"),
    "kitti": ("kitti", "kitti", False, None, "C3: CVM_KITTI, ground 256 x 1024, 16 shifts"),
    "oxford": ("oxford", "oxford", False, None, "CVM_OxfordRobotCar, ground 154 x 231, 20 shifts"),
}


class Counter(TorchDispatchMode):
    """2 x MACs and (input + output) fp32 bytes of every convolution / transposed convolution seen; einsum / matmul work
    (descriptor height collapse, matching numerators) is counted as FLOPs only — its operands are already counted as a
    convolution's output."""

    def __init__(self):
        super().__init__()
        self.flops = 0
        self.bytes = 0
        self.weight_bytes = 0
        self.n_conv = 0
        self.mm_flops = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if "convolution" in name and "backward" not in name:
            x, w = args[0], args[1]
            transposed = bool(args[6])
            groups = int(args[8])
            if transposed:                      # weight [Cin, Cout/groups, kh, kw]: every input pixel meets every (Cout, kh, kw)
                macs = x.numel() * w.shape[1] * w.shape[2] * w.shape[3]
            else:                               # weight [Cout, Cin/groups, kh, kw]: every output value is Cin/groups * kh * kw MACs
                macs = out.numel() * w.shape[1] * w.shape[2] * w.shape[3]
            del groups
            self.flops += 2 * macs
            self.bytes += 4 * (x.numel() + out.numel())
            self.weight_bytes += 4 * w.numel()
            self.n_conv += 1
        elif name.startswith("aten.mm") or name.startswith("aten.bmm") or name.startswith("aten.addmm"):
            a, b = (args[1], args[2]) if name.startswith("aten.addmm") else (args[0], args[1])
            self.mm_flops += 2 * a.numel() * b.shape[-1]
        return out


def measure(key):
    kind, gshape, circ, ori_noise, what = CONFIGS[key]
    sd = synth.synthetic_state_dict(kind, 0)
    grd, sat = synth.synthetic_pair(1, gshape, 1234)
    with torch.no_grad(), Counter() as c:
        out = O.forward(sd, grd, sat, kind, circ, ori_noise=ori_noise)
    # matching: per level, scores[B, n, H, W] = n shifts x (L-term dot product + window norm) per aerial cell; the score volumes the
    # forward returns carry n and the cell count, L is the ground descriptor length of the level
    spec = synth.MODEL_SPECS[kind]
    gw = grd.shape[-1] // 32                     # ground feature-map width
    gh = grd.shape[-2] // 32
    del gh
    match_flops = 0
    for lvl in range(6):
        s = out[3 + lvl]
        n_sh, cells = s.shape[1], s.shape[2] * s.shape[3]
        cd = spec["cd"][lvl] if "cd" in spec else None
        L = gw * cd if cd is not None else 0
        match_flops += 2 * 2 * n_sh * L * cells   # numerator + window norm
    return {"what": what, "gflop_per_pair": round(c.flops / 1e9, 3), "mb_per_pair_fp32": round(c.bytes / 1e6, 1),
            "weights_mb_fp32": round(c.weight_bytes / 1e6, 1), "convs": c.n_conv,
            "matching_gflop_per_pair": round(match_flops / 1e9, 3), "other_matmul_gflop": round(c.mm_flops / 1e9, 4),
            "grd_shape": list(grd.shape[1:]), "sat_shape": list(sat.shape[1:]),
            "score_shifts": [int(out[3 + l].shape[1]) for l in range(6)]}


def main():
    res = {"#meta": {"script": "tools/algo_work.py", "rule": "BASELINE.md section 3: 2 x MACs of every conv / deconv / linear; bytes = "
                     "each op's fp32 input + output once (bf16 = half); B = 1 oracle forward under a TorchDispatchMode",
                     "torch": torch.__version__}}
    for key in CONFIGS:
        res[key] = measure(key)
        print(key, json.dumps(res[key]))
    path = os.path.join(ROOT, "profiles", "algo_work.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
