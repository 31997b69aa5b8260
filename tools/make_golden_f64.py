"""tests/golden/grad_vigor_trainmode_f64.npz: the gradients of golden_util.train_loss for the train-mode parity case,
computed by autograd through the ORACLE in float64 (the oracle's fp32 gradients are pinned against the reference's
in tests/test_oracle_golden.py).  It is the round-off-free value of what the reference computes in fp32, used to
judge fp32 implementations against each other: B = 2 batch-statistic BatchNorm amplifies fp32 round-off to the
0.25 % (median) .. 2.5 % (worst tensor) level in the reference's own gradients.
Usage: python tools/make_golden_f64.py     (CPU, ~1 min; needs nothing from /root/reference)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as G                           # noqa: E402
from ccvpe_amd import synth                       # noqa: E402
from oracle import ccvpe_oracle as O              # noqa: E402


def main():
    torch.set_num_threads(8)
    c = G.TRAIN_CASE
    sd = synth.synthetic_state_dict(c["kind"], c["wseed"])
    params = {}
    for k, v in sd.items():
        if v.is_floating_point():
            v = v.double()
            params[k] = v.clone().requires_grad_(True) if "running_" not in k else v.clone()
        else:
            params[k] = v.clone()
    grd, sat = synth.synthetic_pair(c["batch"], c["grd"], c["pseed"])
    _, scales, _ = G.train_drop_masks(c["batch"])
    scales = {e: {i: m.double() for i, m in d.items()} for e, d in scales.items()}
    out = O.forward(params, grd.double(), sat.double(), c["kind"], c["circular"], None, train_stats={}, drop_scales=scales)
    G.train_loss(out).backward()
    ref = G.load("grad_vigor_trainmode")
    names = set(str(n) for n in ref["names"])
    d = G.summarize_grads([(k, v.grad) for k, v in params.items() if k in names])
    assert [str(n) for n in d["names"]] == [str(n) for n in ref["names"]]
    path = os.path.join(G.GOLDEN_DIR, "grad_vigor_trainmode_f64.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
