"""Shared between tools/make_golden.py (writes fixtures from the REFERENCE) and the tests
(compare the oracle / the HIP path with them).  A fixture holds expected OUTPUTS only: the
inputs and weights are regenerated bit-exactly from ccvpe_amd.synth (integer hash)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> dict(kind, model args, weight seed, pair seed, batch, ground shape key)
FORWARD_CASES = {
    "vigor_train": dict(kind="vigor", ori_noise=None, circular=True, wseed=0, pseed=1234,
                        batch=1, grd="vigor"),
    "vigor_prior0": dict(kind="vigor", ori_noise=0, circular=True, wseed=0, pseed=77,
                         batch=1, grd="vigor"),
    "vigor_prior180_fov180": dict(kind="vigor", ori_noise=180, circular=False, wseed=0,
                                  pseed=78, batch=1, grd="vigor_fov180"),
    "kitti": dict(kind="kitti", ori_noise=None, circular=False, wseed=1, pseed=5,
                  batch=1, grd="kitti"),
}


BLOCK_HW = (8, 12)                       # MBConv module fixtures: input [2, cin, 8, 12]
ZERO_PAD_BLOCKS = (0, 1, 3, 5, 8, 11, 15)  # subset stored for the non-circular encoder


TRAIN_CASE = dict(kind="vigor", circular=True, wseed=0, pseed=2024, batch=2, grd="vigor")


def train_drop_masks(batch):
    """Deterministic drop_connect draws {(encoder, block): [B] 0/1} for the train-mode parity case and
    the matching per-sample scales mask/keep_prob (rate = 0.2*block/16, model.py:293-295)."""
    from ccvpe_amd import synth
    skip = [i for i, (k, s, e, cin, cout) in enumerate(synth.B0_BLOCKS) if s == 1 and cin == cout and i > 0]
    masks, scales = {}, {"grd_efficientnet": {}, "sat_efficientnet": {}}
    for enc, seed in (("grd_efficientnet", 1), ("sat_efficientnet", 2)):
        for i in skip:
            mk = (synth.uniform((batch,), 9000 + seed * 100 + i) > 0.3).float()
            masks[(enc, i)] = mk
            scales[enc][i] = mk / (1.0 - 0.2 * i / 16)
    return masks, scales, skip


RUNNING_STAT_SAMPLES = ("grd_efficientnet._bn0", "grd_efficientnet._blocks.3._bn1", "grd_efficientnet._blocks.15._bn2",
                        "sat_efficientnet._blocks.0._bn1", "sat_efficientnet._blocks.9._bn0", "sat_efficientnet._bn1")


def summarize_forward(out):
    """9-tuple -> dict of small numpy arrays (strided samples + arg-max facts)."""
    logits, heat, ori = out[0], out[1], out[2]
    B = logits.shape[0]
    lg = logits.reshape(B, 512, 512)
    d = {}
    d["logits_s4"] = lg[:, ::4, ::4]
    top = logits.topk(4, dim=1)
    d["top4_val"] = top.values
    d["top4_idx"] = top.indices
    d["heat_at_top4"] = heat.reshape(B, -1).gather(1, top.indices)
    d["heat_sum"] = heat.reshape(B, -1).double().sum(1)
    d["ori_s8"] = ori[:, :, ::8, ::8]
    d["ori_at_argmax"] = ori.reshape(B, 2, -1)[torch.arange(B), :, top.indices[:, 0]]
    for i, s in enumerate(out[3:], 1):
        step = max(1, s.shape[-1] // 32)
        d["score%d" % i] = s[:, :, ::step, ::step]
        d["score%d_mean" % i] = s.double().mean(dim=(2, 3))
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def assert_close(got, want, rtol, atol, what):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    if got.size == 1 and want.size == 1:
        got, want = got.reshape(()), want.reshape(())
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    assert not bad.any(), "%s: %d/%d out of tol, max err %.3e (ref max %.3e)" % (
        what, bad.sum(), bad.size, err.max(), np.abs(want).max())
