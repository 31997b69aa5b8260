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
    # the remaining ori_prior shapes at FoV 360 (models.py:489-511): ori_noise = 180 -> 21 shifts (+-10 coincide when L = C),
    # ori_noise = 72 -> 9 shifts; score1 keeps its 20 channels for the orientation decoder
    "vigor_prior180": dict(kind="vigor", ori_noise=180, circular=True, wseed=0, pseed=79,
                           batch=1, grd="vigor"),
    "vigor_prior72": dict(kind="vigor", ori_noise=72, circular=True, wseed=0, pseed=80,
                          batch=1, grd="vigor"),
    "kitti": dict(kind="kitti", ori_noise=None, circular=False, wseed=1, pseed=5,
                  batch=1, grd="kitti"),
    # CVM_OxfordRobotCar: 154x231 ground image, centred matching window
    "oxford": dict(kind="oxford", ori_noise=None, circular=False, wseed=2, pseed=9,
                   batch=1, grd="oxford"),
}


BLOCK_HW = (8, 12)                       # MBConv module fixtures: input [2, cin, 8, 12]
ZERO_PAD_BLOCKS = (0, 1, 3, 5, 8, 11, 15)  # subset stored for the non-circular encoder


TRAIN_CASE = dict(kind="vigor", circular=True, wseed=0, pseed=2024, batch=2, grd="vigor")
# BASELINE config C3's model in .train(): CVM_KITTI (models.py:752-950), 256 x 1024 ground image, 16 rotation hypotheses
TRAIN_CASE_KITTI = dict(kind="kitti", circular=False, wseed=1, pseed=2025, batch=2, grd="kitti")
TRAIN_CASES = {"vigor": TRAIN_CASE, "kitti": TRAIN_CASE_KITTI}


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


def train_loss(out):
    """Deterministic scalar touching all nine outputs (golden gradients): soft-label cross entropy on the logits
    (losses.py:26-28 form) plus fixed pseudo-random linear functionals of the heat-map, the orientation map and the six
    matching-score volumes.  Plain torch, so it runs on the reference's outputs (CPU) and on ours (device) alike."""
    from ccvpe_amd import synth
    logits, heat, ori = out[0], out[1], out[2]
    dev = logits.device
    b = logits.shape[0]
    lab = synth.uniform(tuple(logits.shape), 7001) ** 30
    lab = (lab / lab.sum(1, keepdim=True)).to(dev)
    loss = -(lab * torch.log_softmax(logits, dim=1)).sum() / b
    loss = loss + 50.0 * (heat.reshape(b, -1) * synth.normal(tuple(logits.shape), 7002).to(dev)).sum()
    loss = loss + (ori * synth.normal(tuple(ori.shape), 7003).to(dev)).mean() * 10.0
    for i, s in enumerate(out[3:]):
        loss = loss + (s * synth.normal(tuple(s.shape), 7010 + i).to(dev)).mean() * 3.0
    return loss


GRAD_SAMPLE_MAX = 2048


def summarize_grads(named_grads):
    """{name: grad tensor or None} -> per-parameter L2 norms (all parameters) + strided samples of every tensor."""
    d = {}
    names, norms = [], []
    for n, g in named_grads:
        if g is None:
            continue
        g = g.detach().cpu().double().reshape(-1)
        names.append(n)
        norms.append(float(g.norm()))
        step = max(1, g.numel() // GRAD_SAMPLE_MAX)
        d["g:" + n] = g[::step][:GRAD_SAMPLE_MAX].float().numpy()
    d["names"] = np.array(names)
    d["norms"] = np.array(norms, dtype=np.float64)
    return d


# Parameters whose TRUE gradient is exactly zero, so that fp32 autograd returns round-off noise there:
#   conv1.2.bias       softmax / cross-entropy are invariant to a constant added to every logit;
#   most _bn2.bias     a per-channel constant is removed again by the batch-statistic BatchNorm that follows
# (checked with the oracle in float64: norms ~1e-17).  They are only required to stay at noise level.
def compare_grads(got, want, rel_l2=3e-2):
    """Relative-L2 comparison of two summarize_grads() dicts.  The reference's own fp32 gradients sit 0.25 % (median)
    to 2.5 % (worst tensor: early squeeze-excite layers) away from the float64 gradient of the same graph (B=2
    batch-statistic BatchNorm amplifies round-off), so 3 % per tensor is the resolution of this fixture.
    Returns (list of offending tensors, median relative error)."""
    assert [str(n) for n in got["names"]] == [str(n) for n in want["names"]], "set of parameters with a gradient differs"
    top = float(want["norms"].max())
    bad, rels = [], []
    for n, wn, gn in zip(want["names"], want["norms"], got["norms"]):
        n = str(n)
        w, g = want["g:" + n].astype(np.float64), got["g:" + n].astype(np.float64)
        if wn < 1e-6 * top or n == "conv1.2.bias":
            if gn > 1e-4 * top:
                bad.append((n, "noise-level gradient expected", float(wn), float(gn)))
            continue
        r = float(np.linalg.norm(w - g) / np.linalg.norm(w))
        rels.append(r)
        if not (r <= rel_l2 and abs(gn - wn) <= rel_l2 * wn):
            bad.append((n, r, float(wn), float(gn)))
    return bad, float(np.median(rels))


def grad_rel_errors(got, truth):
    """{name: relative L2 error of got's sample vs truth's} for the tensors whose true gradient is not noise."""
    top = float(truth["norms"].max())
    out = {}
    for n, tn in zip(truth["names"], truth["norms"]):
        n = str(n)
        if tn < 1e-6 * top or n == "conv1.2.bias":
            continue
        t, g = truth["g:" + n].astype(np.float64), got["g:" + n].astype(np.float64)
        out[n] = float(np.linalg.norm(t - g) / np.linalg.norm(t))
    return out


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
