"""Sample-sharded evaluation loop (the caller side of the path: /root/reference/train_VIGOR.py:246-338,
train_KITTI.py:283-380), MI355X-style:

  * samples shard across ranks (harness.shard_indices) — inference has no data-path collective;
  * per batch: forward -> device-side post-processing (ccvpe_eval_postprocess_f32): 6 floats per
    sample come back instead of the reference's D2H copy of the full 512x512 heat-map and
    orientation field (3 MB per sample, train_VIGOR.py:288-289);
  * per-sample metrics exactly as the reference computes them on the host
    (train_VIGOR.py:294-324): pixel distance of the arg-max to the ground truth, metres via the
    per-sample ground resolution, orientation error min(|d|, 360-|d|);
  * one all_gather of the [n,4] result rows at the end (RCCL on GPUs, gloo in the CPU test), rank 0
    reports mean / median (train_VIGOR.py:330-336).

`forward_fn(grd, sat)` returns the 9-tuple; `postprocess_fn(heatmap, ori)` returns [B,6]
(y, x, cos, sin, angle_deg, prob) — ccvpe_amd.ops.eval_postprocess on the GPU.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import harness


def _angle_from_cos_sin(c, s):
    """train_VIGOR.py:317-322 (ground-truth side of the orientation error)."""
    a = math.degrees(math.acos(max(-1.0, min(1.0, c))))
    return (-a) % 360 if s < 0 else a


def sample_metrics(post_row, gt_yx, gt_cos_sin, metres_per_pixel):
    """(pixel_distance, metre_distance, orientation_error or NaN, prob) for one sample."""
    y, x, c, s, ang, prob = [float(v) for v in post_row]
    pd = math.sqrt((gt_yx[0] - y) ** 2 + (gt_yx[1] - x) ** 2)                  # train_VIGOR.py:298
    md = pd * metres_per_pixel                                                  # :300-307
    oe = float("nan")
    if not math.isnan(ang):                                                     # :311 (|cos|,|sin| <= 1)
        a_gt = _angle_from_cos_sin(gt_cos_sin[0], gt_cos_sin[1])
        d = abs(a_gt - ang)
        oe = min(d, 360 - d)                                                    # :324
    return pd, md, oe, prob


def evaluate(forward_fn, postprocess_fn, samples, batch_size, device=None):
    """samples: sequence of dicts with keys grd [3,h,w], sat [3,512,512], gt_yx (y,x), gt_cos_sin,
    metres_per_pixel.  Returns (on every rank) a dict of global statistics + the gathered rows
    [n_samples,4] ordered by sample index."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    mine = harness.shard_indices(len(samples), world, rank)
    rows = []
    for i in range(0, len(mine), batch_size):
        idx = mine[i:i + batch_size]
        grd = torch.stack([samples[j]["grd"] for j in idx])
        sat = torch.stack([samples[j]["sat"] for j in idx])
        if device is not None:
            grd, sat = grd.to(device), sat.to(device)
        out = forward_fn(grd, sat)
        post = postprocess_fn(out[1], out[2]).cpu()                             # [b,6] only
        for k, j in enumerate(idx):
            s = samples[j]
            rows.append((j,) + sample_metrics(post[k], s["gt_yx"], s["gt_cos_sin"], s["metres_per_pixel"]))
    local = torch.tensor(rows, dtype=torch.float64).reshape(-1, 5)
    if distributed and world > 1:
        # shards differ by at most one sample: pad to the common length, gather, drop the padding
        nmax = (len(samples) + world - 1) // world
        buf = torch.full((nmax, 5), -1.0, dtype=torch.float64)
        buf[:local.shape[0]] = local
        if device is not None and dist.get_backend() == "nccl":
            buf = buf.to(device)
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        allrows = torch.cat([g.cpu() for g in gathered])
        allrows = allrows[allrows[:, 0] >= 0]
    else:
        allrows = local
    allrows = allrows[allrows[:, 0].argsort()]
    res = allrows[:, 1:].numpy()
    oe = res[:, 2][~np.isnan(res[:, 2])]
    return {
        "n": int(res.shape[0]),
        "mean_pixel_error": float(np.mean(res[:, 0])), "median_pixel_error": float(np.median(res[:, 0])),
        "mean_metre_error": float(np.mean(res[:, 1])), "median_metre_error": float(np.median(res[:, 1])),
        "mean_orientation_error": float(np.mean(oe)) if oe.size else float("nan"),
        "median_orientation_error": float(np.median(oe)) if oe.size else float("nan"),
        "mean_probability": float(np.mean(res[:, 3])),
        "rows": res,
    }
