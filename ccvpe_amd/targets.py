"""Device-side construction of the training ground truth: what datasets.py:145-166 (VIGOR) /
:470-501 (KITTI) build with numpy per sample and train_VIGOR.py:114-128 copies to the GPU and max-pools — 24 MB per
sample — is generated on the MI355X from three scalars per sample (ccvpe_train_targets_f32)."""
import torch

from . import _lib, ops
from ._lib import check


def train_targets(center_xy, angle_deg, n_bins, height=512, width=512, sigma=4.0, ascending=False):
    """center_xy [B,2] = (cx, cy): the Gaussian sits where x_j = -W/2 + cx + j*W/(W-1) and y_i = -H/2 + cy + i*H/(H-1)
    vanish (VIGOR: cx = col_offset, cy = -row_offset; KITTI: cx = x_offset, cy = y_offset; Oxford: cx = col_offset_resized,
    cy = row_offset_resized).  angle_deg [B] in [0,360).  ascending: the Oxford loader's bin order (datasets.py:340-347).
    Returns (gt [B,1,H,W], gt_flattened [B,H*W] normalised to sum 1, gt_orientation [B,2,H,W],
             [gt_bottleneck1..6]: [B,n_bins,H/k,W/k] for k = 64,32,16,8,4,2)."""
    lib = _lib.load()
    center_xy = center_xy.contiguous().float()
    angle_deg = angle_deg.contiguous().float()
    ops._chk(center_xy, "center_xy")
    ops._chk(angle_deg, "angle_deg")
    b, dev = center_xy.shape[0], center_xy.device
    gt = torch.empty((b, 1, height, width), device=dev, dtype=torch.float32)
    gt_norm = torch.empty((b, height * width), device=dev, dtype=torch.float32)
    gt_ori = torch.empty((b, 2, height, width), device=dev, dtype=torch.float32)
    labs = [torch.empty((b, n_bins, height // k, width // k), device=dev, dtype=torch.float32) for k in (64, 32, 16, 8, 4, 2)]
    scratch = torch.empty((b * lib.ccvpe_train_targets_nblk(height, width),), device=dev, dtype=torch.float32)
    check(lib.ccvpe_train_targets_ordered_f32(ops._ptr(center_xy), ops._ptr(angle_deg), n_bins, int(bool(ascending)), float(sigma), ops._ptr(gt),
                                      ops._ptr(gt_norm), ops._ptr(gt_ori), *[ops._ptr(t) for t in labs], ops._ptr(scratch), b,
                                      height, width, ops._stream()), "ccvpe_train_targets_ordered_f32")
    return gt, gt_norm, gt_ori, labs
