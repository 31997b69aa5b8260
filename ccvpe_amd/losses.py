"""Drop-in for the reference's losses.py (same function names and argument order), on the MI355X:
    infoNCELoss(scores, labels, temperature=0.1)      /root/reference/losses.py:4-20
    cross_entropy_loss(logits, labels)                /root/reference/losses.py:23-24
    orientation_loss(ori, gt_orientation, gt)         /root/reference/losses.py:28-29
Forward and backward are libccvpe_hip.so kernels (csrc/heads.hip, csrc/heads_bwd.hip); the gradient flows to the
prediction only (labels are data), as in the reference's use.  No CPU fallback.
"""
import torch

from . import _lib, ops
from ._lib import check


def _flat2(t):
    return t.reshape(t.shape[0], -1).contiguous().float()


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, labels, temperature):
        s, l = _flat2(scores), _flat2(labels)
        loss, den, rows = ops.infonce_loss(s, l, temperature, want_den=True, want_rows=True)
        ctx.save_for_backward(s, l, rows)
        ctx.temperature, ctx.shape = float(temperature), scores.shape
        den = den.clone()
        ctx.mark_non_differentiable(den)
        return loss.reshape(()), den.reshape(())

    @staticmethod
    def backward(ctx, dloss, _dden=None):
        s, l, rows = ctx.saved_tensors
        lib = _lib.load()
        out = torch.empty_like(s)
        g = dloss.reshape(1).contiguous().float()
        check(lib.ccvpe_infonce_loss_bwd_f32(ops._ptr(s), ops._ptr(l), ctx.temperature, ops._ptr(g), ops._ptr(rows),
                                             ops._ptr(out), s.shape[0], s.shape[1], ops._stream()),
              "ccvpe_infonce_loss_bwd_f32")
        return out.reshape(ctx.shape), None, None


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        s, l = _flat2(logits), _flat2(labels)
        ctx.save_for_backward(s, l)
        ctx.shape = logits.shape
        return ops.cross_entropy_loss(s, l).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        s, l = ctx.saved_tensors
        lib = _lib.load()
        out = torch.empty_like(s)
        g = dloss.reshape(1).contiguous().float()
        check(lib.ccvpe_cross_entropy_loss_bwd_f32(ops._ptr(s), ops._ptr(l), ops._ptr(g), ops._ptr(out), s.shape[0],
                                                   s.shape[1], ops._stream()), "ccvpe_cross_entropy_loss_bwd_f32")
        return out.reshape(ctx.shape), None


class _Orientation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ori, gt_orientation, gt):
        o, go, g = ori.contiguous().float(), gt_orientation.contiguous().float(), gt.contiguous().float()
        ctx.save_for_backward(o, go, g)
        return ops.orientation_loss(o, go, g).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        o, go, g = ctx.saved_tensors
        lib = _lib.load()
        out = torch.empty_like(o)
        d = dloss.reshape(1).contiguous().float()
        check(lib.ccvpe_orientation_loss_bwd_f32(ops._ptr(o), ops._ptr(go), ops._ptr(g), ops._ptr(d), ops._ptr(out),
                                                 o.shape[0], o.shape[2] * o.shape[3], ops._stream()),
              "ccvpe_orientation_loss_bwd_f32")
        return out, None, None


def infoNCELoss(scores, labels, temperature=0.1):
    return _InfoNCE.apply(scores, labels, temperature)[0]


def infoNCELoss_global(scores, labels, temperature=0.1, group=None):
    """infoNCELoss over the GLOBAL batch of a data-parallel job: losses.py:18 divides by the label mass of the whole
    batch, so the exact multi-GPU equivalent of the reference's single-process loss all-reduces (numerator, label mass)
    — two scalars per level — before the division (harness.global_ratio_loss).  Same value on every rank; with the
    data-parallel gradient averaging the parameter gradients equal those of one process holding all samples.  With one
    process it is infoNCELoss."""
    from .harness import global_ratio_loss
    loss, den = _InfoNCE.apply(scores, labels, temperature)
    return global_ratio_loss(loss, den, group)


def cross_entropy_loss(logits, labels):
    return _CrossEntropy.apply(logits, labels)


def orientation_loss(ori, gt_orientation, gt):
    return _Orientation.apply(ori, gt_orientation, gt)
