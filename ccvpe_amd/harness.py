"""Replica / data-parallel timing harness shared by bench.py and the gloo CPU test.

Inference shards by sample: every rank owns `batch_per_rank` independent image
pairs and runs the same step on them; there is NO data-path collective.  The only collectives
are the barrier that brackets the timed region and the MAX over ranks of the elapsed time.
`shard_indices` is the (deterministic) assignment of a global batch to ranks used by callers that
evaluate a fixed dataset.
"""
import time

import torch
import torch.distributed as dist


def shard_indices(n_items, world, rank):
    """Contiguous, balanced shard of range(n_items) for `rank` (first `n_items % world` ranks get
    one extra item).  Union over ranks == range(n_items), shards disjoint."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def timed_steps(step_fn, steps, warmup, sync_fn=None, device=None):
    """Run `warmup` untimed + exactly `steps` timed calls of step_fn, bracketed by
    barrier + device sync on both sides; returns the MAX elapsed seconds over ranks."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def fence():
        if distributed:
            dist.barrier()
        if sync_fn is not None:
            sync_fn()

    for _ in range(warmup):
        step_fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_throughput(units_per_rank_per_step, steps, elapsed_max):
    """Whole-job units/s: all ranks' units divided by the slowest rank's time."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return units_per_rank_per_step * world * steps / elapsed_max


def _avg_supported():
    """ReduceOp.AVG exists for the nccl (= RCCL) backend only; gloo gets SUM + an explicit scale."""
    return dist.get_backend() == "nccl" and __import__("os").environ.get("CCVPE_ALLREDUCE_AVG", "1") == "1"


def _allreduce_mean(flat, want_async=True):
    """In-place mean over the ranks of `flat`.  Returns (work handle, post-scale): RCCL averages inside the collective
    (ncclAvg: no extra pass over the data); elsewhere the caller applies `post-scale` after work.wait()."""
    world = dist.get_world_size()
    if _avg_supported():
        try:
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=want_async), 1.0
        except (RuntimeError, ValueError):           # a build without ncclAvg: fall back to SUM + scale
            pass
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=want_async), 1.0 / world


# backward production order of the model's single autograd node (ccvpe_amd/train.py: backward_train): the decoders, the
# matching blocks and the aerial descriptor first (~80 % of the parameter bytes), then the aerial encoder, then the ground
# descriptor heads + ground encoder
def grad_group(name):
    if name.startswith("sat_efficientnet."):
        return 1
    if name.startswith("grd_efficientnet.") or name.startswith("grd_feature_to_descriptor"):
        return 2
    return 0


_MULTI_COPY_MAX_FLOATS = 65536       # larger gradients go through torch's copy (bandwidth-bound anyway)


def _multi_copy(entries):
    """entries: (src ptr, dst ptr, floats) on the current device/stream -> ccvpe_multi_copy_f32 (~420 gradient vectors per
    training step, most of 16-1280 floats: one launch per 96 instead of one copy kernel each)."""
    import ctypes
    from . import _lib, ops
    lib = _lib.load()
    n = len(entries)
    srcs = (ctypes.c_void_p * n)(*[e[0] for e in entries])
    dsts = (ctypes.c_void_p * n)(*[e[1] for e in entries])
    cnts = (ctypes.c_int * n)(*[e[2] for e in entries])
    _lib.check(lib.ccvpe_multi_copy_f32(srcs, dsts, cnts, n, ops._stream()), "ccvpe_multi_copy_f32")


class GradientAllReducer:
    """Data-parallel training (BASELINE C3): average the parameter gradients over the ranks.

    One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI.  xGMI is point-to-point and a ring
    all-reduce is bound per link, so the ~58 M-parameter model travels as a few LARGE flat messages.

    Two modes:
      * attach(model) — the FLAT GRADIENT ARENA (what bench.py and a training script should use): one flat fp32 buffer holds
        every parameter's gradient, laid out in the order the model's backward produces them (three contiguous groups).
        The backward writes each gradient into its slot (that copy replaces the layout-fixing `.contiguous()` it did
        anyway), the slice of a finished group is all-reduced IN PLACE with ncclAvg while the rest of the backward still
        runs, and `p.grad` is a view of the arena: no torch.cat, no unpack, no separate scaling pass.
      * called after loss.backward() without attach (any model / any autograd graph): gradients are packed into fixed
        flat buckets (default 64 MiB), reduced asynchronously back to back and copied back.
    The layouts are fixed by parameter order / name, so every rank reduces the same bytes in the same order."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            nbytes = p.numel() * 4
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = None
        self._arena = None
        self._done_in_backward = False
        self.allreduce_calls = 0              # collectives issued so far (tests / diagnostics / bench.py's config.collective)
        self.bytes_reduced_last_step = 0      # payload bytes handed to all-reduce by the last arena step

    # ---- arena mode: reduce inside the model's backward -------------------------------------------------
    def attach(self, model, optimizer=None, step_in_backward=False):
        """Build the flat gradient arena for `model` (a ccvpe_amd CVM_* module) and register with it: its backward then
        calls begin() / ready() (after each of the three gradient groups) / finish().  After finish() every `p.grad` holds
        the MEAN over the ranks, whatever the backend: RCCL averages inside the collective (ncclAvg); a backend that can
        only SUM (gloo, a build without ncclAvg) gets one in-place 1/world pass over the reduced slice (~230 MB, once per
        step) — the scale is NOT deferred to the optimizer, so clipping / logging between backward() and step() see the
        same numbers on every backend.
        step_in_backward=True (needs a ccvpe_amd.optim.Adam as `optimizer`): finish() runs `optimizer.step_subset()` on each
        gradient group as soon as that group's all-reduce has completed, in issue order — the update of the early (large)
        groups then runs beside the collective of the last one (the ground encoder's 16 MB), which would otherwise be exposed
        in front of a whole-model optimizer step; the caller's `optimizer.step()` after backward() skips what was updated.
        Same arithmetic and bit-identical parameters (tests/test_rccl_single_rank_gpu.py).  Off by default: a training loop that
        clips or inspects gradients between backward() and step() needs them untouched.  Without it `optimizer` is not touched."""
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and "._fc." not in n]
        named.sort(key=lambda np_: grad_group(np_[0]))           # stable: keeps parameter order inside a group
        off, slots, bounds = 0, {}, [[None, None] for _ in range(3)]
        for n, p in named:
            g = grad_group(n)
            if bounds[g][0] is None:
                bounds[g][0] = off
            slots[n] = (off, p.numel(), tuple(p.shape), g, p)
            off += (p.numel() + 3) // 4 * 4                       # every slot 16-byte aligned
            bounds[g][1] = off
        dev = named[0][1].device
        self._arena = dict(flat=torch.zeros((off,), device=dev, dtype=torch.float32), slots=slots, bounds=bounds,
                           views={n: None for n in slots})
        for n, (o, cnt, shp, g, p) in slots.items():
            self._arena["views"][n] = self._arena["flat"][o:o + cnt].view(shp)
        self._optimizer = optimizer
        self._step_in_backward = bool(step_in_backward)
        if self._step_in_backward and not hasattr(optimizer, "step_subset"):
            raise ValueError("step_in_backward needs an optimizer with step_subset() (ccvpe_amd.optim.Adam)")
        model._grad_sync = self
        return self

    @staticmethod
    def active():
        """Collectives run when the job has more than one rank — or, with CCVPE_ALLREDUCE_SINGLE_RANK=1, also on a 1-rank
        process group (the GPU boxes of the test pool have one GPU: the real RCCL path still runs end to end)."""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or __import__("os").environ.get("CCVPE_ALLREDUCE_SINGLE_RANK") == "1"

    def arena_ok(self):
        """The arena path needs FRESH gradients: every p.grad is None (`zero_grad(set_to_none=True)`, torch's default).
        Any live .grad — including this arena's own view from the previous step, which may or may not have been zeroed —
        means the caller may be accumulating over several backward passes: that step goes through autograd's own
        accumulation and the caller's `reducer()` after backward() averages it with the bucketed path (no overlap).
        begin_fallback() is what the backward calls in that case."""
        if self._arena is None:
            return False
        for n, (o, cnt, shp, g, p) in self._arena["slots"].items():
            if p.grad is not None:
                return False
        return True

    def begin_fallback(self):
        """The backward of this step did NOT reduce (arena bypassed): the next reducer() call must do the work."""
        self._done_in_backward = False

    def begin(self):
        # step_in_backward: the previous backward's in-backward updates are only "consumed" by optimizer.step(); a second
        # backward without it (gradient accumulation, a step skipped after an inf / nan check) would find every parameter
        # marked as already stepped — its step_subset() a silent no-op and the following step() skipping everything
        opt = getattr(self, "_optimizer", None)
        if getattr(self, "_step_in_backward", False) and opt is not None and getattr(opt, "_pre_stepped", None):
            raise RuntimeError("GradientAllReducer(step_in_backward=True): optimizer.step() was not called since the last "
                               "in-backward update; gradient accumulation / skipped steps are not supported in this mode")
        self._pending, self._sent = [], set()
        self._done_in_backward = False
        self._group = 0
        self.bytes_reduced_last_step = 0

    def ready(self, grads):
        """Called by the backward after each gradient group with the dict of gradients finished so far.  Arena mode:
        copy the new ones into their slots and start the group's in-place all-reduce.  Without an arena (legacy callers
        with plain dicts): concatenate the new ones and reduce that."""
        names = [n for n in grads if n not in self._sent and grads[n] is not None]
        if not names:
            return
        self._sent.update(names)
        if self._arena is None:
            shapes = [tuple(grads[n].shape) for n in names]
            flat = torch.cat([grads[n].reshape(-1) for n in names])
            if self.active():
                work, scale = _allreduce_mean(flat)
                self.allreduce_calls += 1
            else:
                work, scale = None, 1.0
            self._pending.append((work, scale, flat, names, shapes))
            return
        ar = self._arena
        glo, ghi = None, None
        small = []                                           # (src ptr, dst ptr, floats): one multi-copy launch per 96
        in_arena = []
        for n in names:
            if n not in ar["slots"]:
                continue                                     # frozen / excluded parameter: stays with autograd
            in_arena.append(n)
            o, cnt, shp, g, p = ar["slots"][n]
            gr, view = grads[n], ar["views"][n]
            if (cnt <= _MULTI_COPY_MAX_FLOATS and gr.is_cuda and gr.dtype == torch.float32 and gr.is_contiguous()
                    and gr.numel() == cnt):
                small.append((gr.data_ptr(), view.data_ptr(), cnt))
            else:
                view.copy_(gr.reshape(shp))
            glo = g if glo is None else min(glo, g)
            ghi = g if ghi is None else max(ghi, g)
        if small:
            _multi_copy(small)
        if glo is None:
            return
        lo, hi = ar["bounds"][glo][0], ar["bounds"][ghi][1]
        if self.active():
            work, scale = _allreduce_mean(ar["flat"][lo:hi])
            self.allreduce_calls += 1
            self.bytes_reduced_last_step += (hi - lo) * 4
            self._pending.append((work, scale, ar["flat"][lo:hi], None, in_arena))
        elif getattr(self, "_step_in_backward", False):
            self._pending.append((None, 1.0, ar["flat"][lo:hi], None, in_arena))      # no collective: the group is final as it is

    def finish(self, grads):
        """Wait for the collectives; arena mode: hand the arena views to the parameters as their .grad and drop those
        entries from `grads` (nothing is returned to autograd for them)."""
        for work, scale, flat, names, shapes in self._pending:
            if work is not None:
                work.wait()
            if scale != 1.0:
                flat.mul_(scale)                         # SUM backends: p.grad is the mean here too (see attach)
            if names is None and getattr(self, "_step_in_backward", False) and shapes:
                # `shapes` carries the arena names of this group: their gradients are final -> update them now
                ar = self._arena
                group = []
                for n in shapes:
                    ar["slots"][n][4].grad = ar["views"][n]
                    group.append(ar["slots"][n][4])
                self._optimizer.step_subset(group)
            if names is not None:
                off = 0
                for n, shp in zip(names, shapes):
                    cnt = 1
                    for d in shp:
                        cnt *= d
                    grads[n] = flat[off:off + cnt].view(shp)
                    off += cnt
        self._pending = []
        if self._arena is not None:
            ar = self._arena
            for n in list(grads):
                if n in ar["slots"] and n in self._sent:
                    ar["slots"][n][4].grad = ar["views"][n]
                    grads[n] = None
        self._done_in_backward = True

    def __call__(self):
        if self._done_in_backward:      # already averaged inside the backward of this step
            self._done_in_backward = False
            return
        if not self.active():
            return
        if self._flat is None:
            self._flat = [torch.empty((sum(p.numel() for p in b),), device=b[0].device, dtype=torch.float32)
                          for b in self.buckets]
        works = []
        for flat, bucket in zip(self._flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            works.append(_allreduce_mean(flat))
            self.allreduce_calls += 1
        for (w, scale), flat, bucket in zip(works, self._flat, self.buckets):
            w.wait()
            if scale != 1.0:
                flat.mul_(scale)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n


class _GlobalRatio(torch.autograd.Function):
    """loss = (sum over ranks of num_r) / (sum over ranks of den_r) for a per-rank loss given as num_r / den_r.

    infoNCELoss (losses.py:4-20) is such a ratio: -sum(log p * label) over the batch / sum(label) over the batch.  Under
    data parallelism each rank only sees its own samples; averaging per-rank ratios is NOT the single-process big-batch
    loss.  Forward: ONE all-reduce of the two scalars (num_r = loss_r * den_r, den_r), returns the global ratio on every
    rank.  Backward: the gradient averaging of data parallelism computes (1/W) sum_r dL_r, so rank r must backpropagate
    W * den_r / DEN times its local ratio's gradient — then the averaged parameter gradients equal the big-batch ones."""

    @staticmethod
    def forward(ctx, loss_local, den_local, group):
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        den = den_local.detach().reshape(())
        # a rank whose shard carries no label mass has loss_local = 0/0: its numerator is 0, not NaN * 0
        num = torch.where(den > 0, loss_local.detach().reshape(()) * den, torch.zeros_like(den))
        pair = torch.stack([num, den])
        if world > 1:
            dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
        ctx.save_for_backward(den * float(world) / pair[1])
        return pair[0] / pair[1]

    @staticmethod
    def backward(ctx, g):
        (k,) = ctx.saved_tensors          # 0 on a rank without label mass: its local ratio (0/0) contributes nothing
        return g * k, None, None


def global_ratio_loss(loss_local, den_local, group=None):
    """See _GlobalRatio.  loss_local: this rank's ratio loss (differentiable); den_local: its denominator (label mass)."""
    return _GlobalRatio.apply(loss_local, den_local, group)
