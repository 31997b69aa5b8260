"""hipGraph-captured inference (BASELINE config C4: "hipGraph-captured inference").

The eval forward is a static chain of ~270 library launches on one stream with no host
synchronisation, no allocation outside torch's graph-private pool and no data-dependent shapes
(infoNCE's masked_select was rewritten as a masked sum for the same reason), so the whole
forward(grd, sat) captures into ONE hipGraph per (model, batch, ground shape).  Replay removes the
per-launch host cost (ctypes marshalling + hipLaunchKernel, ~10 us each); the kernels, their
order and their results are identical to the eager path (tests/test_forward_gpu.py checks
bit-equality).
"""
import torch


class GraphedForward(object):
    def __init__(self, net, grd, sat, warmup=2):
        if net.training:
            raise RuntimeError("GraphedForward captures the eval forward; call net.eval() first")
        self.net = net
        self.grd = grd.detach().clone().contiguous()
        self.sat = sat.detach().clone().contiguous()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):          # also packs the weights before capture
                net(self.grd, self.sat)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._wkey = (net.precision,) + net._weights_key()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = net(self.grd, self.sat)

    def __call__(self, grd, sat):
        """Copies the inputs into the captured buffers, replays, returns the captured outputs
        (static tensors: clone them if they must survive the next replay)."""
        if (self.net.precision,) + self.net._weights_key() != self._wkey:
            raise RuntimeError("model weights changed after capture; build a new GraphedForward")
        if grd.shape != self.grd.shape or sat.shape != self.sat.shape:
            raise ValueError("captured for %s / %s" % (tuple(self.grd.shape), tuple(self.sat.shape)))
        if grd.data_ptr() != self.grd.data_ptr():
            self.grd.copy_(grd)
        if sat.data_ptr() != self.sat.data_ptr():
            self.sat.copy_(sat)
        self.graph.replay()
        return self.out


class GraphedTrainStep(object):
    """hipGraph-captured training iteration up to, not including, the optimizer step — for the small-batch regime (the
    reference's default batch is 8, train_VIGOR.py:29), where a step is ~2 000 launches of a few microseconds each and the host
    cannot keep the device busy.

        step = GraphedTrainStep(lambda: loss_of(net(grd, sat), ...), net)                  # captures forward + losses + backward
        for it in range(n):
            grd_buf.copy_(next_grd); ...            # refill the tensors the closure reads (same storage every step)
            loss = step()                           # one graph launch: weight re-pack, forward, losses, backward
            opt.step()                              # eager: one Adam launch (its bias corrections are host numbers per step)

    What is captured: everything `loss_fn()` enqueues — the per-step weight re-pack (one gather launch, ccvpe_amd/repack.py),
    the train-mode forward on its streams, the losses, `loss.backward()` with its stream forks and joins, the BatchNorm
    running-statistic updates.  drop_connect draws come from torch's generator, which is graph-safe: every replay consumes fresh
    random numbers.  Gradients land in `p.grad` tensors of the graph's private pool at fixed addresses (the closure must not free
    them: this class clears them ONCE before capture).  Single process only: the data-parallel all-reduce hooks are host logic."""

    def __init__(self, loss_fn, net, warmup=2):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep: single-process training only")
        if not net.training:
            raise RuntimeError("GraphedTrainStep captures a training iteration; call net.train() first")
        self.net, self.params = net, list(net.parameters())
        self.loss_fn = loss_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                 # warms the allocator, builds the re-pack plan, the Adam tables, ...
                self._clear()
                loss_fn().backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._clear()
        # The capture must record the ONE-LAUNCH gather re-pack.  Without a verified plan (CCVPE_PACK_GATHER=0, or the plan failed
        # its bit-for-bit check) the forward would call graph.replay() of the re-pack hipGraph inside this capture — illegal, and
        # reported by HIP as an opaque capture error: say what is wrong instead.
        if getattr(net, "_pack_plan", None) is None:
            raise RuntimeError("GraphedTrainStep needs the one-launch gather re-pack (models._pack_plan); it is unavailable: "
                               + ("the plan failed verification" if getattr(net, "_pack_plan_failed", False)
                                  else "CCVPE_PACK_GATHER=0 or a non-fp32 model"))
        # the weights have not changed since the last warm-up step, so the forward would re-use its packed copies and the graph
        # would hold no re-pack: drop them — the capture then records the gather launch that re-derives them on every replay
        net.invalidate()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = loss_fn()
            self.loss.backward()
        self.loss = self.loss.detach()
        # the gradient tensors the graph writes to: p.grad must STAY these objects (see __call__ / zero_grad)
        self._static_grads = [p.grad for p in self.params]

    def zero_grad(self):
        """Zeroes the captured gradient tensors IN PLACE (never `optimizer.zero_grad(set_to_none=True)`: that detaches the
        static tensors the graph writes to; not needed between replays either — a replay overwrites every gradient)."""
        for g in self._static_grads:
            if g is not None:
                g.zero_()

    def _clear(self):
        for p in self.params:
            p.grad = None

    def __call__(self):
        """Replays forward + losses + backward on the tensors the closure captured; returns the (static) loss tensor.  The
        gradients are in `p.grad` (static tensors, overwritten by the next replay)."""
        self.graph.replay()
        for p, g in zip(self.params, self._static_grads):     # re-attach what a set_to_none zero_grad() dropped: the optimizer
            if p.grad is not g:                               # must step on the tensors this replay has just written
                p.grad = g
        return self.loss
