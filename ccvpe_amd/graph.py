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
