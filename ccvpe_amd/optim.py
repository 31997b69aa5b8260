"""Adam for the training step on the MI355X: torch.optim.Adam's arithmetic (train_VIGOR.py:104 / train_KITTI.py:
`torch.optim.Adam(params, lr, betas=(0.9, 0.999))`, eps 1e-8, no weight decay, no amsgrad) with ONE kernel launch for
all ~520 parameter tensors (ccvpe_adam_step_f32: a device table of (param, grad, exp_avg, exp_avg_sq, numel) rows and
a workgroup -> (tensor, chunk) map).  Same constructor / step() / zero_grad() / state_dict() surface as the torch class
for the options the reference uses; the state dict is in torch's format, so checkpoints interchange."""
import numpy as np
import torch

from . import _lib, ops
from ._lib import check


class Adam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("ccvpe_amd.optim.Adam needs contiguous fp32 parameters on the MI355X (no CPU fallback)")
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)]
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.steps = [0 for _ in self.params]          # per parameter, as torch: a tensor without a gradient does not step
        lib = _lib.load()
        chunk = lib.ccvpe_adam_chunk_elems()
        ct, co = [], []
        for t, p in enumerate(self.params):
            n = (p.numel() + chunk - 1) // chunk
            ct.append(np.full((n,), t, dtype=np.int32))
            co.append(np.arange(n, dtype=np.int32))
        dev = self.params[0].device
        self._chunk_tensor = torch.from_numpy(np.concatenate(ct)).to(dev)
        self._chunk_off = torch.from_numpy(np.concatenate(co)).to(dev)
        self._table_host = np.zeros((len(self.params), 5), dtype=np.int64)
        for t, p in enumerate(self.params):
            self._table_host[t] = (p.data_ptr(), 0, self.exp_avg[t].data_ptr(), self.exp_avg_sq[t].data_ptr(), p.numel())

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        g = self.param_groups[0]
        keep = []                                       # contiguous copies must outlive the launch
        stepped = None
        for t, p in enumerate(self.params):
            gr = p.grad
            if gr is None:
                self._table_host[t, 1] = 0
                continue
            if not gr.is_contiguous() or gr.dtype != torch.float32:
                gr = gr.contiguous().float()
                keep.append(gr)
            self._table_host[t, 0] = p.data_ptr()
            self._table_host[t, 1] = gr.data_ptr()
            self.steps[t] += 1
            if stepped is None:
                stepped = self.steps[t]
            elif stepped != self.steps[t]:
                raise RuntimeError("ccvpe_amd.optim.Adam: parameters with different step counts in one step() "
                                   "(a parameter started receiving gradients later) are not supported")
        if stepped is None:
            return
        table = torch.from_numpy(self._table_host).to(self.params[0].device)
        check(lib.ccvpe_adam_step_f32(ops._ptr(table), ops._ptr(self._chunk_tensor), ops._ptr(self._chunk_off),
                                      self._chunk_tensor.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                      float(g["eps"]), stepped, ops._stream()), "ccvpe_adam_step_f32")
        _lib.weights_epoch += 1          # parameters changed without a torch version bump: invalidate packed weights
        table.record_stream(torch.cuda.current_stream())
        for k in keep:
            k.record_stream(torch.cuda.current_stream())

    # torch.optim.Adam-compatible state dict -------------------------------------------------------------
    def state_dict(self):
        state = {}
        for t in range(len(self.params)):
            if self.steps[t]:
                state[t] = dict(step=torch.tensor(float(self.steps[t])), exp_avg=self.exp_avg[t], exp_avg_sq=self.exp_avg_sq[t])
        groups = [dict(self.param_groups[0], params=list(range(len(self.params))))]
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        for t, st in sd["state"].items():
            t = int(t)
            self.exp_avg[t].copy_(st["exp_avg"])
            self.exp_avg_sq[t].copy_(st["exp_avg_sq"])
            self.steps[t] = int(st["step"])
        g = sd["param_groups"][0]
        self.param_groups[0].update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"])
