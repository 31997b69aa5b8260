"""Adam for the training step on the MI355X: torch.optim.Adam's arithmetic (train_VIGOR.py:104 / train_KITTI.py:
`torch.optim.Adam(params, lr, betas=(0.9, 0.999))`, eps 1e-8, no weight decay, no amsgrad) with ONE kernel launch for
all ~520 parameter tensors (ccvpe_adam_step_f32: a device table of (param, grad, exp_avg, exp_avg_sq, numel) rows, a
per-tensor hyper-parameter row and a workgroup -> (tensor, chunk) map).

It IS a torch.optim.Optimizer: param groups (per-group lr / betas / eps), `state[p] = {step, exp_avg, exp_avg_sq}` in
torch.optim.Adam's format, the base class's validated state_dict() / load_state_dict() (checkpoints interchange with
torch.optim.Adam), add_param_group(), and torch LR schedulers all work.  Like torch, every parameter keeps its own step
count (a parameter that starts receiving gradients later gets its own bias correction).  Not supported (raises):
weight_decay, amsgrad, maximize, sparse gradients."""
import math

import numpy as np
import torch

from . import _lib, ops
from ._lib import check


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay or amsgrad:
            raise ValueError("ccvpe_amd.optim.Adam implements the reference's configuration: no weight_decay / amsgrad")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False))
        self._layout = None
        self._subset_layouts = {}    # tuple of parameter ids -> launch geometry of step_subset()
        self._pre_stepped = set()    # ids updated by step_subset() since the last step(): step() skips them once
        self.grad_scale = 1.0        # caller-set multiplier applied to every gradient inside the update kernel (e.g. loss scaling)

    # ---- static launch geometry: rebuilt only when the parameter set changes --------------------------------
    def _build_layout(self, params=None):
        lib = _lib.load()
        whole = params is None
        if whole:
            params = [p for g in self.param_groups for p in g["params"]]
        for p in params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("ccvpe_amd.optim.Adam needs contiguous fp32 parameters on the MI355X (no CPU fallback)")
        chunk = lib.ccvpe_adam_chunk_elems()
        ct, co = [], []
        for t, p in enumerate(params):
            n = (p.numel() + chunk - 1) // chunk
            ct.append(np.full((n,), t, dtype=np.int32))
            co.append(np.arange(n, dtype=np.int32))
        dev = params[0].device
        lay = dict(params=params, ids=tuple(id(p) for p in params),
                   chunk_tensor=torch.from_numpy(np.concatenate(ct)).to(dev),
                   chunk_off=torch.from_numpy(np.concatenate(co)).to(dev),
                   table=np.zeros((len(params), 5), dtype=np.int64),
                   hyper=np.zeros((len(params), lib.ccvpe_adam_hyper_floats()), dtype=np.float32))
        if whole:
            self._layout = lay
        return lay

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:                                     # lazy, as torch.optim.Adam._init_group
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lay = self._layout
        if lay is None or lay["ids"] != tuple(id(p) for g in self.param_groups for p in g["params"]):
            lay = self._build_layout()
        skip, self._pre_stepped = self._pre_stepped, set()
        self._launch(lay, skip)
        return loss

    @torch.no_grad()
    def step_subset(self, params):
        """Update ONLY `params` now (their .grad must be final) and remember them: the next step() skips them once.
        harness.GradientAllReducer(step_in_backward=True) calls this per gradient group as soon as the group's all-reduce has
        finished, so the optimizer of the early groups runs beside the collectives of the late ones and the last (small)
        all-reduce is not exposed in front of a whole-model update.  Same arithmetic, same per-parameter step counts as step()."""
        params = [p for p in params if p.grad is not None and id(p) not in self._pre_stepped]
        if not params:
            return
        key = tuple(id(p) for p in params)
        lay = self._subset_layouts.get(key)
        if lay is None:
            lay = self._subset_layouts[key] = self._build_layout(params)
        self._launch(lay, set())
        self._pre_stepped.update(key)

    def _group_of(self):
        return {id(p): g for g in self.param_groups for p in g["params"]}

    def _launch(self, lay, skip):
        lib = _lib.load()
        table, hyper = lay["table"], lay["hyper"]
        group_of = self._group_of()
        keep, any_grad = [], False
        for t, p in enumerate(lay["params"]):
            g = group_of[id(p)]
            lr, (b1, b2), eps = float(g["lr"]), g["betas"], float(g["eps"])
            if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("ccvpe_amd.optim.Adam: weight_decay / amsgrad / maximize are not implemented")
            gr = p.grad
            if gr is None or id(p) in skip:
                table[t, 1] = 0
                continue
            if gr.is_sparse:
                raise RuntimeError("ccvpe_amd.optim.Adam does not support sparse gradients")
            if not gr.is_contiguous() or gr.dtype != torch.float32:
                gr = gr.contiguous().float()
                keep.append(gr)
            st = self._state_of(p)
            st["step"] += 1
            k = float(st["step"])
            # bias corrections and 1 - beta in double on the host, as torch.optim.Adam does with Python floats
            hyper[t] = (lr / (1.0 - b1 ** k), b1, b2, 1.0 - b1, 1.0 - b2, eps, math.sqrt(1.0 - b2 ** k), 0.0)
            table[t] = (p.data_ptr(), gr.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            any_grad = True
        if not any_grad:
            return
        dev = lay["params"][0].device
        tab_d = torch.from_numpy(table).to(dev, non_blocking=False)
        hyp_d = torch.from_numpy(hyper).to(dev, non_blocking=False)
        check(lib.ccvpe_adam_step_f32(ops._ptr(tab_d), ops._ptr(hyp_d), ops._ptr(lay["chunk_tensor"]), ops._ptr(lay["chunk_off"]),
                                      lay["chunk_tensor"].numel(), float(self.grad_scale), ops._stream()), "ccvpe_adam_step_f32")
        _lib.weights_epoch += 1          # parameters changed without a torch version bump: invalidate packed weights
        cur = torch.cuda.current_stream()
        for k in keep + [tab_d, hyp_d]:
            k.record_stream(cur)
