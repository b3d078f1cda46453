"""Fused Adam for the NeRF parameter lists -- a drop-in for the `torch.optim.Adam(params=grad_vars,
lr=args.lrate, betas=(0.9, 0.999))` of model.py:52 / vanilla_dynamic/model.py:115:

* `param_groups[0]['lr']` is writable, as the reference's schedule does every step (train.py:247-251);
* `state_dict()` / `load_state_dict()` use torch.optim.Adam's layout, so `optimizer_state_dict` entries of the
  reference's checkpoints (train.py:260-265) load, and ours load into torch.optim.Adam;
* `step(max_grad_norm=...)` folds `torch.nn.utils.clip_grad_norm_` (vanilla_dynamic/train.py:273-274) into the step
  without a host synchronisation (the coefficient stays on the device).
One kernel launch per step for all tensors (nrf_adam_step)."""
import ctypes as C

import torch

from . import _lib
from .ops import _stream


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.param_groups = [{"params": self.params, "lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0,
                              "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                              "differentiable": False, "fused": None}]
        self.steps = [0] * len(self.params)        # torch.optim.Adam counts steps per parameter
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params]
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("nerf_amd.FusedAdam: parameters must be contiguous float32 GPU tensors")

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @staticmethod
    def _table(tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    @torch.no_grad()
    def step(self, max_grad_norm=None):
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not idx:
            return None
        grads = [self.params[i].grad if self.params[i].grad.is_contiguous() else self.params[i].grad.contiguous()
                 for i in idx]
        scale = None
        total_norm = None
        if max_grad_norm is not None:   # clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max=1)
            total_norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
            scale = torch.clamp(max_grad_norm / (total_norm + 1e-6), max=1.0).float()
        for i in idx:
            self.steps[i] += 1
        g = self.param_groups[0]
        n = len(idx)
        numel = (C.c_int64 * n)(*[self.params[i].numel() for i in idx])
        steps = (C.c_int64 * n)(*[self.steps[i] for i in idx])
        _lib.check(_lib.lib().nrf_adam_step(
            n, self._table([self.params[i] for i in idx]), self._table(grads),
            self._table([self.exp_avg[i] for i in idx]), self._table([self.exp_avg_sq[i] for i in idx]), numel,
            steps, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
            C.c_void_p(scale.data_ptr()) if scale is not None else None, _stream()), "nrf_adam_step")
        # the kernel wrote through raw pointers: tell autograd (and the packed-weight caches keyed on `_version`,
        # ops.PackedNet.refresh) that these tensors changed
        torch.autograd.graph.increment_version([self.params[i] for i in idx])
        return total_norm

    # ---- torch.optim.Adam-compatible (de)serialisation
    def state_dict(self):
        state = {}
        for i in range(len(self.params)):
            if self.steps[i] > 0:
                state[i] = {"step": torch.tensor(float(self.steps[i])), "exp_avg": self.exp_avg[i],
                            "exp_avg_sq": self.exp_avg_sq[i]}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        n = sum(len(g["params"]) for g in groups)
        if n != len(self.params):
            raise ValueError(f"loaded state dict has {n} parameters, this optimizer {len(self.params)}")
        self.param_groups[0]["lr"] = groups[0]["lr"]
        self.param_groups[0]["betas"] = tuple(groups[0].get("betas", self.param_groups[0]["betas"]))
        self.param_groups[0]["eps"] = groups[0].get("eps", self.param_groups[0]["eps"])
        self.steps = [0] * len(self.params)
        for i, st in sd["state"].items():
            i = int(i)
            self.exp_avg[i].copy_(st["exp_avg"])
            self.exp_avg_sq[i].copy_(st["exp_avg_sq"])
            self.steps[i] = int(float(st["step"]))


def make_adam(params, lr, betas=(0.9, 0.999)):
    """The optimizer create_nerf / create_dnerf hand out: the fused step for GPU parameters, torch.optim.Adam otherwise
    (CPU construction is only used to inspect / convert checkpoints)."""
    params = list(params)
    if params and all(p.is_cuda for p in params):
        return FusedAdam(params, lr=lr, betas=betas)
    return torch.optim.Adam(params=params, lr=lr, betas=betas)
