"""Adam for the training step (reference CVRP/train.py:101: torch.optim.Adam(lr, weight_decay=1e-6)) as ONE HIP launch.

torch.optim.Adam costs ~1 ms of host time per step for the ~110 parameter tensors of the model (the GPU idles behind
it).  Here the gradients autograd produced are packed into one flat buffer by a single batched copy (torch.cat), the
moments are flat, and csrc/elg_train.hip::adam_kernel updates the parameters where they are through a pointer
table -- parameter storage is never moved, so captured hipGraphs and state_dict round trips stay valid.  The flat
gradient buffer is also what the data-parallel all-reduce runs on (parallel.GradBucket).
`state_dict()` / `load_state_dict()` use torch.optim.Adam's layout, so optimizer checkpoints are interchangeable."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List

import torch

from . import _lib as L


class Adam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("elg_amd.optim.Adam runs on the GPU only (no CPU fallback)")
        torch.cuda.set_device(dev)                # the update kernel launches on the current device / its current stream
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("parameters must be contiguous fp32 tensors on one device")
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                             maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        self.param_groups = [dict(self.defaults, params=self.params)]
        offs = [0]
        for p in self.params:
            offs.append(offs[-1] + p.numel())
        self.numel = offs[-1]
        self._offsets = offs
        self.grad_flat = torch.zeros(self.numel, device=dev)
        self.exp_avg = torch.zeros(self.numel, device=dev)
        self.exp_avg_sq = torch.zeros(self.numel, device=dev)
        self._off_dev = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._table = torch.tensor([p.data_ptr() for p in self.params], dtype=torch.int64, device=dev)
        self._ptrs = [p.data_ptr() for p in self.params]
        self.step_count = 0
        self.grad_scale = 1.0                   # set to 1/world when gradients are summed across ranks
        self._gathered = False

    def zero_grad(self, set_to_none: bool = True):
        """Drop the gradients (autograd then hands its buffers over instead of launching an add per parameter)."""
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        self._gathered = False

    def gather_grads(self) -> torch.Tensor:
        """Pack every p.grad into the flat buffer (one batched copy); parameters without a gradient count as 0."""
        grads = [p.grad for p in self.params]
        if all(g is not None for g in grads):
            torch.cat([g.reshape(-1) for g in grads], out=self.grad_flat)
        else:
            self.grad_flat.zero_()
            for g, a, b in zip(grads, self._offsets[:-1], self._offsets[1:]):
                if g is not None:
                    self.grad_flat[a:b].copy_(g.reshape(-1))
        self._gathered = True
        return self.grad_flat

    @torch.no_grad()
    def step(self):
        if not self._gathered:
            self.gather_grads()
        self._gathered = False
        ptrs = [p.data_ptr() for p in self.params]
        if ptrs != self._ptrs:                  # a parameter was re-allocated (.to(), load with assign=True, ...)
            self._table = torch.tensor(ptrs, dtype=torch.int64, device=self.grad_flat.device)
            self._ptrs = ptrs
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        dev = self.grad_flat.device
        if dev.index != torch.cuda.current_device():
            raise RuntimeError(f"elg_amd.optim.Adam: parameters on {dev}, current device cuda:{torch.cuda.current_device()}")
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(L.lib().elg_adam_step(None, C.c_void_p(self._table.data_ptr()), C.c_void_p(self._off_dev.data_ptr()),
                                      len(self.params), C.c_void_p(self.grad_flat.data_ptr()),
                                      C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                      self.numel, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                      float(g["weight_decay"]), self.step_count, float(self.grad_scale), stream),
                "elg_adam_step")

    # ---- torch.optim.Adam-compatible checkpoints -------------------------------------------------------------
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, a, b) in enumerate(zip(self.params, self._offsets[:-1], self._offsets[1:])):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[a:b].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[a:b].view_as(p).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameters")
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in group:
                self.param_groups[0][k] = tuple(group[k]) if k == "betas" else group[k]
        steps = set()
        for i, (p, a, b) in enumerate(zip(self.params, self._offsets[:-1], self._offsets[1:])):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; not representable in the fused update")
        self.step_count = steps.pop() if steps else 0
