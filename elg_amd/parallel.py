"""Data parallelism over the instance batch: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm, over xGMI), one flat fp32 gradient bucket all-reduced per step.

The reference has no distributed code at all; instances never interact in the forward pass and the loss is
a mean over instances, so every rank rolls out its own shard and only the 5 MB gradient is exchanged."""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def world_info():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(backend: Optional[str] = None) -> tuple:
    rank, world, local = world_info()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradBucket:
    """Flat fp32 bucket holding every parameter gradient; one all-reduce(SUM) + 1/world scale.

    With `optimizer` = elg_amd.optim.Adam the gradients are packed into its flat buffer by one batched copy, the
    all-reduce runs on that buffer, and the 1/world factor is folded into the Adam kernel (no scatter back)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], optimizer=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.optimizer = optimizer if hasattr(optimizer, "grad_flat") else None
        if self.optimizer is not None:
            assert self.optimizer.numel == self.numel
            self.flat = self.optimizer.grad_flat
        else:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)

    def allreduce(self, world: int):
        if world <= 1:
            return
        if self.optimizer is not None:
            self.optimizer.gather_grads()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.optimizer.grad_scale = 1.0 / world
            return
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def broadcast_object(obj, src: int = 0):
    """The same Python object on every rank (curriculum choice, validation costs); identity without a process group."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return obj


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
