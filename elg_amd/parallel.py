"""Data parallelism over the instance batch: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm, over xGMI), one flat fp32 gradient bucket all-reduced per step.

The reference has no distributed code at all; instances never interact in the forward pass and the loss is
a mean over instances, so every rank rolls out its own shard and only the 5 MB gradient is exchanged.

Collectives on device memory are issued on the `nccl` backend only.  On any other backend (gloo: the CPU tests and
the two-ranks-on-one-GPU test) the flat buffer is staged through pinned host memory and the collective runs on the
host copy -- gloo's own device path is not exercised anywhere in this package."""
from __future__ import annotations

import datetime
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def respect_cpu_quota(cgroup_root: str = "/sys/fs/cgroup") -> int:
    """Cap torch's intra-op thread pool at the container's CPU quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`).  torch sizes
    the pool by the machine's CPU count; in a container limited to fewer CPUs every CPU tensor op above 32 768 elements then wakes
    far more threads than may run, their spin-waiting exhausts the quota and the kernel stalls the whole process until the next
    period (measured on the GPU box: 128 threads, quota 16, 40 - 60 ms stalls in a training loop).  Called by the
    entry points (train.py, the testers, bench.py), not at import.  Returns the thread count in force."""
    quota = None
    try:
        with open(os.path.join(cgroup_root, "cpu.max")) as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")) as f, open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")) as g:
                q, period = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, int(quota) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))      # the node's ranks share the quota
        if n < torch.get_num_threads():
            torch.set_num_threads(n)
    return torch.get_num_threads()


def world_info():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def force_group() -> bool:
    """ELG_FORCE_DIST=1: build the process group and run the gradient all-reduce even at world size 1 (the RCCL branch
    of a training step on a single GPU: `python -m torch.distributed.run --nproc-per-node 1 ...`)."""
    return os.environ.get("ELG_FORCE_DIST", "0") not in ("", "0")


def active() -> bool:
    """True when a training step has to go through the gradient bucket."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_group())


def init_distributed(backend: Optional[str] = None, timeout_s: float = 600.0) -> tuple:
    rank, world, local = world_info()
    if (world > 1 or force_group()) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        one_node = (os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1")
                    or os.environ.get("LOCAL_WORLD_SIZE") == str(world))
        if backend == "gloo" and one_node:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: never go looking for the host's name
            # (a multi-node gloo run must reach the other hosts: binding to loopback there would hang until the timeout)
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)          # eager communicator: a bad RCCL setup fails here
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, world, local


def _device_collectives() -> bool:
    return (not dist.is_initialized()) or dist.get_backend() == "nccl"


class _HostStage:
    """Pinned mirror of a flat device buffer for backends without a device path."""

    def __init__(self, numel: int, dtype):
        self.buf = torch.empty(numel, dtype=dtype, pin_memory=torch.cuda.is_available())

    def down(self, t: torch.Tensor) -> torch.Tensor:
        self.buf.copy_(t, non_blocking=True)
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        return self.buf

    def up(self, t: torch.Tensor):
        t.copy_(self.buf, non_blocking=True)


def all_reduce_sum(flat: torch.Tensor, stage: Optional[_HostStage] = None):
    """In-place SUM over the ranks of a flat buffer.  nccl: on the buffer itself, ordered on the current stream by
    ProcessGroupNCCL (its collective waits for the current stream, and the current stream waits for the collective)."""
    if not flat.is_cuda or _device_collectives():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return
    stage = stage or _HostStage(flat.numel(), flat.dtype)
    dist.all_reduce(stage.down(flat), op=dist.ReduceOp.SUM)
    stage.up(flat)


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
        self._stage = None
        self.calls = 0                        # all-reduces issued (bench.py / tests: proof that the branch ran)

    def _reduce(self):
        if self.flat.is_cuda and not _device_collectives() and self._stage is None:
            self._stage = _HostStage(self.numel, self.flat.dtype)
        all_reduce_sum(self.flat, self._stage)
        self.calls += 1

    def allreduce(self, world: int):
        if world <= 1 and not force_group():
            return
        if self.optimizer is not None:
            self.optimizer.gather_grads()
            self._reduce()
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
        self._reduce()
        self.flat.mul_(1.0 / world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def make_bucket(params, optimizer=None) -> Optional[GradBucket]:
    """The bucket a training loop needs: None for a plain single-process run."""
    return GradBucket(params, optimizer) if active() else None


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Rank `src`'s parameters and buffers on every rank: ONE broadcast of the packed values."""
    if not active():
        return
    ts = [t.data for t in list(module.parameters()) + list(module.buffers())]
    if not ts:
        return
    for dtype in sorted({t.dtype for t in ts}, key=str):
        group = [t for t in ts if t.dtype == dtype]
        flat = torch.cat([t.reshape(-1) for t in group])
        if flat.is_cuda and not _device_collectives():
            host = flat.cpu()
            dist.broadcast(host, src=src)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def broadcast_object(obj, src: int = 0):
    """The same Python object on every rank (curriculum choice, validation costs); identity without a process group."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return obj


def collective_world() -> tuple:
    """(rank, world) of the process group the collectives run in; (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def sum_over_ranks(values) -> list:
    """Element-wise SUM over the ranks of a short list of numbers (float64; identity without a process group)."""
    _, world = collective_world()
    if world <= 1:
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device="cuda" if _device_collectives() else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu()]


def guarded(fn):
    """Run fn() -- rank-local work WITHOUT collectives (a rank's share of a sharded validation) -- on every rank, then exchange
    an error flag: if ANY rank raised, every rank raises here, before the collective that would combine the results -- a rank
    that fails must not leave the others sitting in that collective until its timeout (600 s).  The failing rank re-raises its
    own exception, the others a RuntimeError naming it."""
    _, world = collective_world()
    if world <= 1:
        return fn()
    err, out = None, None
    try:
        out = fn()
    except BaseException as e:          # noqa: BLE001 -- the flag must be exchanged whatever went wrong
        err = e
    box = [None] * world
    dist.all_gather_object(box, None if err is None else f"{type(err).__name__}: {err}")
    if err is not None:
        raise err
    bad = [(r, m) for r, m in enumerate(box) if m is not None]
    if bad:
        raise RuntimeError(f"rank {bad[0][0]} failed in a collective phase: {bad[0][1]}")
    return out


def ranks_seen() -> int:
    """World size as the data path sees it: an all-reduce of ones (1 without a process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    one = torch.ones(1, device="cuda" if _device_collectives() else "cpu")
    dist.all_reduce(one)
    return int(one.item())


def barrier():
    if dist.is_initialized() and (dist.get_world_size() > 1 or force_group()):
        if _device_collectives():
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
