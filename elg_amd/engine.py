"""Host side of the MI355X ELG-POMO rollout engine: folds the decoder / local-policy weights into
the per-instance tables the HIP kernels consume, launches the kernels through the C ABI
(include/elg_hip.h) and provides the autograd glue for training.

PyTorch is plumbing here (device memory, streams, the small dense folds that autograd carries back
to the parameters); every per-step operation of the reference runs inside libelg_hip.so.
There is NO CPU fallback: without a GPU + the built library these functions raise."""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib as L

E, H, DK = 128, 8, 16
LE, LH, LDK = 32, 4, 8


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


_SIDE = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    key = str(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


# A training batch from the host generator depends on nothing the GPU has computed: its two uploads and the neighbour-table kernel
# are issued on their own stream, where they run while the PREVIOUS step's backward still occupies the main stream (the host is a step
# ahead of the GPU), instead of sitting between Adam and the encoder (two DMA round trips + one launch: ~55 us of a 4 - 5 ms step).
PREP_STREAM = os.environ.get("ELG_PREP_STREAM", "1") not in ("", "0")
_PREP = {}


def on_prep_stream(dev, fn):
    """fn() -> tuple of device tensors, built from HOST data only.  Runs fn on the device's preparation stream; the current stream
    waits for it and the tensors are marked as used there (the allocator must not hand their memory out again while the current
    stream still reads them)."""
    if not PREP_STREAM:
        return fn()
    dev = torch.device(dev)
    key = str(dev)
    if key not in _PREP:
        _PREP[key] = torch.cuda.Stream(device=dev)
    main, prep = torch.cuda.current_stream(dev), _PREP[key]
    with torch.cuda.stream(prep):
        outs = fn()
        ev = torch.cuda.Event()
        ev.record(prep)
    main.wait_event(ev)
    for t in outs:
        t.record_stream(main)
    return outs


class HostFetch:
    """A few device integers copied to pinned host memory behind the work queued so far; get() waits for that copy only,
    not for what was queued after it (a plain .tolist() would wait for the whole stream)."""

    def __init__(self, t: torch.Tensor):
        self.buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self.buf.copy_(t, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream(t.device))

    # seconds the host has spent BLOCKED on the device in the step's syncs (this fetch and the pinned staging buffers' reuse
    # events): wall time per step minus this is what the Python thread needs to enqueue a step (bench.py: host_ms_per_step)
    waited_s = 0.0

    def get(self) -> list:
        t0 = time.perf_counter()
        self.ev.synchronize()
        HostFetch.waited_s += time.perf_counter() - t0
        return self.buf.tolist()


def _wait_event(ev):
    t0 = time.perf_counter()
    ev.synchronize()
    HostFetch.waited_s += time.perf_counter() - t0


_PINNED = {}


def _host_copy(dst: torch.Tensor, src: torch.Tensor):
    """dst[...] = src between host tensors WITHOUT torch's intra-op thread pool (numpy: one thread).  A CPU tensor op over more than
    32 768 elements wakes all of torch's OpenMP threads (128 on the GPU box); in a container with a CPU quota (16 CPUs there) their
    spin-waiting exhausts the quota and the kernel throttles the whole process for the rest of the 100 ms period -- measured as a
    40 - 60 ms stall in every third training step at batches >= 180 (depot | customers = B x 101 x 2 > 32 768 floats), in whatever
    call the main thread happened to be; gone with OMP_NUM_THREADS=1 and with this copy.  tools/time_b120_geometry.py."""
    if dst.dtype == src.dtype and dst.dtype != torch.bfloat16:
        dst.numpy()[...] = src.detach().numpy()
    else:
        dst.copy_(src)


def h2d_parts(parts, dev, pad: int = 64):
    """Several host float32 tensors -> ONE pinned staging buffer (each part starts at a multiple of `pad` floats) -> ONE asynchronous
    copy -> device views of the parts' shapes.  The parts are written into the staging buffer by numpy slices (see _host_copy)."""
    offs, n = [], 0
    for p in parts:
        offs.append(n)
        n += (p.numel() + pad - 1) // pad * pad
    key = (("parts", n), torch.float32, str(dev))
    ent = _PINNED.get(key)
    if ent is None:
        ent = _PINNED[key] = [torch.zeros(n, dtype=torch.float32).pin_memory(), None]
    buf, ev = ent
    if ev is not None:
        _wait_event(ev)                         # the previous copy out of this buffer has completed
    host = buf.numpy()
    for p, o in zip(parts, offs):
        host[o:o + p.numel()] = p.detach().float().contiguous().numpy().reshape(-1)
    flat = buf.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ent[1] = ev
    return flat, [flat[o:o + p.numel()].view(p.shape) for p, o in zip(parts, offs)]


def h2d(t: torch.Tensor, dev) -> torch.Tensor:
    """Host -> HBM copy that does not stall the host behind the work already queued on the stream: the tensor is
    staged in a cached pinned buffer and copied asynchronously (a pageable .to(device) blocks until every kernel
    queued before it has finished, i.e. the previous step's backward, and the GPU then idles while the host
    enqueues the next step)."""
    if t.is_cuda:
        return t
    key = (tuple(t.shape), t.dtype, str(dev))
    ent = _PINNED.get(key)
    if ent is None:
        ent = _PINNED[key] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True), None]
    buf, ev = ent
    if ev is not None:
        _wait_event(ev)                         # the previous copy out of this buffer has completed
    _host_copy(buf, t)
    out = buf.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ent[1] = ev
    return out


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"elg_amd: {what} must live on the GPU -- the HIP path has no CPU fallback")
    # the C ABI launches on the calling thread's current HIP device and on the stream passed in: both must be the tensor's
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"elg_amd: {what} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device() first (CVRPEnv / TSPEnv / elg_amd.optim.Adam do it for their device)")


# ----------------------------------------------------------------------------------------------
# small ops
# ----------------------------------------------------------------------------------------------
def aug8(xy: torch.Tensor) -> torch.Tensor:
    """utils.augment_xy_data_by_8_fold (reference CVRP/utils.py:69-87): (B,N,2) -> (8B,N,2)."""
    _need_cuda(xy, "xy")
    xy = xy.contiguous().float()
    B, N, _ = xy.shape
    out = torch.empty(8 * B, N, 2, device=xy.device, dtype=torch.float32)
    L.check(L.lib().elg_aug8(_ptr(xy), _ptr(out), B, N, _stream()), "elg_aug8")
    return out


def dist_matrix(xy: torch.Tensor) -> torch.Tensor:
    _need_cuda(xy, "xy")
    xy = xy.contiguous().float()
    B, N, _ = xy.shape
    out = torch.empty(B, N, N, device=xy.device, dtype=torch.float32)
    L.check(L.lib().elg_dist_matrix(_ptr(xy), _ptr(out), B, N, _stream()), "elg_dist_matrix")
    return out


@dataclass
class NbrTables:
    idx: torch.Tensor      # (B,N,N) int32   neighbours of every node sorted by (dist, index)
    dist: torch.Tensor     # (B,N,N) f32
    theta: torch.Tensor    # (B,N,N) f32     atan2(y_n - y_c, x_n - x_c)


def nbr_tables(xy: torch.Tensor) -> NbrTables:
    _need_cuda(xy, "xy")
    xy = xy.contiguous().float()
    B, N, _ = xy.shape
    idx = torch.empty(B, N, N, device=xy.device, dtype=torch.int32)
    dist = torch.empty(B, N, N, device=xy.device, dtype=torch.float32)
    theta = torch.empty(B, N, N, device=xy.device, dtype=torch.float32)
    L.check(L.lib().elg_nbr_tables(_ptr(xy), _ptr(idx), _ptr(dist), _ptr(theta), B, N, _stream()), "elg_nbr_tables")
    return NbrTables(idx, dist, theta)


def route_length(xy: torch.Tensor, tour: torch.Tensor, rounding: bool = False) -> torch.Tensor:
    """Closed-tour length (reference CVRPEnv.py:251-288, TSPEnv.py:158-184).  xy (B,N,2) f32,
    tour (B,M,T) int64 -> (B,M) f32 (positive length)."""
    _need_cuda(xy, "xy")
    xy = xy.contiguous().float()
    tour = tour.contiguous().to(torch.int64)
    B, M, T = tour.shape
    out = torch.empty(B, M, device=xy.device, dtype=torch.float32)
    L.check(L.lib().elg_route_length(_ptr(xy), _ptr(tour), _ptr(out), B, M, T, xy.shape[1], int(bool(rounding)),
                                     _stream()), "elg_route_length")
    return out


def feasibility_flags_launch(pi: torch.Tensor, demand: Optional[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Enqueue elg_check_feasible for the tours `pi` (multi, T) int64 of ONE instance (the reference's utils.check_feasible as
    one launch; demand (problem,) of the customers, None for TSP) -> device flags [invalid_tour, over_capacity].  No host
    sync.  For CVRP, trailing depot visits (the padding of unfinished time steps) do not change either flag."""
    _need_cuda(pi, "pi")
    assert pi.dim() == 2 and pi.dtype == torch.int64 and pi.stride(1) == 1
    n = int(demand.numel()) if demand is not None else int(pi.shape[1])
    if demand is not None:
        demand = demand.contiguous().float()
    flags = out if out is not None else torch.zeros(2, dtype=torch.int32, device=pi.device)    # `out`: zeroed by the caller
    with torch.cuda.device(pi.device):
        L.check(L.lib().elg_check_feasible(_ptr(pi), pi.stride(0), _ptr(demand), pi.shape[0], pi.shape[1], n, _ptr(flags),
                                           _stream(pi.device)), "elg_check_feasible")
    return flags


def feasibility_flags(pi: torch.Tensor, demand: Optional[torch.Tensor]) -> tuple:
    """(invalid_tour, over_capacity) of the tours `pi` (feasibility_flags_launch + one host sync)."""
    bad, over = feasibility_flags_launch(pi, demand).tolist()
    return bool(bad), bool(over)


# ----------------------------------------------------------------------------------------------
# fp32 MFMA GEMM (encoder layers)
# ----------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a=False, trans_b=False, bias=None, relu=False, split_k=1,
         out: Optional[torch.Tensor] = None, a_rowsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] (+)= op(a) op(b) (+bias)(ReLU) through elg_gemm_f32 (v_mfma_f32_32x32x2_f32).  2-D fp32 inputs."""
    _need_cuda(a, "a")
    assert a.dim() == 2 and b.dim() == 2 and a.dtype == torch.float32 and b.dtype == torch.float32
    a, b = a.contiguous(), b.contiguous()
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    if out is None:
        out = torch.zeros(M, N, device=a.device) if split_k > 1 else torch.empty(M, N, device=a.device)
    L.check(L.lib().elg_gemm_f32(_ptr(a), _ptr(b), _ptr(out), _ptr(bias), M, N, K, a.shape[1], b.shape[1], N,
                                 int(trans_a), int(trans_b), int(relu), split_k, _ptr(a_rowsum), _stream()),
            "elg_gemm_f32")
    return out


# ----------------------------------------------------------------------------------------------
# weight folding (differentiable torch: autograd carries the kernel's table gradients back)
# ----------------------------------------------------------------------------------------------
# The local-policy row backward (elg_local_bwd_rows, 0.57 ms at the bench shape, one 464-register wave per SIMD) depends on the
# pointer backward only and feeds nothing but the fold backward below; the encoder's backward chain that follows the decoder
# backward is ~75 latency-bound launches that leave most of the chip idle.  With ELG_SIDE_LOCAL_BWD (default on) the row kernel
# runs on a side stream over a PART of the CUs (it cannot share a CU with anything: registers) next to that chain, and
# _FoldLocal.backward -- which autograd runs after the encoder's node, see CVRPModel.pre_forward -- waits for it.
SIDE_LOCAL_BWD = os.environ.get("ELG_SIDE_LOCAL_BWD", "1") not in ("", "0")
SIDE_LOCAL_BWD_GRID = int(os.environ.get("ELG_LOCAL_BWD_GRID", "0"))       # workgroups of the side-stream launch (0: half the CUs)
SIDE_LOCAL_MIN_TILES_PER_CU = 64         # below this many 16-row tiles per CU the row kernel is too short to be worth a second stream
_N_CUS: Dict[int, int] = {}


def n_cus(dev=None) -> int:
    """Compute units of the device (256 on an MI355X), cached."""
    idx = torch.cuda.current_device() if dev is None else torch.device(dev).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _N_CUS:
        _N_CUS[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return _N_CUS[idx]
_PENDING_SIDE: list = []        # (device index, event) of side-stream launches whose result the next _FoldLocal.backward consumes


def _drain_side(dev):
    """Make the current stream of `dev` wait for the side-stream launches still pending on that device.  Called by the consumer
    of their result (_FoldLocal.backward) and -- in case that node never ran (torch.autograd.grad over a subset of the inputs,
    an exception in between) -- before a training forward overwrites the rows / scratch the side launch reads."""
    idx = torch.device(dev).index
    if idx is None:
        idx = torch.cuda.current_device()
    keep = []
    for i, ev in _PENDING_SIDE:
        if i == idx:
            torch.cuda.current_stream(idx).wait_event(ev)
        else:
            keep.append((i, ev))
    _PENDING_SIDE[:] = keep


class _FoldLocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nfeat, n_slots, positional, We, be, c, Wq, Wk, Wv, Wc, bc):
        params = (We, be, c, Wq, Wk, Wv, Wc, bc)
        w = L.LocalWeights(*[C.c_void_p(p.data_ptr()) for p in params])
        loc = torch.empty(L.LOC_SIZE, device=We.device)
        with torch.cuda.device(We.device):
            L.check(L.lib().elg_local_fold_fwd(C.byref(w), nfeat, n_slots, int(positional), _ptr(loc), _stream(We.device)),
                    "elg_local_fold_fwd")
        ctx.save_for_backward(*params)
        ctx.meta = (nfeat, n_slots, int(positional))
        return loc

    @staticmethod
    def backward(ctx, gloc):
        params = ctx.saved_tensors
        nfeat, n_slots, positional = ctx.meta
        dev = params[0].device
        _drain_side(dev)                             # gloc was produced on the side stream
        w = L.LocalWeights(*[C.c_void_p(p.data_ptr()) for p in params])
        sizes = [p.numel() for p in params]
        flat = torch.empty(sum(sizes), device=dev)
        grads, off = [], 0
        for p, n in zip(params, sizes):
            grads.append(flat[off:off + n].view_as(p))
            off += n
        g = L.LocalWeights(*[C.c_void_p(t.data_ptr()) for t in grads])
        gloc = gloc.contiguous().float()
        with torch.cuda.device(dev):
            L.check(L.lib().elg_local_fold_bwd(C.byref(w), nfeat, n_slots, positional, _ptr(gloc), C.byref(g), _stream(dev)),
                    "elg_local_fold_bwd")
        return (None, None, None, *grads)


def fold_local_tables(lp: Dict[str, torch.Tensor], nfeat: int, n_slots: int, positional: bool = True) -> torch.Tensor:
    """Fold local_policy_att's projections into slot tables (layout: include/elg_hip.h ELG_LOC_*), one HIP launch each
    way (csrc/elg_fold.hip).  reference models.py:133-166: e_j = We f_j + be + PE[j]; q = Wq c; k_j = Wk e_j; v_j = Wv e_j;
    u_j = (Wc softmax(q k / sqrt 8) v + bc) . e_j / sqrt 32 -- everything that does not depend on the features f_j is
    precomputed per slot j.  `positional` = model_params['positional'] (models.py:142-143)."""
    names = ("init_emb.weight", "init_emb.bias", "cur_token_emb", "Wq.weight", "Wk.weight", "Wv.weight",
             "multi_head_combine.weight", "multi_head_combine.bias")
    params = [lp[n] for n in names]
    _need_cuda(params[0], "local policy parameters")
    for p in params:
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise ValueError("local policy parameters must be contiguous fp32 tensors")
    return _FoldLocal.apply(nfeat, n_slots, bool(positional), *params)


# ----------------------------------------------------------------------------------------------
# launch configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class Problem:
    """Device-resident description of one batch of instances."""
    kind: int                       # L.PROBLEM_*
    xy: torch.Tensor                # (B,N1,2)
    demand: Optional[torch.Tensor]  # (B,N1) CVRP
    nbr: NbrTables

    @property
    def B(self):
        return self.xy.shape[0]

    @property
    def N1(self):
        return self.xy.shape[1]


@dataclass
class Policy:
    """Folded weights of one (model, batch) pair."""
    tables: Dict[str, torch.Tensor]
    loc: Optional[torch.Tensor]
    K: int
    xi: float
    clip: float
    inv_ens: float
    has_local: bool
    has_penalty: bool
    euclidean: bool = False        # local features (x, y) / norm instead of (dist / norm, theta)
    Ks: tuple = ()                 # ensemble_size > 1: local_size of every member (Ks[0] == K); loc is (len(Ks), LOC_SIZE)

    @property
    def ens(self) -> int:
        return len(self.Ks) if len(self.Ks) > 1 else 1

    @property
    def wide_slots(self) -> bool:
        """local_size > 47: more than the 48 slots the matrix-core kernels and the saved training rows are built on -- the
        one-wavefront kernels run (one slot per lane, up to 64) and training goes through the replay backward."""
        return max((self.K,) + tuple(self.Ks)) > L.ROWS_LOCAL_SIZE


def coop_tiles(B: int, M: int, n_cu: int) -> int:
    """Workgroups per instance of the cooperative rollout kernel.  A workgroup (one per CU: the instance's tables fill its LDS) walks
    its ceil(M / tiles) trajectories in evenly sized groups of <= 32 lockstep trajectories, one whole rollout per group, so a launch
    takes  rounds x groups x (time of one group)  with rounds = ceil(B tiles / CUs); a group of <= 16 trajectories (one MFMA row
    tile) costs ~0.9 of a larger one.  Measured (training step, ms; tools/time_b120_geometry.py): B = 64: tiles 4 -> 5.19 (1 x 1),
    2 -> 6.89 (1 x 2), 7 -> 6.52 (2 x 1 x 0.9); B = 120, the reference's default batch: 3 -> 13.48 (what ceil(CUs / B) gave: 360
    workgroups of 34 = 2 groups of 17, two rounds), 2 or 4 -> 10.31.  The chip-filling choice ceil(CUs / B) stays whenever the model
    finds nothing strictly cheaper; among cheaper ones the fewest workgroups win (the tables are staged once per workgroup)."""
    def cost(t):
        traj = (M + t - 1) // t
        groups = (traj + 31) // 32
        gsize = (traj + groups - 1) // groups
        return ((B * t + n_cu - 1) // n_cu) * groups * (0.9 if gsize <= 16 else 1.0)
    best = max(1, min(M, (n_cu + B - 1) // B))
    best_cost = cost(best)
    for t in range(1, min(M, 64) + 1):
        if cost(t) < best_cost - 1e-9:
            best, best_cost = t, cost(t)
    return best


def launch_geometry(B: int, M: int, N1: int, n_cu: Optional[int] = None):
    """(waves, tiles, lds_stage).  One workgroup per CU (256 on an MI355X) when K/V/PK are staged in LDS:
    aim for ~n_cu workgroups of 8 waves (2 per SIMD, 256-VGPR budget: no spills)."""
    lds = 1 if N1 <= 112 else 0          # tables on chip: cooperative MFMA kernel (fused rollouts), LDS copies otherwise
    waves = 8
    if n_cu is None:
        n_cu = n_cus() if torch.cuda.is_available() else 256
    tiles = max(1, min(M, (n_cu + B - 1) // B))
    if lds:
        tiles = coop_tiles(B, M, n_cu)
    if not lds and N1 <= 128:
        tiles = max(tiles, min(M, (4 * n_cu + B - 1) // B))
    # N1 > 128: the fused rollout picks its own geometry (16 or 32 trajectories per workgroup, launch_fwd_mt)
    return waves, tiles, lds


def max_steps(kind: int, N1: int) -> int:
    # CVRP: depot + every customer + at most one depot return per customer
    return N1 if kind == L.PROBLEM_TSP else 2 * N1


def _fill_common(a: L.RolloutArgs, prob: Problem, pol: Policy, M: int, geometry=None):
    waves, tiles, lds = geometry or launch_geometry(prob.B, M, prob.N1)
    a.problem, a.B, a.M, a.N1, a.K = prob.kind, prob.B, M, prob.N1, pol.K
    a.has_local, a.has_penalty = int(pol.has_local), int(pol.has_penalty)
    a.euclidean = int(getattr(pol, 'euclidean', False))
    Ks = tuple(getattr(pol, 'Ks', ()))
    if len(Ks) > 1:
        if len(Ks) > L.MAX_ENS:
            raise ValueError(f"ensemble_size {len(Ks)}: supported 1 .. {L.MAX_ENS}")
        if pol.loc is not None and pol.loc.numel() != len(Ks) * L.LOC_SIZE:
            raise ValueError("ensemble: loc must hold one folded table per member")
        a.ens = len(Ks)
        for i, k in enumerate(Ks):
            a.Kens[i] = int(k)
    a.waves, a.tiles, a.lds_stage = waves, tiles, lds
    a.xi, a.clip, a.inv_ens = pol.xi, pol.clip, pol.inv_ens
    t = pol.tables
    for k in ("K", "V", "PK", "pb", "Q1"):
        _need_cuda(t[k], k)
        assert t[k].dtype == torch.float32 and t[k].is_contiguous()
    a.Kmat, a.Vmat, a.PK, a.pb, a.Q1 = _ptr(t["K"]), _ptr(t["V"]), _ptr(t["PK"]), _ptr(t["pb"]), _ptr(t["Q1"])
    a.Q2, a.wl = _ptr(t.get("Q2")), _ptr(t.get("wl"))
    a.xy, a.demand = _ptr(prob.xy), _ptr(prob.demand)
    a.nbr_idx, a.nbr_dist, a.nbr_theta = _ptr(prob.nbr.idx), _ptr(prob.nbr.dist), _ptr(prob.nbr.theta)
    a.loc = _ptr(pol.loc)


class TrainRows:
    """Workspace for the rows a training forward saves (time-major, Rcap = Tcap*M rows per instance).
    Cached per shape and reused every step; `gen` detects reuse before the matching backward ran."""
    _cache: Dict[tuple, "TrainRows"] = {}

    def __init__(self, B, M, N1, Tcap, dev):
        Rcap = Tcap * M
        self.B, self.M, self.N1, self.Tcap, self.Rcap = B, M, N1, Tcap, Rcap
        self._A = None                          # glimpse weights: only for forwards that cannot save mask rows
        # feasibility mask words per row: 128 bits, or the streaming kernel's node chunks (256 / 512 / 1024 bits)
        self.W = 2 if N1 <= 128 else 4 if N1 <= 256 else 8 if N1 <= 512 else 16
        self.Mask = torch.zeros(B, Rcap, self.W, device=dev, dtype=torch.int64)
        self.Lse = torch.zeros(B, Rcap, H, device=dev)                        # glimpse log2-sum-exp per (row, head)
        self.use_mask = False                   # set by the forward that filled the rows
        self.PC = torch.empty(B, Rcap, N1, device=dev)
        self.Csel = torch.empty(B, Rcap, device=dev)
        self.Q = torch.empty(B, Rcap, E, device=dev)
        self.O = torch.empty(B, Rcap, E, device=dev)
        self.Load = torch.empty(B, Rcap, device=dev)
        self.Slot = torch.empty(B, Rcap, 48, device=dev, dtype=torch.int32)
        self.F = torch.empty(B, Rcap, 3, 48, device=dev)
        # zero ONCE: afterwards every value ever written is finite, and the backward multiplies the rows of
        # undecoded steps (first moves, finished trajectories) by an exactly-zero weight, so stale rows are inert
        for t in (self.PC, self.Csel, self.Q, self.O, self.Load, self.F):
            t.zero_()
        self.Slot.fill_(-1)
        self.gen = 0

    @classmethod
    def get(cls, B, M, N1, Tcap, dev):
        key = (B, M, N1, Tcap, str(dev))
        ws = cls._cache.get(key)
        if ws is None:
            if N1 > 128:
                cls._cache.clear()              # one large-N workspace at a time (GBs each)
            ws = cls._cache[key] = TrainRows(B, M, N1, Tcap, dev)
        return ws

    @staticmethod
    def nbytes(B, M, N1, Tcap):
        """HBM a workspace of this shape takes (the (B,Rcap,N1) score rows dominate)."""
        return 4 * B * Tcap * M * (N1 + 2 * E + 3 * 48 + 48 + 2 + H + 2 * 16)

    @property
    def A(self):
        if self._A is None:
            self._A = torch.zeros(self.B, H, self.Rcap, self.N1, device=self.PC.device)
        return self._A

    def prepare(self):
        self.gen += 1


def rollout_stats_launch(res: "RolloutResult"):
    """Enqueue elg_rollout_stats: (stats, zero_steps, block) stay on the device -- stats = [longest trajectory T, 1 if some
    chosen probability is exactly 0], zero_steps[t] = 1 for the steps where that happened; block = stats followed by two
    zeroed int32 (room for the feasibility flags: one zero fill and one read-back for both).  No host sync."""
    B, M = res.tlen.shape
    Tcap = res.actions.shape[2]
    buf = torch.zeros(4 + Tcap, dtype=torch.int32, device=res.tlen.device)     # stats | room for two feasibility flags | steps
    stats, zsteps = buf[:2], buf[4:]
    with torch.cuda.device(res.tlen.device):
        L.check(L.lib().elg_rollout_stats(_ptr(res.tlen), _ptr(res.probs), B, M, Tcap, _ptr(stats), _ptr(zsteps),
                                          _stream(res.tlen.device)), "elg_rollout_stats")
    return stats, zsteps, buf[:4]


def rollout_stats(res: "RolloutResult") -> tuple:
    """(T, zero_prob): longest trajectory and whether some chosen probability is exactly 0 -- one tiny launch and THE
    host sync of a rollout (elg_rollout_stats)."""
    stats, _, _ = rollout_stats_launch(res)
    T, z = stats.tolist()
    return int(T), bool(z)


@dataclass
class RolloutResult:
    actions: torch.Tensor           # (B,M,Tcap) int32 (slice [:, :, :T])
    probs: Optional[torch.Tensor]   # (B,Tcap,M) f32; None when the caller asked for none (greedy evaluation, N1 > 128)
    reward: torch.Tensor            # (B,M) f32 = -length on the scaled coordinates
    tlen: torch.Tensor              # (B,M) int32
    full_probs: Optional[torch.Tensor] = None
    rows: Optional["TrainRows"] = None      # backward rows saved by a training forward
    kernel_id: int = 0                      # _lib.KERNEL_*: the construction kernel that ran (elg_rollout_last_kernel)


# Arithmetic of the glimpse backward's five products (elg_decoder_bwd_args.mfma_mode; include/elg_hip.h): 0 = f32 MFMAs (exact
# f32 products), 1 = split-bf16 with 2 terms per operand, 2 = split-bf16 with a 3-term score product (the default: 7.5e-6 of the
# largest gradient entry against float64 where the f32 MFMAs give 1.1e-6 -- tests/test_gpu_train_glue.py -- at 0.6 of the time)
BWD_MFMA_MODE = int(os.environ.get("ELG_BWD_MFMA_MODE", "0"))

# Arithmetic of the rollout's three table products (glimpse scores / output, pointer scores; elg_rollout_args.precision):
# "f32" = the parity mode (exact f32 products, the 1e-4 logit bar of north_star) and the default; "bf16" = the throughput mode
# BASELINE configs[1] names: bf16 operands on v_mfma_f32_16x16x32_bf16, f32 accumulation.  Three kernels honour it: the
# cooperative kernel (N + 1 <= 112, training and evaluation), the streaming kernel (128 < N + 1 <= 1024, evaluation only) and the
# N + 1 > 1024 kernel (evaluation) -- so ELG_FWD_MODE=bf16 also changes TSPLIB / VRPLIB / XXL evaluation tours; the one-wavefront
# kernels and the training forward above 128 nodes compute in f32 whatever the mode.  Tolerance of the mode, as tested
# (tests/test_gpu_logits.py::test_bf16_mode_*, tests/test_gpu_backward.py::test_bf16_mode_training_gradients): pinned on the
# oracle's bf16 restatement (the same operands rounded) at the f32 bar -- scores before the clip within 1e-4 max(|ref|, 1) on
# >= 99 % of the open nodes (<= 2e-3 on the rest), gradients within 1e-3 of the tensor's largest entry -- and within 1.4e-2 max(|ref|, 1) of the
# reference's f32 scores at CVRP-100 (per-fixture bounds in the test); greedy tours differ from the f32 ones in a few per
# cent of the steps.
FWD_PRECISION = {"f32": 0, "fp32": 0, "bf16": 1}[os.environ.get("ELG_FWD_MODE", "f32").lower()]

LARGE_ROWS_BUDGET = 0.45          # fraction of the free HBM the saved rows of a 128 < N1 <= 1024 training forward may take
_LOG_PATHS = os.environ.get("ELG_LOG_PATHS", "0") not in ("", "0")      # print which large-instance backward a step takes


def _free_hbm(dev, drop_cached_rows: bool = False) -> int:
    """HBM a new workspace can actually get: the driver's free figure PLUS what torch's caching allocator holds without using
    (reserved - allocated: the previous shape's scratch and rows sit there after `del`).  mem_get_info alone made the choice
    between the saved-rows and the replay backward depend on the allocation history -- and so possibly differ per rank.
    `drop_cached_rows`: the rows of another large shape are about to be evicted by TrainRows.get(); count them as free."""
    free, _ = torch.cuda.mem_get_info(dev)
    cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    evict = 0
    if drop_cached_rows:
        evict = sum(TrainRows.nbytes(w.B, w.M, w.N1, w.Tcap) for w in TrainRows._cache.values() if w.N1 > 128)
    return int(free + max(cached, 0) + evict)


def _rows_fit(B, M, N1, Tcap, dev) -> bool:
    """Saved rows for 128 < N1 <= 1024 are (B, Tcap M, N1) floats: used when they (and one instance of the backward's
    scratch) fit in the free HBM, else the step trains through the replay backward (which holds only the decoded rows)."""
    key = (B, M, N1, Tcap, str(dev))
    if key in TrainRows._cache:
        return True
    one_ws = 4 * int(L.lib().elg_decoder_bwd_ws_floats(1, Tcap * M, N1))
    need = TrainRows.nbytes(B, M, N1, Tcap) + one_ws
    fits = need <= LARGE_ROWS_BUDGET * _free_hbm(dev, drop_cached_rows=True)
    if _LOG_PATHS:
        print(f"[elg_amd] N1={N1} B={B} M={M}: backward over {'saved rows' if fits else 'the replay kernel'} "
              f"({need / 2**30:.1f} GiB of rows + scratch, {_free_hbm(dev) / 2**30:.1f} GiB usable)", flush=True)
    return fits


_F32_NOTED = set()


def _note_f32_fallback(what: str):
    """The bf16 mode (ELG_FWD_MODE=bf16 / engine.FWD_PRECISION = 1) covers the cooperative kernel (N + 1 <= 112, training and
    evaluation) and the streaming kernels' evaluation; where a launch runs in f32 although the mode is on, say so -- once per case."""
    if what not in _F32_NOTED:
        _F32_NOTED.add(what)
        import warnings
        warnings.warn(f"elg_amd: bf16 mode requested, but this launch runs in f32: {what}", RuntimeWarning, stacklevel=3)


def rollout_forward(prob: Problem, pol: Policy, M: int, starts: torch.Tensor, mode: int, *, forced=None, seed: int = 0,
                    uniforms=None, dump_T: int = 0, geometry=None, Tcap: Optional[int] = None,
                    train: bool = False, variant: int = 0, dump: str = "probs", precision: Optional[int] = None,
                    need_probs: bool = True) -> RolloutResult:
    """Run every trajectory to completion in one persistent launch (reference CVRP/utils.py:7-29).  need_probs = False (greedy
    only: the reference's greedy rollout returns no probabilities, utils.py:24-25) leaves RolloutResult.probs None; the kernels for
    N1 > 128 then skip the softmax normaliser of every step (the cooperative kernel's production instantiation writes them anyway)."""
    dev = prob.xy.device
    _need_cuda(prob.xy, "the problem")
    if train and _PENDING_SIDE:
        _drain_side(dev)                 # a side-stream row backward nobody waited for still reads the saved rows
    B, N1 = prob.B, prob.N1
    Tcap = Tcap or max_steps(prob.kind, N1)
    actions = torch.zeros(B, M, Tcap, device=dev, dtype=torch.int32)       # finished -> depot (0)
    if not need_probs and (mode != L.MODE_GREEDY or train or dump_T > 0):
        raise ValueError("need_probs=False is for greedy inference rollouts")
    need_probs = need_probs or N1 <= 128
    probs = torch.ones(B, Tcap, M, device=dev, dtype=torch.float32) if need_probs else None      # finished -> probability 1
    reward = torch.empty(B, M, device=dev, dtype=torch.float32)
    tlen = torch.empty(B, M, device=dev, dtype=torch.int32)
    full = torch.zeros(B, M, dump_T, N1, device=dev, dtype=torch.float32) if dump_T > 0 else None
    a = L.RolloutArgs()
    _fill_common(a, prob, pol, M, geometry)
    a.Tmax, a.mode, a.max_steps, a.do_decode, a.do_update, a.use_state = Tcap, mode, 0, 1, 1, 0
    a.seed = seed & 0xFFFFFFFFFFFFFFFF
    a.variant = variant
    a.precision = FWD_PRECISION if precision is None else int(precision)
    if a.precision == 1 and train and N1 > 128:
        a.precision = 0                  # the saved-rows backward of the streaming kernel differentiates the f32 forward
        _note_f32_fallback(f"training forward at N + 1 = {N1} > 128 nodes")
    elif a.precision == 1 and 112 < N1 <= 128:
        _note_f32_fallback(f"N + 1 = {N1} in 113 .. 128 (the one-wavefront kernel has no bf16 instantiation)")
    a.dump_logits = {"probs": 0, "logits": 1, "scores": 2}[dump]
    # (a pageable host tensor would block the host until the stream has drained: staged through pinned memory instead)
    starts = h2d(starts.to(torch.int32).contiguous(), dev) if not starts.is_cuda else starts.to(torch.int32).contiguous()
    a.starts = _ptr(starts)
    if forced is not None:
        forced = forced.to(device=dev, dtype=torch.int32).contiguous()
        a.forced, a.Tforced = _ptr(forced), forced.shape[2]
    if uniforms is not None:
        uniforms = uniforms.to(device=dev, dtype=torch.float32).contiguous()
        assert uniforms.shape == (B, M, Tcap)
        a.uniforms = _ptr(uniforms)
    a.actions, a.probs, a.reward, a.tlen = _ptr(actions), _ptr(probs), _ptr(reward), _ptr(tlen)
    a.full_probs, a.dump_T = _ptr(full), dump_T
    scratch = None
    n_scratch = int(L.lib().elg_rollout_scratch_floats(B, M, N1, variant))
    if n_scratch:
        scratch = torch.empty(n_scratch, device=dev)           # score rows (N1 > 1024) / fragment-major tables (N1 > 128)
        a.scratch = _ptr(scratch)
    rows = None
    # an ensemble trains through the replay backward; so do shapes whose rows would not fit next to the backward's scratch
    save_rows = (train and pol.ens == 1 and not pol.wide_slots
                 and (N1 <= 128 or (N1 <= 1024 and variant == 0 and _rows_fit(B, M, N1, Tcap, dev))))
    if save_rows:
        rows = TrainRows.get(B, M, N1, Tcap, dev)
        rows.prepare()
        # (only the cooperative kernel has a bf16 mode: every other kernel computes -- and is differentiated -- in f32)
        rows.precision = int(a.precision) if (a.lds_stage and 4 <= N1 <= 112 and variant in (0, 4, 5)) else 0
        # the cooperative kernel (what dispatch_fwd picks for this launch shape) saves the rows' 128-bit mask words and the
        # glimpse log2-sum-exp per head instead of the glimpse weights: the backward recomputes the weights from q, K, the
        # mask and the saved normaliser (28 MFMAs + one exp2 per weight).  4.2 GB less workspace and 6.6 GB less HBM traffic
        # per step at the bench shape, and no slower (the forward's 3.3 GB of scattered stores cost what the recompute does)
        # The streaming kernel (128 < N1 <= 1024) saves the same rows with W = 4 / 8 / 16 mask words.
        rows.use_mask = bool((a.lds_stage and a.waves == 8 and 4 <= N1 <= 112 and variant in (0, 4, 5)) or N1 > 128)
        a.trA = None if rows.use_mask else _ptr(rows.A)
        a.trMask = _ptr(rows.Mask) if rows.use_mask else None
        a.trLse = _ptr(rows.Lse) if rows.use_mask else None
        a.trPC, a.trCsel, a.trQ, a.trO = _ptr(rows.PC), _ptr(rows.Csel), _ptr(rows.Q), _ptr(rows.O)
        a.trLoad, a.trSlot, a.trF = _ptr(rows.Load), _ptr(rows.Slot), _ptr(rows.F)
    L.check(L.lib().elg_rollout_fwd(C.byref(a), _stream()), "elg_rollout_fwd")
    kernel_id = int(L.lib().elg_rollout_last_kernel())
    res = RolloutResult(actions, probs, reward, tlen, full, kernel_id=kernel_id)
    if rows is not None:
        res.rows = rows
        res.rows_gen = rows.gen
    return res


# ----------------------------------------------------------------------------------------------
# backward: saved rows (or the replay kernel) -> elg_decoder_bwd / elg_local_bwd_rows: hand-written kernels, no library GEMM
# ----------------------------------------------------------------------------------------------
class _ChosenProbs(torch.autograd.Function):
    """probs[b,t,m] of the recorded actions as a differentiable function of the folded tables."""

    @staticmethod
    def forward(ctx, prob: Problem, pol_meta: Policy, M, actions, probs_val, T, geometry,
                Kt, Vt, PKt, pbt, Q1t, Q2t, wlt, loct, rows=None, rows_gen=-1, tlen=None, T_dev=None, loc_from_fold=False):
        ctx.prob, ctx.pol_meta, ctx.M, ctx.T, ctx.geometry = prob, pol_meta, M, T, geometry
        ctx.rows, ctx.rows_gen = rows, rows_gen
        ctx.T_dev = T_dev                       # device-resident step count (the host only knows the bound T)
        ctx.probs_val = probs_val
        ctx.tlen = tlen
        ctx.save_for_backward(actions, Kt, Vt, PKt, pbt, Q1t, Q2t if Q2t is not None else Kt.new_empty(0),
                              wlt if wlt is not None else Kt.new_empty(0), loct if loct is not None else Kt.new_empty(0))
        ctx.has = (Q2t is not None, wlt is not None, loct is not None)
        # the consumer of d loc is known to wait for a side-stream launch only when loc came out of _FoldLocal
        ctx.loc_from_fold = bool(loc_from_fold)
        # a NEW tensor over the rollout's own probability buffer (nobody writes that buffer again: every rollout allocates its own);
        # a clone only where the slice is not contiguous (utils.rollout with T below the capacity)
        out = probs_val.detach() if probs_val.is_contiguous() else probs_val.contiguous()
        ctx.probs_out = out.detach()            # (never `out` itself: output -> grad_fn -> ctx -> output is a cycle only the GC frees)
        return out

    @staticmethod
    def backward(ctx, gprob):
        actions, Kt, Vt, PKt, pbt, Q1t, Q2t, wlt, loct = ctx.saved_tensors
        hasQ2, haswl, hasloc = ctx.has
        prob, meta, M, T = ctx.prob, ctx.pol_meta, ctx.M, ctx.T
        dev = Kt.device
        B, N1 = prob.B, prob.N1
        R = M * T
        g = gprob[:, :T, :].contiguous().float()
        rows = ctx.rows
        use_saved = rows is not None and rows.gen == ctx.rows_gen and T <= rows.Tcap
        if use_saved:
            return _ChosenProbs._backward_saved_rows(ctx, g, rows)
        if ctx.T_dev is not None:
            raise RuntimeError("a rollout with a deferred host sync needs the rows its training forward saved "
                               "(another training forward has reused them)")
        tables = dict(K=Kt, V=Vt, PK=PKt, pb=pbt, Q1=Q1t, Q2=Q2t if hasQ2 else None, wl=wlt if haswl else None)
        pol = Policy(tables, loct if hasloc else None, meta.K, meta.xi, meta.clip, meta.inv_ens, meta.has_local,
                     meta.has_penalty, getattr(meta, 'euclidean', False), tuple(getattr(meta, 'Ks', ())))
        forced = actions[:, :, :T].contiguous()
        fl = forced.long()
        # ---- replay path (rollouts that were not run as a training forward, N1 > 128): the recorded actions are replayed
        # inside rollout_bwd_kernel (rows r = m*T + t), dense contractions on the rows it emits
        ba = L.BwdArgs()
        _fill_common(ba.fwd, prob, pol, M, ctx.geometry)
        ba.fwd.Tmax, ba.fwd.mode, ba.fwd.max_steps, ba.fwd.do_decode, ba.fwd.do_update = T, L.MODE_FORCED, 0, 1, 1
        ba.fwd.forced, ba.fwd.Tforced = _ptr(forced), T
        ba.T = T
        gloc = torch.zeros(pol.ens * L.LOC_SIZE, device=dev)
        ba.gloc = _ptr(gloc)
        rowA = torch.empty(B, H, R, N1, device=dev)
        rowDL = torch.empty(B, R, N1, device=dev)
        rowQ = torch.empty(B, R, E, device=dev)
        rowO = torch.empty(B, R, E, device=dev)
        rowLoad = torch.empty(B, R, device=dev) if haswl else None
        rowDU = torch.empty(B, R, 64 if pol.wide_slots else 48, device=dev) if meta.has_local else None    # csrc slot_stride_of()
        ba.gprob = _ptr(g)
        ba.rowA, ba.rowDL, ba.rowQ, ba.rowO = _ptr(rowA), _ptr(rowDL), _ptr(rowQ), _ptr(rowO)
        ba.rowLoad, ba.rowDU = _ptr(rowLoad), _ptr(rowDU)
        L.check(L.lib().elg_rollout_bwd(C.byref(ba), _stream()), "elg_rollout_bwd")
        # the query of decode step t was gathered at cur = action[t-1] (and first = action[0] for TSP)
        prev = torch.cat([torch.zeros(B, M, 1, dtype=torch.long, device=dev), fl[:, :, :-1]], dim=2).reshape(B, R)
        first = fl[:, :, :1].expand(B, M, T).reshape(B, R) if hasQ2 else None
        lib = L.lib()

        def bgemm(A_, B_, C_, Mm, Nn, Kk, lda, ldb, ldc, tA, tB, n_in, sA, sB, sC, what):
            """C(b, i) = op(A(b, i)) op(B(b, i)) for b < B, i < n_in through elg_gemm_f32_batched (f32 MFMA); s* = (outer, inner)."""
            L.check(lib.elg_gemm_f32_batched(_ptr(A_), _ptr(B_), _ptr(C_), Mm, Nn, Kk, lda, ldb, ldc, int(tA), int(tB), B, n_in,
                                             sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], 1.0, _stream()), what)
        # every row contraction of the pointer / glimpse backward is a batched f32 MFMA GEMM of csrc/elg_gemm.hip
        dO = torch.empty(B, R, E, device=dev)                       # d o = d s . PK                (B: R x N1 @ N1 x 128)
        bgemm(rowDL, PKt, dO, R, E, N1, N1, E, E, 0, 0, 1, (R * N1, 0), (N1 * E, 0), (R * E, 0), "dO = dS PK")
        dQ = torch.empty(B, R, E, device=dev)
        if N1 <= 128:
            splits = max(1, min(8, 1024 // (B * H)))
            dKp = torch.empty(splits, B, N1, E, device=dev)
            dVp = torch.empty(splits, B, N1, E, device=dev)
            L.check(lib.elg_glimpse_bwd_fused(_ptr(rowA), None, _ptr(dO), _ptr(rowO), _ptr(rowQ), _ptr(Kt), _ptr(Vt),
                                              _ptr(dQ), _ptr(dKp), _ptr(dVp), B, R, N1, R, R, R, splits, _stream()),
                    "elg_glimpse_bwd_fused")
            dK = dKp[0] if splits == 1 else dKp.sum(0)
            dV = dVp[0] if splits == 1 else dVp.sum(0)
        else:
            dS = torch.empty(B, H, R, N1, device=dev)               # d(q.K)
            if N1 <= 256:
                L.check(lib.elg_glimpse_rows_bwd(_ptr(rowA), _ptr(dO), _ptr(rowO), _ptr(Kt), _ptr(Vt), _ptr(dS),
                                                 _ptr(dQ), B, R, N1, R, R, _stream()), "elg_glimpse_rows_bwd")
            else:
                # dA_h = dO_h V_h^T (per head: R x 16 @ 16 x N1), softmax backward in place, dQ_h = dS_h K_h
                bgemm(dO, Vt, dS, R, N1, DK, E, E, N1, 0, 1, H, (R * E, DK), (N1 * E, DK), (H * R * N1, R * N1), "dA = dO V^T")
                doto = (dO * rowO).view(B, R, H, DK).sum(-1).permute(0, 2, 1)                 # <dO_h, O_h>   (B,H,R)
                dS.sub_(doto[..., None]).mul_(rowA).mul_(0.25)
                bgemm(dS, Kt, dQ, R, DK, N1, N1, E, E, 0, 0, H, (H * R * N1, R * N1), (N1 * E, DK), (R * E, DK), "dQ = dS K")
            dK = torch.empty(B, N1, E, device=dev)                  # dK_h = dS_h^T Q_h , dV_h = a_h^T dO_h   (N1 x R @ R x 16)
            dV = torch.empty(B, N1, E, device=dev)
            bgemm(dS, rowQ, dK, N1, DK, R, N1, E, E, 1, 0, H, (H * R * N1, R * N1), (R * E, DK), (N1 * E, DK), "dK = dS^T Q")
            bgemm(rowA, dO, dV, N1, DK, R, N1, E, E, 1, 0, H, (H * R * N1, R * N1), (R * E, DK), (N1 * E, DK), "dV = a^T dO")
        dPK = torch.empty(B, N1, E, device=dev)                     # dPK = dS_ptr^T O              (B: N1 x R @ R x 128)
        bgemm(rowDL, rowO, dPK, N1, E, R, N1, E, E, 1, 0, 1, (R * N1, 0), (R * E, 0), (N1 * E, 0), "dPK = dS^T O")
        dpb = rowDL.sum(dim=1)
        dQ2 = dwl = None
        # gather backward of the query rows: scatter-add of the row cotangents onto the nodes they were gathered from
        base = (torch.arange(B, device=dev) * N1)[:, None]
        dQ1 = torch.zeros(B * N1, E, device=dev).index_add_(0, (prev + base).reshape(-1), dQ.view(B * R, E)).view(B, N1, E)
        if hasQ2:
            dQ2 = torch.zeros(B * N1, E, device=dev).index_add_(0, (first + base).reshape(-1), dQ.view(B * R, E)).view(B, N1, E)
        if haswl:
            dwl = (rowLoad[:, :, None] * dQ).sum(dim=(0, 1))
        return (None, None, None, None, None, None, None,
                dK, dV, dPK, dpb, dQ1, dQ2, dwl, gloc if hasloc else None, None, None, None, None, None)

    @staticmethod
    def _backward_saved_rows(ctx, g, rows):
        """Training path: elg_decoder_bwd over the rows the forward saved (+ elg_local_bwd_rows) -- two calls into
        libelg_hip.so, no framework kernels besides the zero fill of the gradient buffer."""
        actions, Kt, Vt, PKt, pbt, Q1t, Q2t, wlt, loct = ctx.saved_tensors
        hasQ2, haswl, hasloc = ctx.has
        prob, meta, M, T = ctx.prob, ctx.pol_meta, ctx.M, ctx.T
        dev = Kt.device
        B, N1 = prob.B, prob.N1
        R = M * T
        tsp = prob.kind == L.PROBLEM_TSP
        # one zero-filled buffer for every accumulated output: dK dV dPK dQ1 [dQ2] | dpb | dwl | gloc
        nt = 5 if hasQ2 else 4
        flat = torch.zeros(nt * B * N1 * E + B * N1 + E + L.LOC_SIZE, device=dev)
        blk = B * N1 * E
        dK, dV, dPK, dQ1 = (flat[i * blk:(i + 1) * blk].view(B, N1, E) for i in range(4))
        dQ2 = flat[4 * blk:5 * blk].view(B, N1, E) if hasQ2 else None
        o = nt * blk
        dpb = flat[o:o + B * N1].view(B, N1)
        dwl = flat[o + B * N1:o + B * N1 + E]
        gloc = flat[o + B * N1 + E:]
        ws = _BwdScratch.get(B, rows.Rcap, dev, meta.has_local, hasQ2)        # sized for Rcap, used densely over R rows
        a = L.DecoderBwdArgs()
        a.problem, a.B, a.M, a.N1, a.T, a.Tcap_actions = prob.kind, B, M, N1, T, actions.shape[2]
        a.first_decode_step, a.inv_ens, a.Rcap = (1 if tsp else 2), float(meta.inv_ens), rows.Rcap
        a.gprob, a.pval, a.tlen, a.actions = _ptr(g), _ptr(ctx.probs_out), _ptr(ctx.tlen), _ptr(actions)
        a.trPC, a.trCsel, a.trQ, a.trO = _ptr(rows.PC), _ptr(rows.Csel), _ptr(rows.Q), _ptr(rows.O)
        a.trLoad = _ptr(rows.Load) if haswl else None
        a.trSlot = _ptr(rows.Slot) if meta.has_local else None
        a.trA = None if rows.use_mask else _ptr(rows.A)
        a.trMask = _ptr(rows.Mask) if rows.use_mask else None
        a.trLse = _ptr(rows.Lse) if rows.use_mask else None
        a.Kmat, a.Vmat, a.PK = _ptr(Kt), _ptr(Vt), _ptr(PKt)
        a.dK, a.dV, a.dPK, a.dpb, a.dQ1, a.dQ2 = _ptr(dK), _ptr(dV), _ptr(dPK), _ptr(dpb), _ptr(dQ1), _ptr(dQ2)
        a.dwl = _ptr(dwl) if haswl else None
        a.rowDU = _ptr(ws.rowDU) if meta.has_local else None
        a.dO, a.idx_prev, a.idx_first, a.rowW = _ptr(ws.dO), _ptr(ws.idx_prev), _ptr(ws.idx_first), _ptr(ws.rowW)
        a.T_dev, a.gprob_T = _ptr(ctx.T_dev), g.shape[1]
        # `training: only_local`: the decoder tables are constants (zeros) -- the glimpse backward would compute gradients nobody reads
        a.tables_frozen = int(not any(ctx.needs_input_grad[7:14]))
        # a bf16 forward saved the normaliser of ITS scores: the backward recomputes them the same way (mode 3)
        a.mfma_mode = 3 if getattr(rows, "precision", 0) == 1 else BWD_MFMA_MODE
        big = None
        if N1 > 128:
            # row contractions as batched GEMMs over (8, R, N1) buffers: scratch for as many instances as fit, the call walks
            # the batch in chunks of that many
            per = int(L.lib().elg_decoder_bwd_ws_floats(1, R, N1))
            nb = max(1, min(B, int(0.6 * _free_hbm(dev)) // (4 * per)))      # never more than the batch needs
            big = torch.empty(nb * per, device=dev)
            a.ws, a.ws_floats, a.mask_words = _ptr(big), big.numel(), rows.W
        L.check(L.lib().elg_decoder_bwd(C.byref(a), _stream()), "elg_decoder_bwd")
        del big
        if meta.has_local:
            # rows are independent given the saved slot features: 16 rows per wavefront on the matrix cores
            n_slots = meta.K + (0 if tsp else 1)

            def launch(max_wg=0):
                L.check(L.lib().elg_local_bwd_rows(_ptr(loct), _ptr(rows.F), _ptr(rows.Slot), _ptr(ws.rowDU), _ptr(gloc),
                                                   B, R, rows.Rcap, n_slots, _ptr(ctx.T_dev), M, max_wg, _stream()),
                        "elg_local_bwd_rows")
            # (worth it when the row kernel is long enough to matter: >= 64 16-row tiles per CU it would leave idle)
            if SIDE_LOCAL_BWD and ctx.loc_from_fold and N1 <= 128 and B * R >= 16 * SIDE_LOCAL_MIN_TILES_PER_CU * n_cus(dev):
                main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                side.wait_stream(main)               # rowDU (pointer backward) and the zero fill of gloc are on the main stream
                with torch.cuda.stream(side):
                    launch(SIDE_LOCAL_BWD_GRID or n_cus(dev) // 2)     # half the chip (measured: 5.61 -> 5.43 ms per step)
                    ev = torch.cuda.Event()
                    ev.record(side)
                idx = torch.device(dev).index
                _PENDING_SIDE.append((torch.cuda.current_device() if idx is None else idx, ev))
            else:
                launch()
        return (None, None, None, None, None, None, None,
                dK, dV, dPK, dpb, dQ1, dQ2, dwl if haswl else None, gloc if hasloc else None, None, None, None, None, None)


class _BwdScratch:
    """Scratch of the saved-rows backward, cached per shape (dO, the gather indices, the local-policy cotangents)."""
    _cache: Dict[tuple, "_BwdScratch"] = {}

    def __init__(self, B, R, dev, local, first):
        self.dO = torch.empty(B, R, E, device=dev)
        self.idx_prev = torch.empty(B, R, device=dev, dtype=torch.int32)
        self.rowW = torch.empty(B, R, 4, device=dev)
        self.idx_first = torch.empty(B, R, device=dev, dtype=torch.int32) if first else None
        self.rowDU = torch.empty(B, R, 48, device=dev) if local else None

    @classmethod
    def get(cls, B, R, dev, local, first):
        key = (B, R, str(dev), bool(local), bool(first))
        ws = cls._cache.get(key)
        if ws is None:
            if len(cls._cache) > 8:
                cls._cache.clear()
            ws = cls._cache[key] = _BwdScratch(B, R, dev, local, first)
        return ws


def chosen_probs(prob: Problem, pol: Policy, M: int, res: RolloutResult, T: int, geometry=None, T_dev=None) -> torch.Tensor:
    """Differentiable view of res.probs[:, :T, :] (gradients flow to pol.tables / pol.loc).  With `T_dev` (the device
    tensor holding the rollout's step count, rollout_stats_launch) T is only an upper bound -- normally the capacity of
    res.probs -- and the backward kernels take the count from the device: no host sync between rollout and backward."""
    t = pol.tables
    rows = getattr(res, "rows", None)
    if T_dev is not None and rows is None:
        raise ValueError("chosen_probs: T_dev needs a training forward (saved rows)")
    return _ChosenProbs.apply(prob, pol, M, res.actions, res.probs[:, :T, :], T, geometry,
                              t["K"], t["V"], t["PK"], t["pb"], t["Q1"], t.get("Q2"), t.get("wl"), pol.loc,
                              rows, getattr(res, "rows_gen", -1), res.tlen, T_dev,
                              pol.loc is not None and type(pol.loc.grad_fn).__name__ == "_FoldLocalBackward")


# ----------------------------------------------------------------------------------------------
# REINFORCE / POMO loss (one launch forward, one elementwise op backward)
# ----------------------------------------------------------------------------------------------
class _PomoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, rewards, scale_norm, guard_zero):
        B, T, M = probs.shape
        if probs.stride(2) != 1:
            probs = probs.contiguous()
        rewards = rewards.contiguous().float()
        dev = probs.device
        scal = torch.empty(3, B, device=dev)                       # J_raw, J_scaled, adv_max
        coef = torch.empty(2, B, M, device=dev)                    # raw, scaled
        L.check(L.lib().elg_pomo_loss(_ptr(probs), _ptr(rewards), B, T, M, probs.stride(0), probs.stride(1),
                                      _ptr(scal[0]), _ptr(scal[1]), _ptr(scal[2]), _ptr(coef[0]), _ptr(coef[1]),
                                      _stream()), "elg_pomo_loss")
        if not scale_norm:
            J, c = scal[0].sum(), coef[0]
        elif guard_zero:                                           # TSP/train.py:113-116, decided on the device
            zero = (scal[2] == 0).any()
            J = torch.where(zero, scal[0].sum(), scal[1].sum())
            c = torch.where(zero, coef[0], coef[1])
        else:
            J, c = scal[1].sum(), coef[1]
        ctx.save_for_backward(probs, c)
        ctx.scale = 1.0 / (B * M)
        return J * ctx.scale

    @staticmethod
    def backward(ctx, gout):
        probs, c = ctx.saved_tensors
        return (gout * ctx.scale) * c[:, None, :] / probs, None, None, None


class _PomoLossGrad(torch.autograd.Function):
    """The scaled loss and its gradient in one launch (elg_pomo_loss_grad): the training step's path (CVRP, scale_norm, no
    batch-wide guard).  zero_steps (Tcap int32 device flags, or None): the +1e-6 of CVRPModel.py:67-68 on the steps where a
    chosen probability was exactly 0, applied inside the kernel instead of by an element-wise add in front of it."""
    _tickets: Dict[str, torch.Tensor] = {}

    @staticmethod
    def forward(ctx, probs, rewards, zero_steps, T_dev):
        B, T, M = probs.shape
        if probs.stride(2) != 1:
            probs = probs.contiguous()
        rewards = rewards.contiguous().float()
        dev = probs.device
        key = str(dev)
        if key not in _PomoLossGrad._tickets:
            _PomoLossGrad._tickets[key] = torch.zeros(1, dtype=torch.int32, device=dev)     # left zero by every launch
        out = torch.empty(B + 1, device=dev)                   # J_terms | J
        g = torch.empty(B, T, M, device=dev)
        L.check(L.lib().elg_pomo_loss_grad(_ptr(probs), _ptr(rewards), _ptr(zero_steps), B, T, M, probs.stride(0), probs.stride(1),
                                           1.0 / (B * M), _ptr(out), _ptr(g), _ptr(T_dev), _ptr(out[B:]),
                                           _ptr(_PomoLossGrad._tickets[key]), _stream()), "elg_pomo_loss_grad")
        ctx.save_for_backward(g)
        return out[B]

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        if gout.data_ptr() == unit_grad(g.device).data_ptr():      # J.backward(unit_grad(dev)): the cotangent IS 1, nothing to scale
            return g, None, None, None
        return gout * g, None, None, None


_UNIT: Dict[str, torch.Tensor] = {}


def unit_grad(dev) -> torch.Tensor:
    """The cached device scalar 1.0.  `J.backward(unit_grad(dev))` is `J.backward()` without the fill that creates the implicit
    cotangent and -- for the fused POMO loss, which recognises this tensor -- without the element-wise product with it."""
    key = str(torch.device(dev))
    if key not in _UNIT:
        _UNIT[key] = torch.ones((), device=dev)
    return _UNIT[key]


def pomo_loss(probs: torch.Tensor, rewards: torch.Tensor, scale_norm: bool = True, guard_zero: bool = False,
              zero_steps: Optional[torch.Tensor] = None, T_dev: Optional[torch.Tensor] = None):
    """mean over (instance, trajectory) of -advantage * sum_t log p, advantage = reward - POMO mean
    (reference CVRP/train.py:112-121; guard_zero = the TSP variant's batch-wide zero-normaliser check).
    zero_steps: rollout_train's per-step flags "some chosen probability was exactly 0" -- p + 1e-6 on those steps."""
    _need_cuda(probs, "probs")
    if scale_norm and not guard_zero and probs.dim() == 3:
        # T_dev (device int32, rollout_train's step count): the padded steps behind it hold probability 1 and are not read
        return _PomoLossGrad.apply(probs.float(), rewards, None if zero_steps is None else zero_steps.contiguous(), T_dev)
    if zero_steps is not None:
        probs = torch.add(probs, zero_steps[None, :, None], alpha=1e-6)      # exact + 0.0 unless a chosen probability was 0
    return _PomoLoss.apply(probs.float(), rewards, bool(scale_norm), bool(guard_zero))


# ----------------------------------------------------------------------------------------------
# residual add + instance norm (reference models.py:506-527) as a stand-alone op: the N1 > 128 path of elg_encoder_fwd /
# the norm backward of elg_encoder_bwd use these two kernels; this wrapper exists for their unit test
# ----------------------------------------------------------------------------------------------
class _AddInstNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps):
        a, b = a.contiguous(), b.contiguous()
        B, N, Cn = a.shape
        out = torch.empty_like(a)
        xhat = torch.empty_like(a)
        rstd = torch.empty(B, Cn, device=a.device)
        L.check(L.lib().elg_add_instnorm_fwd(_ptr(a), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(xhat), _ptr(rstd),
                                             B, N, Cn, float(eps), _stream()), "elg_add_instnorm_fwd")
        ctx.save_for_backward(xhat, rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        xhat, rstd, gamma = ctx.saved_tensors
        dout = dout.contiguous()
        B, N, Cn = xhat.shape
        ds = torch.empty_like(xhat)
        dgb = torch.zeros(2, Cn, device=xhat.device)
        L.check(L.lib().elg_add_instnorm_bwd(_ptr(dout), _ptr(xhat), _ptr(rstd), _ptr(gamma), _ptr(ds), _ptr(dgb[0]),
                                             _ptr(dgb[1]), B, N, Cn, _stream()), "elg_add_instnorm_bwd")
        return ds, ds, dgb[0], dgb[1], None


def add_instance_norm(a: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """InstanceNorm1d(affine) of (a + b) over the node axis, (B,N,C) in and out, one HIP launch each way."""
    _need_cuda(a, "a")
    return _AddInstNorm.apply(a.float(), b.float(), gamma, beta, eps)


# ----------------------------------------------------------------------------------------------
# orderly shutdown: drop the cached device / pinned buffers, events and the side stream while the HIP runtime is
# still alive (module globals are otherwise destroyed in arbitrary order at interpreter exit)
# ----------------------------------------------------------------------------------------------
def _teardown():
    # no device synchronisation here: freeing a buffer orders itself behind the work that uses it, and an exit path must
    # never be able to block on the GPU (another process may hold it)
    for cache in (_PINNED, TrainRows._cache, _BwdScratch._cache, _SIDE, _PENDING_SIDE):
        cache.clear()


import atexit  # noqa: E402

atexit.register(_teardown)
