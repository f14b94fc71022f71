"""CVRPEnv with the reference's interface (gaocrr/ELG CVRP/CVRPEnv.py), state held on the MI355X.

Two ways to drive it:
  * fused  -- `utils.rollout(model, env, ...)` runs reset + every step + reward in ONE persistent HIP
              launch (csrc/elg_fwd.hip); this is the training / evaluation path.
  * stepwise -- `reset / pre_step / step / get_cur_feature` keep the reference's per-step protocol;
              each call is one small launch of the same kernel (use_state mode).  Inference only.
Either way the transition itself (load, visited set, feasibility mask, finished flag, tour length)
is computed by env_update() in csrc/elg_rollout.h -- never by PyTorch."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from elg_amd import _lib as L
from elg_amd import engine as eng


class Reset_State:
    """reference CVRPEnv.py:8-13.  `dist` (batch, problem+1, problem+1) is built on first read: the engine works from the
    neighbour tables, and a training step that never looks at it should not pay a launch and 2.6 MB for it."""

    def __init__(self):
        self.depot_xy = None             # (batch, 1, 2)
        self.node_xy = None              # (batch, problem, 2)
        self.node_demand = None          # (batch, problem)
        self._xy = None                  # (batch, problem+1, 2)   depot + customers, as the encoder kernel reads them
        self._demand = None              # (batch, problem+1)      demand[:, 0] = 0
        self._env = None

    @property
    def dist(self):
        return None if self._env is None or self._env.depot_node_xy is None else self._env.dist


@dataclass
class Step_State:
    selected_count: int = None
    load: torch.Tensor = None            # (batch, multi)
    current_node: torch.Tensor = None    # (batch, multi)
    ninf_mask: torch.Tensor = None       # (batch, multi, problem+1)
    finished: torch.Tensor = None        # (batch, multi)
    _env: object = None                  # back-reference used by CVRPModel.one_step_rollout


class CVRPEnv:
    def __init__(self, multi_width, device):
        self._tours_lazy = None
        self._selected_node_list = self._current_node = None
        self._needs_state = False
        self._st_store = self._load = self._finished = self._ninf_mask = None
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("elg_amd.CVRPEnv runs on the GPU only (no CPU fallback)")
        torch.cuda.set_device(self.device)       # kernels launch on the current device / its current stream
        self.vrplib = False
        self.problem_size = None
        self.multi_width = multi_width
        self.batch_size = None
        self.depot_node_xy = None            # (batch, problem+1, 2)
        self.depot_node_demand = None        # (batch, problem+1)
        self.unscaled_depot_node_xy = None
        self.input_mask = None
        self.selected_count = None
        self.current_node = None
        self.selected_node_list = None
        self.load = None
        self.finished = None
        self.ninf_mask = None
        self.reset_state = Reset_State()
        self.reset_state._env = self
        self.step_state = Step_State()
        self.problem = None                  # engine.Problem (coordinates, demands, neighbour tables)
        self._st = None

    # `selected_node_list` (batch, multi, T) int64 / `current_node` as the reference's env holds them after a rollout.  A training
    # step (utils.rollout_train) leaves the engine's int32 tours and their length here and the int64 tensors are formed on first
    # read: nothing in a training step reads them, and the conversion is a 10 us launch per step.
    @property
    def selected_node_list(self):
        if self._tours_lazy is not None:
            tours, T = self._tours_lazy
            self._tours_lazy = None
            self._selected_node_list = tours[:, :, :T].long()
            self._current_node = self._selected_node_list[:, :, -1]
        return self._selected_node_list

    @selected_node_list.setter
    def selected_node_list(self, v):
        self._tours_lazy = None
        self._selected_node_list = v

    @property
    def current_node(self):
        if self._tours_lazy is not None:
            _ = self.selected_node_list
        return self._current_node

    @current_node.setter
    def current_node(self, v):
        self._current_node = v

    def set_tours_lazy(self, tours_i32, T):
        self._tours_lazy = (tours_i32, int(T))
        self.selected_count = int(T)

    # ------------------------------------------------------------------ problem loading
    def _finish_load(self, depot, demand_with_depot, nbr=None):
        self.depot_node_xy = self.depot_node_xy.contiguous().float()
        self.depot_node_demand = demand_with_depot.contiguous().float()
        self.reset_state.depot_xy = depot
        self.reset_state.node_xy = self.depot_node_xy[:, 1:, :]
        self.reset_state.node_demand = self.depot_node_demand[:, 1:]
        self.reset_state._xy, self.reset_state._demand = self.depot_node_xy, self.depot_node_demand
        self.problem_size = self.depot_node_xy.shape[1] - 1
        self.problem = eng.Problem(L.PROBLEM_CVRP, self.depot_node_xy, self.depot_node_demand,
                                   nbr if nbr is not None else eng.nbr_tables(self.depot_node_xy))
        self._dist = None

    @property
    def dist(self):
        """(batch, problem+1, problem+1), built on first use: the engine itself works from the
        neighbour tables (reference CVRPEnv.py:148)."""
        if self._dist is None:
            self._dist = eng.dist_matrix(self.depot_node_xy)
        return self._dist

    def load_random_problems(self, batch, aug_factor=1):
        """reference CVRPEnv.py:125-150."""
        if aug_factor == 1 and not any(batch[k].is_cuda for k in ('loc', 'demand', 'depot')):
            # a training batch from the host generator: depot | customers and 0 | demands are joined on the HOST (26 KB) and reach the
            # device as two copies -- instead of three copies, two concatenations and a zero fill queued in front of the encoder
            depot_h = batch['depot'].float()
            if depot_h.dim() == 2:
                depot_h = depot_h[:, None, :]
            node_h, dem_h = batch['loc'].float(), batch['demand'].float()
            self.vrplib = False
            self.batch_size = node_h.shape[0]
            # ... into ONE pinned buffer by numpy slices (no torch CPU kernel: engine._host_copy says why) and to the device in ONE copy
            B_, n1 = self.batch_size, node_h.shape[1] + 1
            xy_h = torch.empty(B_, n1, 2)
            dem_hh = torch.empty(B_, n1)
            xy_n, dem_n = xy_h.numpy(), dem_hh.numpy()
            xy_n[:, :1] = depot_h.numpy()
            xy_n[:, 1:] = node_h.numpy()
            dem_n[:, 0] = 0.0
            dem_n[:, 1:] = dem_h.numpy()

            def upload():                  # host data only: on the preparation stream, next to the previous step's backward
                flat, (xy, dem) = eng.h2d_parts((xy_h, dem_hh), self.device)
                nbr = eng.nbr_tables(xy)
                # (xy / dem are h2d_parts' own views of `flat`: its layout is not recomputed here; `flat` comes first so that
                # on_prep_stream's record_stream covers the storage the views share)
                return flat, nbr.idx, nbr.dist, nbr.theta, xy, dem
            flat, idx, dist, theta, xy, dem = eng.on_prep_stream(self.device, upload)
            self.depot_node_xy = xy
            self._finish_load(xy[:, :1, :], dem, eng.NbrTables(idx, dist, theta))
            return
        node = eng.h2d(batch['loc'].float(), self.device)
        demand = eng.h2d(batch['demand'].float(), self.device)
        depot = eng.h2d(batch['depot'].float(), self.device)
        if depot.dim() == 2:
            depot = depot[:, None, :]
        self.vrplib = False
        self.batch_size = node.shape[0]
        if aug_factor > 1:
            if aug_factor != 8:
                raise NotImplementedError
            self.batch_size *= 8
            depot = eng.aug8(depot)
            node = eng.aug8(node)
            demand = demand.repeat(8, 1)
        self.depot_node_xy = torch.cat((depot, node), dim=1)
        dem = torch.cat((torch.zeros(self.batch_size, 1, device=self.device), demand), dim=1)
        self._finish_load(depot, dem)

    def load_vrplib_problem(self, instance, aug_factor=1):
        """reference CVRPEnv.py:84-123: per-axis min-max scaling to [0,1], optional 8-fold augmentation of the
        scaled and of the raw coordinates, demand / capacity."""
        self.vrplib = True
        self.batch_size = 1
        # once-per-instance preprocessing in host fp32 (IEEE division), exactly the reference's arithmetic
        coord = torch.as_tensor(instance['node_coord'], dtype=torch.float32)[None]
        demand = torch.as_tensor(instance['demand'], dtype=torch.float32)[None] / instance['capacity']
        lo = coord.min(dim=1, keepdim=True)[0]
        hi = coord.max(dim=1, keepdim=True)[0]
        scaled = ((coord - lo) / (hi - lo)).to(self.device)
        unscaled = coord.to(self.device)
        demand = demand.to(self.device)
        depot_idx = torch.as_tensor(instance['depot']).reshape(-1).long().to(self.device)
        depot = scaled[:, depot_idx, :]
        if aug_factor > 1:
            if aug_factor != 8:
                raise NotImplementedError
            self.batch_size = 8
            depot = eng.aug8(depot)
            scaled = eng.aug8(scaled)
            unscaled = eng.aug8(unscaled)
            demand = demand.repeat(8, 1)
        self.depot_node_xy = scaled
        self.unscaled_depot_node_xy = unscaled.contiguous()
        self._finish_load(depot, demand)

    # ------------------------------------------------------------------ step-wise protocol
    def reset(self):
        B, M, N1 = self.batch_size, self.multi_width, self.problem_size + 1
        dev = self.device
        self.selected_count = 0
        self.current_node = None
        self.selected_node_list = torch.zeros(B, M, 0, dtype=torch.long, device=dev)
        # the step-wise protocol's device state (load / finished / ninf_mask / kernel state words) is built on first use:
        # the fused rollout (utils.rollout) never reads it, and ten small fills per reset are 1 % of a training step
        self._st_store = self._load = self._finished = self._ninf_mask = None
        self._needs_state = True
        return self.reset_state, None, False

    def _ensure_state(self):
        if self._needs_state:
            self._needs_state = False
            B, M, N1 = self.batch_size, self.multi_width, self.problem_size + 1
            dev = self.device
            nw = (N1 + 63) // 64
            self._st_store = dict(cur=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  cnt=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  fin=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  first=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  load=torch.ones(B, M, dtype=torch.float32, device=dev),
                                  len=torch.zeros(B, M, dtype=torch.float32, device=dev),
                                  vis=torch.zeros(B, M, nw, dtype=torch.int64, device=dev))
            self._load = self._st_store["load"]
            self._finished = torch.zeros(B, M, dtype=torch.bool, device=dev)
            self._ninf_mask = torch.zeros(B, M, N1, device=dev)

    def _lazy(name):                       # noqa: N805  (property factory)
        def get(self):
            self._ensure_state()
            return getattr(self, name)

        def put(self, v):
            setattr(self, name, v)
        return property(get, put)

    _st = _lazy("_st_store")
    load = _lazy("_load")
    finished = _lazy("_finished")
    ninf_mask = _lazy("_ninf_mask")
    del _lazy

    def reset_width(self, new_width):
        self.multi_width = new_width

    def _fill_step_state(self):
        s = self.step_state
        s.selected_count, s.load, s.current_node = self.selected_count, self.load, self.current_node
        s.ninf_mask, s.finished, s._env = self.ninf_mask, self.finished, self
        return s

    def pre_step(self):
        return self._fill_step_state(), None, False

    def _state_args(self, a: L.RolloutArgs):
        st = self._st
        a.use_state = 1
        a.st_cur, a.st_cnt, a.st_fin, a.st_first = eng._ptr(st["cur"]), eng._ptr(st["cnt"]), eng._ptr(st["fin"]), eng._ptr(st["first"])
        a.st_load, a.st_len, a.st_vis = eng._ptr(st["load"]), eng._ptr(st["len"]), eng._ptr(st["vis"])

    def _materialise_mask(self):
        """(B,M,N1) float {0,-inf} view of the device state, for callers of the reference protocol
        (CVRPEnv.py:214-232).  The kernels never read this tensor."""
        st = self._st
        N1 = self.problem_size + 1
        bits = torch.arange(64, device=self.device, dtype=torch.int64)
        visited = ((st["vis"][..., None] >> bits) & 1).bool().reshape(self.batch_size, self.multi_width, -1)[..., :N1]
        too_large = (st["load"][:, :, None] + 1e-6) < self.depot_node_demand[:, None, :]
        m = visited | too_large
        fin = st["fin"].bool()
        m[:, :, 0] = m[:, :, 0] & ~fin
        self.ninf_mask = torch.zeros(m.shape, device=self.device).masked_fill_(m, float('-inf'))
        self.finished = fin

    def step(self, selected):
        """One environment transition for externally chosen nodes (reference CVRPEnv.py:190-249)."""
        B, M = self.batch_size, self.multi_width
        forced = selected.to(self.device, torch.int32).reshape(B, M, 1).contiguous()
        a = L.RolloutArgs()
        pol = eng.Policy(_NO_TABLES(self), None, 0, 0.0, 0.0, 1.0, False, False)
        eng._fill_common(a, self.problem, pol, M, geometry=(8, min(M, 4), 0))
        a.Tmax, a.mode, a.max_steps, a.do_decode, a.do_update = 1, L.MODE_FORCED, 1, 0, 1
        a.forced, a.Tforced = eng._ptr(forced), 1
        self._state_args(a)
        L.check(L.lib().elg_rollout_fwd(C.byref(a), eng._stream()), "elg_rollout_fwd(step)")
        self.selected_count += 1
        self.current_node = selected.to(self.device).long()
        self.selected_node_list = torch.cat((self.selected_node_list, self.current_node[:, :, None]), dim=2)
        self.load = self._st["load"]
        self._materialise_mask()
        done = bool(self.finished.all())
        reward = None
        if done:
            reward = self.compute_unscaled_reward() if self.vrplib else self._get_reward()
        return self._fill_step_state(), reward, done

    # ------------------------------------------------------------------ rewards / features
    def _get_reward(self):
        return -eng.route_length(self.depot_node_xy, self.selected_node_list)

    def compute_unscaled_reward(self, solutions=None, rounding=True):
        if solutions is None:
            solutions = self.selected_node_list
        B = self.unscaled_depot_node_xy.shape[0]
        sol = solutions.to(self.device)
        if sol.shape[0] != B:
            sol = sol.expand(B, -1, -1)
        return -eng.route_length(self.unscaled_depot_node_xy, sol, rounding=rounding)

    def get_cur_feature(self):
        """(cur_dist, cur_theta, relative_xy, norm_demand) as the reference returns them
        (CVRPEnv.py:291-318).  Provided for protocol compatibility only: the engine reads the same
        quantities from its neighbour tables inside the kernel."""
        if self.current_node is None:
            return None, None, None, None
        B, M, N1 = self.batch_size, self.multi_width, self.problem_size + 1
        cur = self.current_node
        cur_dist = torch.gather(self.dist, 1, cur[:, :, None].expand(B, M, N1))
        xy = self.depot_node_xy
        cxy = torch.gather(xy, 1, cur[:, :, None].expand(B, M, 2))
        rel = xy[:, None, :, :] - cxy[:, :, None, :]
        theta = torch.atan2(rel[..., 1], rel[..., 0])
        norm_demand = self.depot_node_demand[:, None, :] / self.load[:, :, None]
        return cur_dist, theta, rel, norm_demand


def _NO_TABLES(env):
    """Placeholder table pointers for env-only launches (the kernel never dereferences them when
    do_decode == 0, but the ABI wants valid device pointers)."""
    z = env.depot_node_xy
    return dict(K=z, V=z, PK=z, pb=z, Q1=z, Q2=None, wl=z)
