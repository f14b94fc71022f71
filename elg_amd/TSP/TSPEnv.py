"""TSPEnv with the reference's interface (gaocrr/ELG TSP/TSPEnv.py) on the MI355X engine.  The fused path
is `utils.rollout`; the step-wise protocol (reset / pre_step / step / get_local_feature) issues one small launch
of the same kernel per call (inference only)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from elg_amd import _lib as L
from elg_amd import engine as eng


@dataclass
class Reset_State:
    problems: torch.Tensor               # (batch, problem, 2)


@dataclass
class Step_State:
    BATCH_IDX: torch.Tensor = None
    POMO_IDX: torch.Tensor = None
    current_node: torch.Tensor = None    # (batch, pomo)
    ninf_mask: torch.Tensor = None       # (batch, pomo, node)
    _env: object = None


class TSPEnv:
    def __init__(self, multi_width, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("elg_amd.TSPEnv runs on the GPU only (no CPU fallback)")
        torch.cuda.set_device(self.device)       # kernels launch on the current device / its current stream
        self.problem_size = None
        self.pomo_size = multi_width
        self.tsplib = False
        self.batch_size = None
        self.problems = None
        self.unscaled_problems = None
        self.selected_count = None
        self.current_node = None
        self.selected_node_list = None
        self.problem = None
        self._st_store = None
        self._needs_state = False
        self._dist = None

    def _finish_load(self, nbr=None):
        self.problems = self.problems.contiguous().float()
        self.batch_size, self.problem_size = self.problems.shape[0], self.problems.shape[1]
        self.problem = eng.Problem(L.PROBLEM_TSP, self.problems, None, nbr if nbr is not None else eng.nbr_tables(self.problems))
        self._dist = None
        dev = self.device
        self.BATCH_IDX = torch.arange(self.batch_size, device=dev)[:, None].expand(self.batch_size, self.pomo_size)
        self.POMO_IDX = torch.arange(self.pomo_size, device=dev)[None, :].expand(self.batch_size, self.pomo_size)

    @property
    def dist(self):
        if self._dist is None:
            self._dist = eng.dist_matrix(self.problems)
        return self._dist

    def load_random_problems(self, problems, aug_factor=1):
        """reference TSPEnv.py:53-67."""
        self.tsplib = False
        if aug_factor == 1 and not problems.is_cuda:
            host = problems.float()

            def upload():                  # host data only: on the preparation stream, next to the previous step's backward
                xy = eng.h2d(host, self.device)
                nbr = eng.nbr_tables(xy)
                return xy, nbr.idx, nbr.dist, nbr.theta
            xy, idx, dist, theta = eng.on_prep_stream(self.device, upload)
            self.problems = xy
            self._finish_load(eng.NbrTables(idx, dist, theta))
            return
        self.problems = eng.h2d(problems.float(), self.device)
        if aug_factor > 1:
            if aug_factor != 8:
                raise NotImplementedError
            self.problems = eng.aug8(self.problems)
        self._finish_load()

    def load_tsplib_problem(self, problems, unscaled_problems, aug_factor=1):
        """reference TSPEnv.py:69-85 (scaled coordinates drive the policy, raw ones the reported length)."""
        self.tsplib = True
        self.problems = problems.to(self.device).float()
        self.unscaled_problems = unscaled_problems.to(self.device).float().contiguous()
        if aug_factor > 1:
            if aug_factor != 8:
                raise NotImplementedError
            self.problems = eng.aug8(self.problems)
        self._finish_load()

    def reset(self):
        B, M, N = self.batch_size, self.pomo_size, self.problem_size
        dev = self.device
        self.selected_count = 0
        self.current_node = None
        self.selected_node_list = torch.zeros(B, M, 0, dtype=torch.long, device=dev)
        self._st_store = None                  # kernel state words of the step-wise protocol: built on first use (the fused
        self._needs_state = True               # rollout never reads them)
        self.step_state = Step_State(BATCH_IDX=self.BATCH_IDX, POMO_IDX=self.POMO_IDX, _env=self)
        self.step_state.ninf_mask = torch.zeros(B, M, N, device=dev)
        return Reset_State(self.problems), None, False

    @property
    def _st(self):
        if self._needs_state:
            self._needs_state = False
            B, M, N = self.batch_size, self.pomo_size, self.problem_size
            dev = self.device
            nw = (N + 63) // 64
            self._st_store = dict(cur=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  cnt=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  fin=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  first=torch.zeros(B, M, dtype=torch.int32, device=dev),
                                  load=torch.ones(B, M, dtype=torch.float32, device=dev),
                                  len=torch.zeros(B, M, dtype=torch.float32, device=dev),
                                  vis=torch.zeros(B, M, nw, dtype=torch.int64, device=dev))
        return self._st_store

    @_st.setter
    def _st(self, v):
        self._st_store = v
        self._needs_state = False

    def pre_step(self):
        return self.step_state, None, False

    def _state_args(self, a: L.RolloutArgs):
        st = self._st
        a.use_state = 1
        a.st_cur, a.st_cnt, a.st_fin, a.st_first = eng._ptr(st["cur"]), eng._ptr(st["cnt"]), eng._ptr(st["fin"]), eng._ptr(st["first"])
        a.st_load, a.st_len, a.st_vis = eng._ptr(st["load"]), eng._ptr(st["len"]), eng._ptr(st["vis"])

    def step(self, selected):
        """reference TSPEnv.py:108-133."""
        B, M = self.batch_size, self.pomo_size
        forced = selected.to(self.device, torch.int32).reshape(B, M, 1).contiguous()
        a = L.RolloutArgs()
        z = self.problems
        pol = eng.Policy(dict(K=z, V=z, PK=z, pb=z, Q1=z, Q2=z, wl=None), None, 0, 0.0, 0.0, 1.0, False, False)
        eng._fill_common(a, self.problem, pol, M, geometry=(8, min(M, 4), 0))
        a.Tmax, a.mode, a.max_steps, a.do_decode, a.do_update = 1, L.MODE_FORCED, 1, 0, 1
        a.forced, a.Tforced = eng._ptr(forced), 1
        self._state_args(a)
        L.check(L.lib().elg_rollout_fwd(C.byref(a), eng._stream()), "elg_rollout_fwd(step)")
        self.selected_count += 1
        self.current_node = selected.to(self.device).long()
        self.selected_node_list = torch.cat((self.selected_node_list, self.current_node[:, :, None]), dim=2)
        self.step_state.current_node = self.current_node
        self.step_state.ninf_mask.scatter_(2, self.current_node[:, :, None], float('-inf'))
        done = self.selected_count == self.problem_size
        reward = None
        if done:
            reward = self.compute_unscaled_distance() if self.tsplib else -self._get_travel_distance()
        return self.step_state, reward, done

    def get_local_feature(self):
        """(cur_dist, cur_theta, relative_xy) for protocol compatibility (reference TSPEnv.py:135-156)."""
        if self.current_node is None:
            return None, None, None
        B, M, N = self.batch_size, self.pomo_size, self.problem_size
        cur = self.current_node
        cur_dist = torch.gather(self.dist, 1, cur[:, :, None].expand(B, M, N))
        cxy = torch.gather(self.problems, 1, cur[:, :, None].expand(B, M, 2))
        rel = self.problems[:, None, :, :] - cxy[:, :, None, :]
        return cur_dist, torch.atan2(rel[..., 1], rel[..., 0]), rel

    def _get_travel_distance(self):
        return eng.route_length(self.problems, self.selected_node_list)

    def compute_unscaled_distance(self, solutions=None):
        if solutions is None:
            solutions = self.selected_node_list
        B = self.batch_size
        raw = self.unscaled_problems
        if raw.shape[0] != B:
            raw = raw.expand(B, -1, -1).contiguous()
        return -eng.route_length(raw, solutions.to(self.device), rounding=True)
