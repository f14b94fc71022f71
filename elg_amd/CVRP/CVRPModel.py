"""CVRPModel with the reference's interface (gaocrr/ELG CVRP/CVRPModel.py:10-75): encoder once per
batch, then node selection.  `one_step_rollout` keeps the per-step protocol (one small HIP launch per
call, inference only); training and evaluation go through utils.rollout, which fuses the whole
construction into one persistent launch."""
from __future__ import annotations

import ctypes as C
import gc
import os
import random

import torch
import torch.nn as nn

from elg_amd import _lib as L
from elg_amd import engine as eng
from elg_amd.CVRP.models import CVRP_Decoder, CVRP_Encoder


class _EncodeAndFold(nn.Module):
    """encoder + table folds as ONE static-shape callable, so that a training step can replay them (forward
    and backward) as two hipGraphs instead of ~400 eager launches (the encoder is launch-bound at B=64)."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder

    def forward(self, depot_xy, node_xy_demand):
        enc = self.encoder(depot_xy, node_xy_demand)
        t, loc = self.decoder.fold(enc)
        outs = [enc, t["K"], t["V"], t["PK"], t["pb"], t["Q1"], t["wl"]]
        if loc is not None:
            outs.append(loc)
        return tuple(outs)


class CVRPModel(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.encoder = CVRP_Encoder(**model_params)
        self.decoder = CVRP_Decoder(**model_params)
        self.encoded_nodes = None            # (batch, problem+1, embedding)
        self.__dict__["_graphs"] = {}        # (shape, param ids) -> graphed _EncodeAndFold (not a submodule)
        self.use_graphs = os.environ.get("ELG_HIPGRAPH", "1") != "0"

    def _graphed(self, depot_xy, node_xy_demand):
        """hipGraph-captured encoder+folds for this input shape / parameter set (training mode only)."""
        key = (tuple(depot_xy.shape), tuple(node_xy_demand.shape), str(depot_xy.device),
               tuple(id(p) for p in self.parameters()))
        g = self._graphs.get(key)
        if g is None:
            # retire the previous graph set at a quiescent point: destroying graph executables while another graph
            # is being launched (e.g. from the autograd thread) is not safe in the HIP runtime
            torch.cuda.synchronize()
            self._graphs.clear()
            gc.collect()
            try:
                mod = _EncodeAndFold(self.encoder, self.decoder)
                g = torch.cuda.make_graphed_callables(mod, (depot_xy.detach().clone(), node_xy_demand.detach().clone()))
            except Exception as e:          # capture is an optimisation only: eager PyTorch is the same math
                print(f"[elg_amd] hipGraph capture of the encoder failed ({type(e).__name__}: {e}); running eager")
                g = False
            self._graphs.clear()            # one live graph set (static buffers) at a time
            self._graphs[key] = g
        return g

    def pre_forward(self, reset_state):
        node_xy_demand = torch.cat((reset_state.node_xy, reset_state.node_demand[:, :, None]), dim=2)
        depot_xy = reset_state.depot_xy
        g = None
        if self.use_graphs and self.training and torch.is_grad_enabled() and depot_xy.is_cuda:
            g = self._graphed(depot_xy, node_xy_demand)
        if g:
            outs = g(depot_xy.contiguous(), node_xy_demand.contiguous())
            self.encoded_nodes = outs[0]
            tables = dict(K=outs[1], V=outs[2], PK=outs[3], pb=outs[4], Q1=outs[5], wl=outs[6], Q2=None)
            self.decoder.set_tables(self.encoded_nodes, tables, outs[7] if len(outs) > 7 else None)
            return
        self.encoded_nodes = self.encoder(depot_xy, node_xy_demand, reset_state.dist)
        self.decoder.set_kv(self.encoded_nodes)

    @staticmethod
    def draw_starts(problem_size, multi_width):
        """Second move of every trajectory: the reference's exact draw (CVRPModel.py:46-51) -- Python's
        `random`, the same nodes for every instance, values in [0, N) (may contain the depot)."""
        return random.sample(range(0, problem_size), multi_width)

    def one_step_rollout(self, state, cur_dist=None, cur_theta=None, xy=None, norm_demand=None, eval_type='greedy'):
        env = getattr(state, "_env", None)
        if env is None:
            raise RuntimeError("one_step_rollout needs a Step_State produced by elg_amd's CVRPEnv")
        B, M = env.batch_size, env.multi_width
        dev = env.device
        if state.selected_count == 0:
            return torch.zeros(B, M, dtype=torch.long, device=dev), torch.ones(B, M, device=dev)
        if state.selected_count == 1:
            starts = torch.tensor(self.draw_starts(env.problem_size, M), device=dev)
            return starts[None, :].expand(B, M), torch.ones(B, M, device=dev)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("the step-wise protocol is inference-only; train through utils.rollout()")
        a = L.RolloutArgs()
        eng._fill_common(a, env.problem, self.decoder.policy, M)
        a.Tmax, a.max_steps, a.do_decode, a.do_update = 1, 1, 1, 0
        a.mode = L.MODE_SAMPLE if eval_type == 'sample' else L.MODE_GREEDY
        a.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        sel = torch.zeros(B, M, 1, dtype=torch.int32, device=dev)
        pr = torch.ones(B, 1, M, dtype=torch.float32, device=dev)
        a.actions, a.probs = eng._ptr(sel), eng._ptr(pr)
        dummy = torch.zeros(M, dtype=torch.int32, device=dev)
        a.starts = eng._ptr(dummy)
        env._state_args(a)
        L.check(L.lib().elg_rollout_fwd(C.byref(a), eng._stream()), "elg_rollout_fwd(decode)")
        selected = sel[:, :, 0].long()
        if eval_type != 'sample':
            return selected, None
        prob = pr[:, 0, :]
        if not bool((prob != 0).all()):      # reference CVRPModel.py:67-68
            prob = prob + 1e-6
        return selected, prob
