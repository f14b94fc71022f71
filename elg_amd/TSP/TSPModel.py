"""TSPModel with the reference's interface (gaocrr/ELG TSP/TSPModel.py:11-64)."""
from __future__ import annotations

import ctypes as C
import gc
import os
import random

import torch
import torch.nn as nn

from elg_amd import _lib as L
from elg_amd import engine as eng
from elg_amd.TSP.models import TSP_Decoder, TSP_Encoder


class _EncodeAndFold(nn.Module):
    """encoder + table folds as one static-shape callable (two hipGraphs per training step, see CVRPModel)."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder

    def forward(self, problems):
        enc = self.encoder(problems)
        t, loc = self.decoder.fold(enc)
        outs = [enc, t["K"], t["V"], t["PK"], t["pb"], t["Q1"], t["Q2"]]
        if loc is not None:
            outs.append(loc)
        return tuple(outs)


class TSPModel(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.encoder = TSP_Encoder(**model_params)
        self.decoder = TSP_Decoder(**model_params)
        self.encoded_nodes = None
        self.__dict__["_graphs"] = {}
        self.use_graphs = os.environ.get("ELG_HIPGRAPH", "1") != "0"

    def _graphed(self, problems):
        key = (tuple(problems.shape), str(problems.device), tuple(id(p) for p in self.parameters()))
        g = self._graphs.get(key)
        if g is None:
            # retire the previous graph set at a quiescent point: destroying graph executables while another graph
            # is being launched (e.g. from the autograd thread) is not safe in the HIP runtime
            torch.cuda.synchronize()
            self._graphs.clear()
            gc.collect()
            try:
                g = torch.cuda.make_graphed_callables(_EncodeAndFold(self.encoder, self.decoder),
                                                      (problems.detach().clone(),))
            except Exception as e:          # capture is an optimisation only: eager PyTorch is the same math
                print(f"[elg_amd] hipGraph capture of the encoder failed ({type(e).__name__}: {e}); running eager")
                g = False
            self._graphs.clear()
            self._graphs[key] = g
        return g

    def pre_forward(self, reset_state):
        problems = reset_state.problems
        g = None
        if self.use_graphs and self.training and torch.is_grad_enabled() and problems.is_cuda:
            g = self._graphed(problems)
        if g:
            outs = g(problems.contiguous())
            self.encoded_nodes = outs[0]
            tables = dict(K=outs[1], V=outs[2], PK=outs[3], pb=outs[4], Q1=outs[5], Q2=outs[6], wl=None)
            self.decoder.set_tables(self.encoded_nodes, tables, outs[7] if len(outs) > 7 else None)
            return
        self.encoded_nodes = self.encoder(problems)
        self.decoder.set_kv(self.encoded_nodes)

    @staticmethod
    def draw_starts(problem_size, pomo_size):
        """First move: the reference's draw (TSPModel.py:31) -- a random permutation of range(pomo)."""
        return random.sample(range(0, pomo_size), pomo_size)

    def one_step_rollout(self, state, cur_dist=None, cur_theta=None, xy=None, eval_type='greedy'):
        env = getattr(state, "_env", None)
        if env is None:
            raise RuntimeError("one_step_rollout needs a Step_State produced by elg_amd's TSPEnv")
        B, M = env.batch_size, env.pomo_size
        dev = env.device
        if state.current_node is None:
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
        return (selected, pr[:, 0, :]) if eval_type == 'sample' else (selected, None)
