"""TSPModel with the reference's interface (gaocrr/ELG TSP/TSPModel.py:11-64)."""
from __future__ import annotations

import ctypes as C
import random

import torch
import torch.nn as nn

from elg_amd import _lib as L
from elg_amd import encoder as enc_host
from elg_amd import engine as eng
from elg_amd.TSP.models import TSP_Decoder, TSP_Encoder


class TSPModel(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.encoder = TSP_Encoder(**model_params)
        self.decoder = TSP_Decoder(**model_params)
        self.encoded_nodes = None

    def _encoder_params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in enc_host.parameter_names(L.PROBLEM_TSP, len(self.encoder.layers))]

    def pre_forward(self, reset_state):
        """reference TSPModel.py:17-24: encoder + decoder.set_kv as one call into libelg_hip.so (elg_encoder_fwd)."""
        mp = self.model_params
        loc = self.decoder.fold_local()             # first: see CVRPModel.pre_forward (backward order of the two nodes)
        self.encoded_nodes, tables = enc_host.encode_and_fold(L.PROBLEM_TSP, reset_state.problems, None,
                                                              self._encoder_params(), int(mp['encoder_layer_num']),
                                                              int(mp['ff_hidden_dim']))
        self.decoder.set_tables(self.encoded_nodes, tables, loc)

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
