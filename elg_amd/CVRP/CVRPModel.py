"""CVRPModel with the reference's interface (gaocrr/ELG CVRP/CVRPModel.py:10-75): encoder once per
batch, then node selection.  `one_step_rollout` keeps the per-step protocol (one small HIP launch per
call, inference only); training and evaluation go through utils.rollout, which fuses the whole
construction into one persistent launch."""
from __future__ import annotations

import ctypes as C
import random

import torch
import torch.nn as nn

from elg_amd import _lib as L
from elg_amd import encoder as enc_host
from elg_amd import engine as eng
from elg_amd.CVRP.models import CVRP_Decoder, CVRP_Encoder, local_policy_att


class CVRPModel(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.encoder = CVRP_Encoder(**model_params)
        self.decoder = CVRP_Decoder(**model_params)
        self.encoded_nodes = None            # (batch, problem+1, embedding)

    def _encoder_params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in enc_host.parameter_names(L.PROBLEM_CVRP, len(self.encoder.layers))]

    def pre_forward(self, reset_state):
        """reference CVRPModel.py:21-34: encoder + decoder.set_kv -- one call into libelg_hip.so (elg_encoder_fwd);
        with autograd enabled the result carries elg_encoder_bwd as its backward."""
        xy, demand = getattr(reset_state, "_xy", None), getattr(reset_state, "_demand", None)
        if xy is None:          # a Reset_State that was not produced by elg_amd's CVRPEnv
            xy = torch.cat((reset_state.depot_xy, reset_state.node_xy), dim=1)
            demand = torch.cat((torch.zeros_like(reset_state.depot_xy[:, :, 0]), reset_state.node_demand), dim=1)
        mp = self.model_params
        # (the local fold FIRST: autograd runs the younger node first, so the encoder's backward is queued before the fold's
        # backward has to wait for the side-stream row kernel -- engine.SIDE_LOCAL_BWD)
        loc = self.decoder.fold_local()
        self.encoded_nodes, tables = enc_host.encode_and_fold(L.PROBLEM_CVRP, xy, demand, self._encoder_params(),
                                                              int(mp['encoder_layer_num']), int(mp['ff_hidden_dim']))
        self.decoder.set_tables(self.encoded_nodes, tables, loc)

    @staticmethod
    def draw_starts(problem_size, multi_width):
        """Second move of every trajectory: the reference's exact draw (CVRPModel.py:46-51) -- Python's
        `random`, the same nodes for every instance, values in [0, N) (may contain the depot)."""
        return random.sample(range(0, problem_size), multi_width)

    def one_step_rollout(self, state, cur_dist=None, cur_theta=None, xy=None, norm_demand=None, eval_type='greedy'):
        return _one_step(self, self.decoder.policy, state, eval_type)


def _one_step(model, policy, state, eval_type):
    """One decode step of the reference's protocol (CVRPModel.py:36-75 / :86-131) as a single small launch."""
    env = getattr(state, "_env", None)
    if env is None:
        raise RuntimeError("one_step_rollout needs a Step_State produced by elg_amd's CVRPEnv")
    B, M = env.batch_size, env.multi_width
    dev = env.device
    if state.selected_count == 0:
        return torch.zeros(B, M, dtype=torch.long, device=dev), torch.ones(B, M, device=dev)
    if state.selected_count == 1:
        starts = torch.tensor(model.draw_starts(env.problem_size, M), device=dev)
        return starts[None, :].expand(B, M), torch.ones(B, M, device=dev)
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        raise RuntimeError("the step-wise protocol is inference-only; train through utils.rollout()")
    a = L.RolloutArgs()
    eng._fill_common(a, env.problem, policy, M)
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


class CVRPModel_local(nn.Module):
    """reference CVRPModel.py:78-131 (`training: only_local`): the local policy alone decodes -- no encoder, no global
    decoder, no distance penalty; logits = logit_clipping * tanh(u_local), u_local = 0 outside the k nearest open customers
    (+ depot).  Same kernels as CVRPModel: the decoder tables of the batch are zeros (pointer score 0 everywhere), the local
    policy's folded tables carry the model.  Parameter names as the reference's (`local_policy.*`)."""

    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.local_policy = local_policy_att(model_params, idx=0)
        self.policy = None

    draw_starts = staticmethod(CVRPModel.draw_starts)

    def pre_forward(self, reset_state):
        """The reference's pre_forward is empty; here the per-batch policy object is (re)built: folded local tables (with
        their backward when autograd is on) + zero decoder tables of the batch's shape."""
        depot = reset_state.depot_xy
        B, N1 = depot.shape[0], reset_state.node_xy.shape[1] + 1
        dev = next(self.parameters()).device
        key = (B, N1, str(dev))
        if getattr(self, "_zero_key", None) != key:
            z = torch.zeros(B, N1, eng.E, device=dev)
            self._zero = dict(K=z, V=z, PK=z, pb=torch.zeros(B, N1, device=dev), Q1=z, Q2=None, wl=torch.zeros(eng.E, device=dev))
            self._zero_key = key
        mp = self.model_params
        K = int(mp['local_size'][0])
        loc = self.local_policy.folded_tables(K + 1)
        self.policy = eng.Policy(self._zero, loc, K, 0.0, float(mp['logit_clipping']), 1.0, True, False,
                                 bool(mp.get('euclidean', False)))

    def one_step_rollout(self, state, cur_dist=None, cur_theta=None, xy=None, norm_demand=None, eval_type='greedy'):
        return _one_step(self, self.policy, state, eval_type)
