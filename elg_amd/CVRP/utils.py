"""rollout / augmentation / feasibility helpers with the reference's names (gaocrr/ELG CVRP/utils.py)."""
from __future__ import annotations

import json
import random

import numpy as np
import torch

from elg_amd import _lib as L
from elg_amd import engine as eng


def rollout(model, env, eval_type='greedy'):
    """reference utils.py:7-29 -> (actions (B,M,T) int64, probs (B,T,M) | None, reward (B,M)).

    The reference loops `get_cur_feature -> one_step_rollout -> step` in Python until every trajectory
    is finished; here the whole construction is ONE persistent HIP launch.  With autograd enabled the
    returned probabilities carry a grad_fn whose backward is the replay kernel pair of csrc/elg_bwd.hip."""
    env.reset()
    B, M, N = env.batch_size, env.multi_width, env.problem_size
    pol = model.policy if hasattr(model, 'policy') else model.decoder.policy      # CVRPModel_local owns its policy
    if pol is None:
        raise RuntimeError("call model.pre_forward(reset_state) before rollout")
    starts = torch.tensor(model.draw_starts(N, M), dtype=torch.int32)
    mode = L.MODE_SAMPLE if eval_type == 'sample' else L.MODE_GREEDY
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if mode == L.MODE_SAMPLE else 0
    needs_grad = (eval_type != 'greedy' and torch.is_grad_enabled()
                  and any(p.requires_grad for p in model.parameters()))
    res = eng.rollout_forward(env.problem, pol, M, starts, mode, seed=seed, train=needs_grad, need_probs=eval_type != 'greedy')
    T, zero_prob = eng.rollout_stats(res)           # the one host sync of the rollout
    actions = res.actions[:, :, :T].long()
    env.selected_count = T
    env.selected_node_list = actions
    env.current_node = actions[:, :, -1]
    reward = env.compute_unscaled_reward() if env.vrplib else res.reward
    if eval_type == 'greedy':
        return actions, None, reward
    probs = eng.chosen_probs(env.problem, pol, M, res, T) if needs_grad else res.probs[:, :T, :]
    if zero_prob:
        # reference CVRPModel.py:67-68: a step in which some chosen probability is exactly 0 gets +1e-6
        zero_step = (probs.detach() == 0).flatten(2).any(dim=2).any(dim=0)
        probs = probs + 1e-6 * zero_step[None, :, None].to(probs.dtype)
    return actions, probs, reward


class TrainRollout:
    """A sampled training rollout whose one host sync is deferred (rollout_train).  `probs` (B, Tcap, M) is differentiable and
    padded with ones past each trajectory's end (log 1 = 0: the loss does not see the padding); `reward` (B, M).  finish()
    waits for the rollout's length and the feasibility flags, which were copied to the host right behind the rollout -- by
    then the backward is already queued, so the GPU never idles on the host -- and returns the actions (B, M, T)."""

    def __init__(self, env, res, probs, fetch, checked, zero_steps=None, T_dev=None):
        self.env, self.res, self.reward = env, res, res.reward
        self.T_dev = T_dev                  # device int32: the longest trajectory's step count (probs beyond it are exactly 1)
        # chosen probabilities as the kernels produced them + the steps on which one of them was exactly 0 (device flags): the
        # loss kernel adds the reference's 1e-6 there (train.pomo_loss(..., zero_steps=)); `probs` forms the sum for other callers
        self.probs_raw, self.zero_steps = probs, zero_steps
        self._probs = None
        self._fetch, self._checked = fetch, checked

    @property
    def probs(self):
        if self._probs is None:
            self._probs = self.probs_raw if self.zero_steps is None else \
                torch.add(self.probs_raw, self.zero_steps[None, :, None], alpha=1e-6)      # exact + 0.0 unless a chosen probability was 0
        return self._probs

    def finish(self, want_actions: bool = False):
        vals = self._fetch.get()
        T = int(vals[0])
        # env.selected_node_list / current_node: formed from the engine's tours when somebody reads them (CVRPEnv.set_tours_lazy)
        self.env.set_tours_lazy(self.res.actions, T)
        if self._checked:                                   # utils.check_feasible's assertions (reference utils.py:90-119)
            assert not vals[2], "Invalid tour"
            assert not vals[3], "Used more than capacity"
        return self.env.selected_node_list if want_actions else None


def rollout_train(model, env, check_demand=None):
    """The sampled rollout of a training step (reference train.py:108-111 = rollout(..., 'sample') + check_feasible of
    instance 0) without a host round trip before the backward: the rollout's length stays on the device
    (elg_decoder_bwd_args.T_dev), the +1e-6 of CVRPModel.py:67-68 is applied per step from device flags, the feasibility
    kernel runs over the padded tours.  Needs the rows a training forward saves (N + 1 <= 128)."""
    env.reset()
    B, M, N = env.batch_size, env.multi_width, env.problem_size
    pol = model.policy if hasattr(model, 'policy') else model.decoder.policy
    if pol is None:
        raise RuntimeError("call model.pre_forward(reset_state) before rollout")
    starts = torch.tensor(model.draw_starts(N, M), dtype=torch.int32)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    res = eng.rollout_forward(env.problem, pol, M, starts, L.MODE_SAMPLE, seed=seed, train=True)
    stats, zsteps, block = eng.rollout_stats_launch(res)     # block = [T, zero_prob, invalid_tour, over_capacity]
    if check_demand is not None:
        eng.feasibility_flags_launch(res.actions[0].long(), check_demand.reshape(-1), out=block[2:4])
    fetch = eng.HostFetch(block)
    Tcap = res.probs.shape[1]
    probs = eng.chosen_probs(env.problem, pol, M, res, Tcap, T_dev=stats)
    return TrainRollout(env, res, probs, fetch, check_demand is not None, zero_steps=zsteps, T_dev=stats)


def augment_xy_data_by_8_fold(problems):
    """reference utils.py:69-87 (elg_aug8 kernel)."""
    return eng.aug8(problems)


def check_feasible(pi, demand):
    """reference utils.py:90-119: every customer exactly once, capacity never exceeded (the reference's sequential
    fp32 scan).  pi (1, multi, T) node ids, demand (1, problem).  One HIP launch (elg_check_feasible)."""
    pi = pi.squeeze(0)
    if pi.stride(1) != 1:
        pi = pi.contiguous()
    bad, over = eng.feasibility_flags(pi.long(), demand.reshape(-1))
    assert not bad, "Invalid tour"
    assert not over, "Used more than capacity"


def seed_everything(seed=2022):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class Logger(object):
    """JSON log of the validation costs, same file format as the reference (utils.py:130-151)."""

    def __init__(self, filename, config):
        self.filename = filename
        self.logger = config
        self.logger['result'] = {'val_100': [], 'val_200': [], 'val_500': []}

    def log(self, info):
        for key, v in zip(('val_100', 'val_200', 'val_500'), info):
            self.logger['result'][key].append(v)
        with open(self.filename, 'w') as f:
            json.dump(self.logger, f)
