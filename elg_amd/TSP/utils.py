"""rollout and helpers with the reference's names (gaocrr/ELG TSP/utils.py)."""
from __future__ import annotations

import json

import torch

from elg_amd import _lib as L
from elg_amd import engine as eng
from elg_amd.CVRP.utils import seed_everything  # noqa: F401  (same function in both reference trees)


def rollout(model, env, eval_type='greedy'):
    """reference TSP/utils.py:7-26 as one persistent HIP launch; T = problem_size exactly."""
    env.reset()
    B, M, N = env.batch_size, env.pomo_size, env.problem_size
    pol = model.decoder.policy
    if pol is None:
        raise RuntimeError("call model.pre_forward(reset_state) before rollout")
    starts = torch.tensor(model.draw_starts(N, M), dtype=torch.int32)
    mode = L.MODE_SAMPLE if eval_type == 'sample' else L.MODE_GREEDY
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if mode == L.MODE_SAMPLE else 0
    needs_grad = (eval_type != 'greedy' and torch.is_grad_enabled()
                  and any(p.requires_grad for p in model.parameters()))
    res = eng.rollout_forward(env.problem, pol, M, starts, mode, seed=seed, train=needs_grad, need_probs=eval_type != 'greedy')
    actions = res.actions[:, :, :N].long()
    env.selected_count = N
    env.selected_node_list = actions
    env.current_node = actions[:, :, -1]
    reward = env.compute_unscaled_distance() if env.tsplib else res.reward
    if eval_type == 'greedy':
        return actions, None, reward
    probs = eng.chosen_probs(env.problem, pol, M, res, N) if needs_grad else res.probs[:, :N, :]
    return actions, probs, reward


def augment_xy_data_by_8_fold(problems):
    return eng.aug8(problems)


def check_feasible(pi):
    """reference TSP/utils.py:72-78: every node exactly once.  pi (1, multi, problem).  One HIP launch."""
    pi = pi.squeeze(0)
    if pi.stride(1) != 1:
        pi = pi.contiguous()
    bad, _ = eng.feasibility_flags(pi.long(), None)
    return not bad


class Logger(object):
    def __init__(self, filename, config):
        self.filename = filename
        self.logger = config
        self.logger['result'] = {'val_100': [], 'val_200': [], 'val_500': []}

    def log(self, info):
        for key, v in zip(('val_100', 'val_200', 'val_500'), info):
            self.logger['result'][key].append(float(v))
        with open(self.filename, 'w') as f:
            json.dump(self.logger, f)
