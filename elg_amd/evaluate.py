"""Greedy evaluation shared by the CVRP / TSP `test.py` entry points: x-fold augmented batch -> one fused greedy
rollout -> best over POMO, then best over augmentations (the protocol of the reference's test.py files)."""
from __future__ import annotations

import time
from typing import Callable, Tuple

import torch
import yaml


def best_costs(rewards: torch.Tensor, aug_factor: int, n_inst: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """rewards (aug*n_inst, pomo) -> (cost without augmentation (n_inst,), cost with augmentation (n_inst,))."""
    per_aug = -rewards.reshape(aug_factor, n_inst, -1).max(dim=2).values.float()
    return per_aug[0], per_aug.min(dim=0).values


def evaluate_loader(loader, model, env, aug_factor: int, rollout: Callable, n_inst_of: Callable) -> Tuple[float, float]:
    """Mean augmented / plain greedy cost over a loader; prints the two lines the reference's test.py prints."""
    model.eval()
    model.requires_grad_(False)
    sums = torch.zeros(2, dtype=torch.float64)
    n_batches = 0
    t0 = time.time()
    with torch.no_grad():
        for batch in loader:
            env.load_random_problems(batch, aug_factor)
            state, _, _ = env.reset()
            model.pre_forward(state)
            rewards = rollout(model=model, env=env, eval_type='greedy')[2]
            plain, aug = best_costs(rewards, aug_factor, n_inst_of(batch))
            sums += torch.stack((aug.mean(), plain.mean())).double().cpu()
            n_batches += 1
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    aug_cost, plain_cost = (sums / max(n_batches, 1)).tolist()
    print("Aug cost: {:.4f}".format(aug_cost))
    print("no aug Avg cost: {:.4f}, Wall-clock time: {:.2f}s".format(plain_cost, time.time() - t0))
    return aug_cost, plain_cost


def load_run_config(path: str = 'config.yml'):
    with open(path, 'r', encoding='utf-8') as fh:
        cfg = yaml.load(fh.read(), Loader=yaml.FullLoader)
    return cfg, "cuda:{}".format(cfg['cuda_device_num'])


def build_model(model_cls, cfg, device):
    """Model of the config, local policy attached when `ensemble`, checkpoint loaded when one is named."""
    model = model_cls(**cfg['model_params'])
    if cfg['model_params']['ensemble']:
        model.decoder.add_local_policy(device)
    if cfg['load_checkpoint']:
        model.load_state_dict(torch.load(cfg['load_checkpoint'], map_location=device)['model_state_dict'])
    return model.to(device)
