"""Greedy evaluation on a pickled test set with x8 augmentation -- the reference's `python test.py`
(gaocrr/ELG CVRP/test.py): prints the augmented / un-augmented mean cost and the wall-clock time."""
from __future__ import annotations

import os
import sys
import time

import torch
import yaml
from torch.utils.data import DataLoader

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import VRPDataset
from elg_amd.CVRP.utils import rollout


def test(dataloader, model, env, aug_factor):
    """reference test.py:14-56: best over POMO, then best over the 8 augmentations, averaged over instances."""
    model.eval()
    model.requires_grad_(False)
    aug_total, plain_total, batches = 0.0, 0.0, 0
    start = time.time()
    for batch in dataloader:
        env.load_random_problems(batch, aug_factor)
        reset_state, _, _ = env.reset()
        with torch.no_grad():
            model.pre_forward(reset_state)
            _, _, rewards = rollout(model=model, env=env, eval_type='greedy')
        best_pomo = rewards.reshape(aug_factor, batch['loc'].shape[0], env.multi_width).max(dim=2)[0]
        plain_total += float(-best_pomo[0].float().mean())
        aug_total += float(-best_pomo.max(dim=0)[0].float().mean())
        batches += 1
    torch.cuda.synchronize()
    elapsed = time.time() - start
    aug_cost, plain_cost = aug_total / batches, plain_total / batches
    print("Aug cost: {:.4f}".format(aug_cost))
    print("no aug Avg cost: {:.4f}, Wall-clock time: {:.2f}s".format(plain_cost, elapsed))
    return aug_cost, plain_cost


if __name__ == "__main__":
    with open('config.yml', 'r', encoding='utf-8') as fh:
        config = yaml.load(fh.read(), Loader=yaml.FullLoader)
    device = "cuda:{}".format(config['cuda_device_num'])
    p = config['params']
    model = CVRPModel(**config['model_params'])
    if config['model_params']['ensemble']:
        model.decoder.add_local_policy(device)
    if config['load_checkpoint']:
        model.load_state_dict(torch.load(config['load_checkpoint'], map_location=device)['model_state_dict'])
    model.to(device)
    data = VRPDataset(config['test_filename'], num_samples=p['test_size'])
    env = CVRPEnv(multi_width=p['multiple_width'], device=device)
    test(DataLoader(data, batch_size=p['test_batch_size']), model, env, p['aug_factor'])
