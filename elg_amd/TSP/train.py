"""REINFORCE/POMO training for TSP -- the reference's `python train.py` (gaocrr/ELG TSP/train.py) on the
MI355X engine; same config.yml and checkpoint layout (weights/{name}_{ts}/model_epoch_{k}.pt)."""
from __future__ import annotations

import datetime
import os
import sys

import numpy as np
import torch
import yaml
from elg_amd.optim import Adam as Optimizer      # one-launch Adam, torch.optim.Adam-compatible checkpoints

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd import engine as eng
from elg_amd import parallel
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.TSP.generate_data import generate_tsp_data
from elg_amd.TSP.utils import Logger, check_feasible, rollout, seed_everything


def pomo_loss(probs, rewards, scale_norm=True):
    """reference TSP/train.py:107-118 (scale only when no instance has a zero normaliser)."""
    return eng.pomo_loss(probs, rewards, scale_norm, guard_zero=True)     # csrc/elg_train.hip; GPU tensors only (no CPU path)


def train_step(model, env, optimizer, batch, scale_norm=True, bucket=None, world=1):
    env.load_random_problems(batch)
    reset_state, _, _ = env.reset()
    model.pre_forward(reset_state)
    solutions, probs, rewards = rollout(model=model, env=env, eval_type='sample')
    check_feasible(solutions[0:1])
    optimizer.zero_grad()
    J = pomo_loss(probs, rewards, scale_norm)
    J.backward()
    if bucket is not None:
        bucket.allreduce(world)
    optimizer.step()
    return J.detach(), rewards


def train(model, training, T, start_steps, train_steps, mixed, train_batch_size, problem_size, distribution,
          multiple_width, lr, device, logger, scale_norm, fileLogger, dir_path, log_step):
    rank, world, _ = parallel.world_info()
    env = TSPEnv(multi_width=multiple_width, device=device)
    distribution_ = dict(distribution)
    optimizer = Optimizer(model.parameters(), lr=lr, weight_decay=1e-6)
    bucket = parallel.GradBucket(model.parameters(), optimizer) if world > 1 else None
    for i in range(train_steps - start_steps + 1):
        model.train()
        if (i == T - start_steps) and training == 'joint':
            print("Enable joint training.")
            model.decoder.add_local_policy(device)
            parallel.broadcast_parameters(model)
            optimizer = Optimizer(model.parameters(), lr=lr, weight_decay=1e-6)
            bucket = parallel.GradBucket(model.parameters(), optimizer) if world > 1 else None
        distribution_['data_type'] = 'uniform' if not mixed else str(np.random.choice(['uniform', 'cluster', 'mixed']))
        batch = generate_tsp_data(batch_size=train_batch_size // world, problem_size=problem_size, distribution=distribution_)
        train_step(model, env, optimizer, batch, scale_norm, bucket, world)
        if (i + 1) % log_step == 0 and rank == 0:
            torch.save({'step': i, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict()},
                       dir_path + '/model_epoch_{}.pt'.format(int((i + 1) / log_step)))


if __name__ == "__main__":
    with open('config.yml', 'r', encoding='utf-8') as fh:
        config = yaml.load(fh.read(), Loader=yaml.FullLoader)
    rank, world, local = parallel.init_distributed()
    if config['training'] not in ('joint', 'only_global'):
        # the reference's own entry point names a class it never defines for this mode (train.py:199-200)
        raise NotImplementedError("training: {} is not built (the reference's 'only_local_att' model class does not exist either)".format(config['training']))
    p = config['params']
    device = "cuda:{}".format(local if world > 1 else config['cuda_device_num'])
    seed_everything(config['seed'] + rank)
    ts = datetime.datetime.utcnow() + datetime.timedelta(hours=+8)
    ts_name = f'-ts{ts.month}-{ts.day}-{ts.hour}-{ts.minute}-{ts.second}'
    dir_path = 'weights/{}_{}'.format(config['name'], ts_name)
    fileLogger = None
    if rank == 0:
        os.makedirs(dir_path, exist_ok=True)
        os.makedirs('log', exist_ok=True)
        fileLogger = Logger('log/{}_{}'.format(config['name'], ts_name), config)
    model = TSPModel(**config['model_params'])
    if config['load_checkpoint'] is not None:
        ck = torch.load(config['load_checkpoint'], map_location=device)
        if any(k.startswith('decoder.local_policy_0') for k in ck['model_state_dict']):
            model.decoder.add_local_policy(device)
        model.load_state_dict(ck['model_state_dict'])
    model.to(device)
    parallel.broadcast_parameters(model)
    train(model=model, training=config['training'], T=p['T'], start_steps=p['start_steps'], train_steps=p['train_steps'],
          mixed=p['mixed'], train_batch_size=p['train_batch_size'], problem_size=p['problem_size'],
          distribution=config['distribution'], multiple_width=p['multiple_width'], lr=p['learning_rate'], device=device,
          logger=None, scale_norm=p['scale_norm'], fileLogger=fileLogger, dir_path=dir_path, log_step=p['log_step'])
