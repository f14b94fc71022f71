"""REINFORCE/POMO training for TSP -- the reference's `python train.py` (gaocrr/ELG TSP/train.py) on the
MI355X engine; same config.yml and checkpoint layout (weights/{name}_{ts}/model_epoch_{k}.pt)."""
from __future__ import annotations

import datetime
import os
import sys

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader

if __package__ in (None, ""):                      # `cd elg_amd/TSP && python train.py`, as the reference is run
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd.optim import Adam as Optimizer      # one-launch Adam, torch.optim.Adam-compatible checkpoints
from elg_amd import engine as eng
from elg_amd import parallel
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.TSP.generate_data import TSPDataset, generate_tsp_data
from elg_amd.TSP.utils import Logger, check_feasible, rollout, seed_everything


def pomo_loss(probs, rewards, scale_norm=True):
    """reference TSP/train.py:107-118 (scale only when no instance has a zero normaliser)."""
    return eng.pomo_loss(probs, rewards, scale_norm, guard_zero=True)     # csrc/elg_train.hip; GPU tensors only (no CPU path)


def train_step(model, env, optimizer, batch, scale_norm=True, bucket=None, world=1):
    env.load_random_problems(batch)
    reset_state, _, _ = env.reset()
    model.pre_forward(reset_state)
    solutions, probs, rewards = rollout(model=model, env=env, eval_type='sample')
    # the feasibility flags are computed right behind the rollout but read after the backward and the update are queued:
    # the tour length is known (N), so this would be the step's only host round trip before the backward
    pi = solutions[0]
    flags = eng.HostFetch(eng.feasibility_flags_launch(pi if pi.stride(1) == 1 else pi.contiguous(), None))
    optimizer.zero_grad()
    J = pomo_loss(probs, rewards, scale_norm)
    J.backward()
    if bucket is not None:
        bucket.allreduce(world)
    optimizer.step()
    flags.get()                     # TSP/train.py:105 calls check_feasible and drops its result; the wait bounds the run-ahead
    return J.detach(), rewards


def softmax(x):
    e = np.exp(x)
    return e / e.sum(axis=0)


def test_rollout(loader, env, model, weighted=False):
    """reference TSP/train.py:20-38: mean over batches of the best-of-POMO greedy tour length.
    weighted: return (sum of the instances' lengths, instances) -- a rank's share of a sharded validation."""
    total, batches, cost_sum, count = 0.0, 0, 0.0, 0
    for batch in loader:
        env.load_random_problems(batch)
        reset_state, _, _ = env.reset()
        model.eval()
        with torch.no_grad():
            model.pre_forward(reset_state)
            solutions, _, rewards = rollout(model=model, env=env, eval_type='greedy')
        check_feasible(solutions[0:1])
        best = -rewards.max(1)[0]
        total += float(best.mean())
        batches += 1
        cost_sum += float(best.double().sum())
        count += int(best.numel())
    return (cost_sum, count) if weighted else total / max(batches, 1)


def validate(model, multiple_width, device, mixed=True, data_dir='data'):
    """reference TSP/train.py:40-78 (the data/*.pkl validation sets next to this file)."""
    env = TSPEnv(multi_width=multiple_width, device=device)
    if mixed:
        sets = [('tsp_uniform100_1000_seed1234.pkl', 1000, 1000), ('tsp_cluster100_1000_seed1234.pkl', 1000, 1000),
                ('tsp_mixed100_1000_seed1234.pkl', 1000, 1000)]
    else:
        sets = [('tsp_100_val.pkl', 1000, 500), ('tsp_200_val.pkl', 1000, 500), ('tsp_500_val.pkl', 100, 10)]
    rank, world = parallel.collective_world()
    if world <= 1:
        return [test_rollout(DataLoader(TSPDataset(os.path.join(data_dir, f), num_samples=n), batch_size=bs), env, model)
                for f, n, bs in sets]
    # data parallel: every rank evaluates the instances rank, rank + world, ... of each set; the cost sums are added over the ranks
    def local_sums():                     # no collective in here: a rank that fails must still reach the guard's exchange
        sums = []
        for f, n, bs in sets:
            rows = TSPDataset(os.path.join(data_dir, f), num_samples=n)
            mine = [rows[i] for i in range(rank, len(rows), world)]
            s_, c_ = test_rollout(DataLoader(mine, batch_size=max(1, -(-bs // world))), env, model, weighted=True) if mine else (0.0, 0)
            sums += [s_, float(c_)]
        return sums
    # a rank that raises makes every rank raise here, instead of leaving the others in the sum below until its timeout
    tot = parallel.sum_over_ranks(parallel.guarded(local_sums))
    return [tot[2 * i] / max(tot[2 * i + 1], 1.0) for i in range(len(sets))]


def train(model, training, T, start_steps, train_steps, mixed, train_batch_size, problem_size, distribution,
          multiple_width, lr, device, logger, scale_norm, fileLogger, dir_path, log_step):
    rank, world, _ = parallel.world_info()
    if train_batch_size % world:
        raise ValueError(f"train_batch_size {train_batch_size} is not divisible by the {world} data-parallel ranks")
    env = TSPEnv(multi_width=multiple_width, device=device)
    distribution_ = dict(distribution)
    gaps = np.array([1, 1, 1])
    optimizer = Optimizer(model.parameters(), lr=lr, weight_decay=1e-6)
    bucket = parallel.make_bucket(model.parameters(), optimizer)
    for i in range(train_steps - start_steps + 1):
        model.train()
        if (i == T - start_steps) and training == 'joint':
            print("Enable joint training.")
            model.decoder.add_local_policy(device)
            parallel.broadcast_parameters(model)
            optimizer = Optimizer(model.parameters(), lr=lr, weight_decay=1e-6)
            bucket = parallel.make_bucket(model.parameters(), optimizer)
        if mixed:                                                   # curriculum: families weighted by their validation gaps
            kind = str(np.random.choice(['uniform', 'cluster', 'mixed'], size=1, p=softmax(gaps))[0])
            kind = parallel.broadcast_object(kind)                   # every rank must draw the same family
            distribution_['data_type'] = kind
        else:
            distribution_['data_type'] = 'uniform'
        batch = generate_tsp_data(batch_size=train_batch_size // world, problem_size=problem_size, distribution=distribution_)
        train_step(model, env, optimizer, batch, scale_norm, bucket, world)
        if (i + 1) % log_step == 0:
            # every rank evaluates its share of the validation sets (a rank that fails takes the others down with it: validate())
            val_info = validate(model, multiple_width, device, mixed)
            if rank == 0:
                fileLogger.log(val_info)
                if logger is not None:
                    logger.log({'val_100_cost': val_info[0], 'val_200_cost': val_info[1], 'val_500_cost': val_info[2]}, step=i)
                torch.save({'step': i, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict()},
                           dir_path + '/model_epoch_{}.pt'.format(int((i + 1) / log_step)))
            if mixed:                                                # the gaps drive every rank's next draws (same on every rank)
                opts = np.array([7.753418, 3.667576, 6.729566])      # reference solver means (TSP/train.py:148)
                gaps = (np.array(val_info) - opts) / opts


if __name__ == "__main__":
    with open('config.yml', 'r', encoding='utf-8') as fh:
        config = yaml.load(fh.read(), Loader=yaml.FullLoader)
    parallel.respect_cpu_quota()
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
