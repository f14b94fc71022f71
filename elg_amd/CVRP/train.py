"""REINFORCE training with the POMO shared baseline -- the reference's `python train.py` entry point
(gaocrr/ELG CVRP/train.py) on the MI355X engine.  Same config.yml, same checkpoint dictionary
({'step','model_state_dict','optimizer_state_dict'} at weights/{name}_{ts}_{seed}/model_epoch_{k}.pt), same
log JSON.  New: launched under `python -m torch.distributed.run` it shards the batch over the GPUs of the
node and all-reduces the gradient over RCCL (elg_amd/parallel.py)."""
from __future__ import annotations

import datetime
import os
import sys

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader

if __package__ in (None, ""):                      # `cd elg_amd/CVRP && python train.py`, as the reference is run
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd.optim import Adam as Optimizer      # one-launch Adam, torch.optim.Adam-compatible checkpoints
from elg_amd import engine as eng
from elg_amd import parallel
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel, CVRPModel_local
from elg_amd.CVRP.generate_data import VRPDataset, generate_vrp_data
from elg_amd.CVRP.utils import Logger, check_feasible, rollout, rollout_train, seed_everything


def softmax(x):
    e = np.exp(x)
    return e / e.sum(axis=0)


def pomo_loss(probs, rewards, scale_norm=True, zero_steps=None, T_dev=None):
    """reference train.py:112-121: shared baseline = mean reward over the POMO trajectories of an instance.
    zero_steps: utils.rollout_train's device flags of the steps with a chosen probability of exactly 0 (the reference's +1e-6);
    T_dev: its device-resident step count (the padded steps behind it are not read)."""
    return eng.pomo_loss(probs, rewards, scale_norm, zero_steps=zero_steps, T_dev=T_dev)     # csrc/elg_train.hip; GPU tensors only (no CPU path)


_NOTED = set()


def _note_host_sync_path(N1, vrplib, ens, wide=False):
    """Say ONCE per configuration that a training step runs the reference's own sequence (rollout's length read back on the host
    before the loss is built: ~0.4 ms of idle GPU per step at the bench scale) instead of the deferred-sync path."""
    key = (N1 > 128, bool(vrplib), ens > 1, bool(wide))
    if key in _NOTED:
        return
    _NOTED.add(key)
    why = [w for w, c in (("N + 1 > 128 nodes", N1 > 128), ("VRPLIB instance", vrplib), ("ensemble_size > 1", ens > 1),
                               ("local_size > 47", wide)) if c]
    print(f"[elg_amd] train_step: host-synchronised rollout path ({', '.join(why)}); the deferred-sync path covers N + 1 <= 128, "
          "one local policy, local_size <= 47", file=sys.stderr, flush=True)


def train_step(model, env, optimizer, batch, scale_norm=True, bucket=None, world=1, check=True):
    """One optimisation step (reference train.py:103-125): load -> encoder -> sampled rollout -> loss ->
    backward -> [gradient all-reduce] -> Adam.  Returns (loss, rewards)."""
    env.load_random_problems(batch)
    reset_state, _, _ = env.reset()
    model.pre_forward(reset_state)
    pol = model.policy if hasattr(model, 'policy') else model.decoder.policy
    if env.problem.N1 > 128 or env.vrplib or pol.ens > 1 or pol.wide_slots:
        # sizes / variants whose training forward saves no rows: the reference's sequence with the host sync inside rollout()
        _note_host_sync_path(env.problem.N1, env.vrplib, pol.ens, pol.wide_slots)
        solutions, probs, rewards = rollout(model=model, env=env, eval_type='sample')
        if check:
            check_feasible(solutions[0:1], reset_state.node_demand[0:1])
        optimizer.zero_grad()
        J = pomo_loss(probs, rewards, scale_norm)
        J.backward()
        if bucket is not None:
            bucket.allreduce(world)
        optimizer.step()
        return J.detach(), rewards
    # Same operations; the rollout's length and the feasibility flags are read AFTER the backward and the update are queued.
    # Intentional difference in ORDER to the reference (train.py:111 asserts before zero_grad / backward): the assertion still
    # fires in the step that produced the infeasible tour, but that step's update has already been queued -- the run stops
    # either way (the reference never catches the AssertionError), and the GPU does not idle 0.4 ms per step on the read-back.
    # finish() runs on every exit path, so the flags are always read.
    ro = rollout_train(model, env, reset_state.node_demand[0] if check else None)
    try:
        optimizer.zero_grad()
        J = pomo_loss(ro.probs_raw, ro.reward, scale_norm, zero_steps=ro.zero_steps, T_dev=ro.T_dev)
        J.backward(eng.unit_grad(J.device))          # = J.backward(), minus the fill and the product of the implicit cotangent
        if bucket is not None:
            bucket.allreduce(world)
        optimizer.step()
    finally:
        ro.finish()                 # the step's host sync + the feasibility assertions
    return J.detach(), ro.reward


def test_rollout(loader, env, model, weighted=False):
    """reference train.py:20-40: mean over the batches of the batch's mean best-of-POMO greedy cost.
    weighted: return (sum of the instances' costs, instances) instead -- what a rank contributes to a sharded validation."""
    total, batches, cost_sum, count = 0.0, 0, 0.0, 0
    for batch in loader:
        env.load_random_problems(batch)
        reset_state, _, _ = env.reset()
        model.eval()
        with torch.no_grad():
            model.pre_forward(reset_state)
            solutions, _, rewards = rollout(model=model, env=env, eval_type='greedy')
        check_feasible(solutions[0:1], reset_state.node_demand[0:1])
        best = -rewards.max(1)[0]
        total += float(best.mean())
        batches += 1
        cost_sum += float(best.double().sum())
        count += int(best.numel())
    return (cost_sum, count) if weighted else total / max(batches, 1)


def validate(model, multiple_width, device, mixed=True, data_dir='data'):
    """reference train.py:42-80 (the data/*.pkl validation sets next to this file).
    Under data parallelism EVERY rank calls this: the instances of a set are dealt round-robin over the ranks (they are
    independent: SURVEY 8e "replicas only"), the per-set cost sums are added over the ranks and every rank returns the same
    means -- no rank idles through an evaluation, nothing is broadcast afterwards.  One process: the reference's arithmetic."""
    env = CVRPEnv(multi_width=multiple_width, device=device)
    if mixed:
        sets = [('vrp_uniform100_1000_seed1234.pkl', 1000, 1000), ('vrp_cluster100_1000_seed1234.pkl', 1000, 1000),
                ('vrp_mixed100_1000_seed1234.pkl', 1000, 1000)]
    else:
        sets = [('vrp100_val.pkl', 1000, 1000), ('vrp200_val.pkl', 1000, 1000), ('vrp500_val.pkl', 100, 10)]
    rank, world = parallel.collective_world()
    if world <= 1:
        return [test_rollout(DataLoader(VRPDataset(os.path.join(data_dir, f), num_samples=n), batch_size=bs), env, model)
                for f, n, bs in sets]
    def local_sums():                     # no collective in here: a rank that fails must still reach the guard's exchange
        sums = []
        for f, n, bs in sets:
            rows = VRPDataset(os.path.join(data_dir, f), num_samples=n)
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
    env = CVRPEnv(multi_width=multiple_width, device=device)
    distribution_ = dict(distribution)
    gaps = np.array([1, 1, 1])
    optimizer = Optimizer(model.parameters(), lr=lr, weight_decay=1e-6)
    bucket = parallel.make_bucket(model.parameters(), optimizer)
    local_batch = train_batch_size // world
    for i in range(train_steps - start_steps + 1):
        model.train()
        if i == T - start_steps and training == 'joint':           # enable the local policy (train.py:93-96)
            print("Enable joint training.")
            model.decoder.add_local_policy(device)
            parallel.broadcast_parameters(model)
            optimizer = Optimizer(model.parameters(), lr=lr, weight_decay=1e-6)
            bucket = parallel.make_bucket(model.parameters(), optimizer)
        if mixed:
            kind = str(np.random.choice(['uniform', 'cluster', 'mixed'], size=1, p=softmax(gaps))[0])
            distribution_['data_type'] = parallel.broadcast_object(kind)     # every rank must draw the same family
        else:
            distribution_['data_type'] = 'uniform'
        batch = generate_vrp_data(batch_size=local_batch, problem_size=problem_size, distribution=distribution_)
        train_step(model, env, optimizer, batch, scale_norm, bucket, world)
        if (i + 1) % log_step == 0:
            # every rank evaluates its share of the validation sets (a rank that fails takes the others down with it: validate())
            val_info = validate(model, multiple_width, device, mixed)
            if rank == 0:
                fileLogger.log(val_info)
                if logger is not None:
                    logger.log({'val_100_cost': val_info[0], 'val_300_cost': val_info[1], 'val_500_cost': val_info[2]}, step=i)
                torch.save({'step': i, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict()},
                           dir_path + '/model_epoch_{}.pt'.format(int((i + 1) / log_step)))
            if mixed:                                               # the gaps drive every rank's next draws (same on every rank)
                opts = np.array([15.740834, 7.909336, 14.294179])  # reference solver means (train.py:146)
                gaps = (np.array(val_info) - opts) / opts


if __name__ == "__main__":
    with open('config.yml', 'r', encoding='utf-8') as fh:
        config = yaml.load(fh.read(), Loader=yaml.FullLoader)
    parallel.respect_cpu_quota()
    rank, world, local = parallel.init_distributed()
    if config['training'] not in ('joint', 'only_global', 'only_local'):
        raise ValueError("training: {} (config.yml:9 allows joint, only_local, only_global)".format(config['training']))
    p = config['params']
    device = "cuda:{}".format(local if world > 1 else config['cuda_device_num']) if config['use_cuda'] else 'cpu'
    seed_everything(config['seed'] + rank)
    ts = datetime.datetime.utcnow() + datetime.timedelta(hours=+8)
    ts_name = f'-ts{ts.month}-{ts.day}-{ts.hour}-{ts.minute}-{ts.second}'
    dir_path = 'weights/{}_{}_{}'.format(config['name'], ts_name, config['seed'])
    fileLogger = None
    if rank == 0:
        os.makedirs(dir_path, exist_ok=True)
        os.makedirs('log', exist_ok=True)
        fileLogger = Logger('log/{}_{}'.format(config['name'], ts_name), config)
    model = CVRPModel(**config['model_params'])
    if config['training'] == 'only_local':                         # reference train.py:198-200
        model = CVRPModel_local(**config['model_params'])
    if config['load_checkpoint'] is not None:
        checkpoint = torch.load(config['load_checkpoint'], map_location=device)
        if any(k.startswith('decoder.local_policies') for k in checkpoint['model_state_dict']):
            model.decoder.add_local_policy(device)
        model.load_state_dict(checkpoint['model_state_dict'])
    model.to(device)
    parallel.broadcast_parameters(model)
    train(model=model, training=config['training'], T=p['T'], start_steps=p['start_steps'], train_steps=p['train_steps'],
          mixed=p['mixed'], train_batch_size=p['train_batch_size'], problem_size=p['problem_size'],
          distribution=config['distribution'], multiple_width=p['multiple_width'], lr=p['learning_rate'], device=device,
          logger=None, scale_norm=p['scale_norm'], fileLogger=fileLogger, dir_path=dir_path, log_step=p['log_step'])
