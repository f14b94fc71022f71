"""VRPLIB X / XXL evaluation -- the reference's `python test_vrplib.py` entry point (gaocrr/ELG
CVRP/test_vrplib.py): greedy construction, x8 augmentation, best of (augmentation x POMO), gap to the
best-known cost of the `.sol` file, results dumped to test_results/{name}_vrplib.json."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
import yaml

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd import vrplib_io as vrplib
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.utils import rollout


class VRPLib_Tester:
    def __init__(self, config, model=None):
        self.config = config
        mp = config['model_params']
        self.device = torch.device('cuda', config['cuda_device_num']) if config['use_cuda'] else torch.device('cpu')
        if model is None:
            model = CVRPModel(**mp)
            if mp['ensemble']:
                model.decoder.add_local_policy(self.device)
            if config['load_checkpoint']:
                ck = torch.load(config['load_checkpoint'], map_location=self.device)
                model.load_state_dict(ck['model_state_dict'])
        self.model = model.to(self.device)
        self.vrplib_path = 'VRPLib/Vrp-Set-X/' if config['vrplib_set'] == 'X' else 'VRPLib/Vrp-Set-XXL/'
        self.repeat_times = 1
        self.aug_factor = config['params']['aug_factor']

    def test_on_vrplib(self, names=None):
        files = sorted(f[:-4] for f in os.listdir(self.vrplib_path) if f.endswith('.vrp'))
        if names is not None:
            files = [f for f in files if f in names]
        results, total_time = [], 0.0
        for t in range(self.repeat_times):
            for name in files:
                optimal = vrplib.read_solution(os.path.join(self.vrplib_path, name + '.sol'))['cost']
                rec = {'run_idx': t}
                t0 = time.time()
                self.test_on_one_ins(name, rec, os.path.join(self.vrplib_path, name + '.vrp'), optimal)
                total_time += time.time() - t0
                results.append({'instance': name, 'optimal': optimal, 'record': [rec]})
                print("Instance Name {}: gap {:.4f}".format(name, rec['gap']))
        gaps = np.array([r['record'][-1]['gap'] for r in results])
        scale = np.array([r['record'][-1]['scale'] for r in results])
        summary = {"<200": 100 * gaps[scale <= 200].mean() if (scale <= 200).any() else float('nan'),
                   # the reference stores the >500 bucket under this key (test_vrplib.py:104-106)
                   "200-1000": 100 * gaps[scale > 500].mean() if (scale > 500).any() else float('nan'),
                   "total": 100 * gaps.mean()}
        print("Average gap total: {:.2f}%  Average time: {:.2f}s".format(summary["total"], total_time / max(len(results), 1)))
        os.makedirs('test_results', exist_ok=True)
        with open('test_results/' + self.config['name'] + '_vrplib.json', 'w') as f:
            json.dump(results + [summary], f)
        return results, summary

    def test_on_one_ins(self, name, result_dict, instance_file, optimal):
        instance = vrplib.read_instance(instance_file)
        problem_size = instance['node_coord'].shape[0] - 1
        env = CVRPEnv(min(problem_size, 1000), self.device)
        env.load_vrplib_problem(instance, aug_factor=self.aug_factor)
        reset_state, _, _ = env.reset()
        self.model.eval()
        self.model.requires_grad_(False)
        with torch.no_grad():
            self.model.pre_forward(reset_state)
            _, _, rewards = rollout(self.model, env, 'greedy')
        best = -rewards.reshape(self.aug_factor, 1, env.multi_width).max(dim=2)[0].max(dim=0)[0].float()
        result_dict['best_cost'] = best.cpu().numpy().tolist()[0]
        result_dict['scale'] = problem_size
        result_dict['gap'] = (result_dict['best_cost'] - optimal) / optimal


if __name__ == "__main__":
    with open('config.yml', 'r', encoding='utf-8') as fh:
        config = yaml.load(fh.read(), Loader=yaml.FullLoader)
    VRPLib_Tester(config=config).test_on_vrplib()
