"""TSPLIB evaluation -- the reference's `python test_tsplib.py` (gaocrr/ELG TSP/test_tsplib.py): greedy, x8
augmentation, POMO = N, isotropic min-max scaling (:128), rounded length on the raw coordinates, gap buckets
<=200 / 200-500 / 500-1002 (:103-123), results in test_results/{name}_tsplib.json."""
from __future__ import annotations

import json
import os
import pickle
import sys
import time

import numpy as np
import torch
import yaml

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.TSP.utils import rollout


class TSPLib_Tester:
    def __init__(self, config, model=None):
        self.config = config
        mp = config['model_params']
        self.device = torch.device('cuda', config['cuda_device_num'])
        if model is None:
            model = TSPModel(**mp)
            if mp['ensemble']:
                model.decoder.add_local_policy(self.device)
            if config['load_checkpoint']:
                model.load_state_dict(torch.load(config['load_checkpoint'], map_location=self.device)['model_state_dict'])
        self.model = model.to(self.device)
        self.tsplib_path = 'TSPLib'
        self.aug_factor = config['params']['aug_factor']

    def test_on_tsplib(self, names=None, max_size=1002):
        files = sorted(f[:-4] for f in os.listdir(self.tsplib_path) if f.endswith('.pkl'))
        if names is not None:
            files = [f for f in files if f in names]
        results, total_time = [], 0.0
        for name in files:
            with open(os.path.join(self.tsplib_path, name + '.pkl'), 'rb') as f:
                instance = pickle.load(f)
            if len(instance[0]) > max_size:
                continue
            rec = {'run_idx': 0}
            t0 = time.time()
            self.test_on_one_ins(name, rec, instance)
            total_time += time.time() - t0
            results.append({'instance': name, 'optimal': instance[1], 'record': [rec]})
            print("Instance Name {}: gap {:.4f}".format(name, rec['gap']))
        os.makedirs('test_results', exist_ok=True)
        with open('test_results/' + self.config['name'] + '_tsplib.json', 'w') as f:
            json.dump(results, f)
        cost = np.array([r['record'][-1]['best_cost'] for r in results])
        opt = np.array([r['optimal'] for r in results])
        scale = np.array([r['record'][-1]['scale'] for r in results])
        gap = (cost - opt) / opt

        def bucket(m):
            return float(100 * gap[m].mean()) if m.any() else float('nan')
        summary = {"total": bucket(scale <= 1002), "<=200": bucket(scale <= 200),
                   "200-500": bucket((scale > 200) & (scale <= 500)), "500-1002": bucket((scale > 500) & (scale <= 1002))}
        print("Total average gap {:.2f}%  Average time {:.2f}s".format(summary["total"], total_time / max(len(results), 1)))
        return results, summary

    def test_on_one_ins(self, name, result_dict, instance):
        raw = np.asarray(instance[0], dtype=np.float64)
        unscaled = torch.tensor(raw, dtype=torch.float)[None]
        pts = (raw - raw.min()) / (raw.max() - raw.min())
        batch = torch.tensor(pts, dtype=torch.float)[None]
        n = batch.shape[1]
        env = TSPEnv(n, self.device)
        env.load_tsplib_problem(batch, unscaled, self.aug_factor)
        reset_state, _, _ = env.reset()
        self.model.eval()
        self.model.requires_grad_(False)
        with torch.no_grad():
            self.model.pre_forward(reset_state)
            _, _, rewards = rollout(self.model, env, 'greedy')
        best = -rewards.reshape(self.aug_factor, 1, n).max(dim=2)[0].max(dim=0)[0].float()
        result_dict['best_cost'] = best.cpu().numpy().tolist()[0]
        result_dict['scale'] = n
        result_dict['gap'] = (result_dict['best_cost'] - instance[1]) / instance[1]


if __name__ == "__main__":
    with open('config.yml', 'r', encoding='utf-8') as fh:
        config = yaml.load(fh.read(), Loader=yaml.FullLoader)
    TSPLib_Tester(config=config).test_on_tsplib()
