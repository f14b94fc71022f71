"""Synthetic CVRP instances and the pickled dataset formats of the reference (gaocrr/ELG
CVRP/generate_data.py).  `uniform` draws exactly the reference's tensors from torch's CPU generator
(so a seeded run sees the same instances, generate_data.py:10-14,84-89); `cluster` / `mixed` follow the
same distributions with vectorised draws."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

# From "VRP with RL" (arXiv 1802.04240), as in the reference (generate_data.py:75-83)
CAPACITIES = {10: 20., 20: 30., 50: 40., 100: 50., 200: 80., 500: 100., 1000: 250.}


def _gauss_clusters(batch, count, centers, std):
    """count points per instance around per-instance centres (batch, n_cluster, 2), sizes as the
    reference splits them: equal parts, remainder in the last cluster."""
    n_c = centers.shape[1]
    part = count // n_c
    sizes = [part] * (n_c - 1) + [count - part * (n_c - 1)]
    chunks = [centers[:, i:i + 1, :] + std * torch.randn(batch, s, 2) for i, s in enumerate(sizes)]
    return torch.cat(chunks, dim=1).clamp_(0.0, 1.0)


def generate_vrp_data(batch_size, problem_size, distribution):
    kind = distribution['data_type']
    if isinstance(kind, (list, np.ndarray)):
        kind = kind[0]
    if kind == 'uniform':
        depot_xy = torch.rand(size=(batch_size, 1, 2))
        node_xy = torch.rand(size=(batch_size, problem_size, 2))
    elif kind == 'cluster':
        lo, hi = distribution['lower'], distribution['upper']
        centers = lo + (hi - lo) * torch.rand(batch_size, distribution['n_cluster'], 2)
        pts = _gauss_clusters(batch_size, problem_size + 1, centers, distribution['std'])
        pick = torch.randint(0, problem_size + 1, (batch_size,))
        depot_xy = pts[torch.arange(batch_size), pick][:, None, :]
        # the other problem_size points in order: index i skips `pick` (a gather; boolean-mask indexing of this
        # tiny tensor cost 40 ms on the host -- four GPU training steps)
        ar = torch.arange(problem_size)[None, :]
        idx = ar + (ar >= pick[:, None]).long()
        node_xy = pts.gather(1, idx[:, :, None].expand(-1, -1, 2))
    elif kind == 'mixed':
        lo, hi = distribution['lower'], distribution['upper']
        depot_xy = torch.rand(size=(batch_size, 1, 2))
        node_xy = torch.rand(batch_size, problem_size, 2)
        centers = lo + (hi - lo) * torch.rand(batch_size, distribution['n_cluster_mix'], 2)
        half = problem_size // 2
        clustered = _gauss_clusters(batch_size, half, centers, distribution['std'])
        where = torch.argsort(torch.rand(batch_size, problem_size), dim=1)[:, :half]
        node_xy.scatter_(1, where[:, :, None].expand(-1, -1, 2), clustered)
    else:
        raise KeyError(kind)
    demand = torch.randint(1, 10, size=(batch_size, problem_size)).float() / CAPACITIES[problem_size]
    return {'loc': node_xy, 'demand': demand, 'depot': depot_xy}


def make_instance(args):
    depot, loc, demand, capacity, *rest = args
    grid = rest[2] if len(rest) > 0 else 1
    return {'loc': torch.tensor(loc, dtype=torch.float) / grid,
            'demand': torch.tensor(demand, dtype=torch.float) / capacity,
            'depot': torch.tensor(depot, dtype=torch.float) / grid}


class VRPDataset(Dataset):
    """pkl list of (depot, loc, demand, capacity[, ...]) tuples, or generated on the fly
    (reference generate_data.py:120-170)."""

    def __init__(self, filename=None, size=100, num_samples=10000, offset=0, distribution=None):
        super().__init__()
        if filename is not None:
            assert os.path.splitext(filename)[1] == '.pkl'
            with open(filename, 'rb') as f:
                data = _Unpickler(f).load()
            if isinstance(data, VRPDataset):
                data = data.data
            self.data = [make_instance(a) if not isinstance(a, dict) else a for a in data[offset:offset + num_samples]]
        else:
            dist = distribution or {'data_type': 'uniform'}
            d = generate_vrp_data(num_samples, size, dist)
            self.data = [{'loc': d['loc'][i], 'demand': d['demand'][i], 'depot': d['depot'][i, 0]}
                         for i in range(num_samples)]
        self.size = len(self.data)

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        return self.data[idx]


class _Unpickler(pickle.Unpickler):
    """vrp{100,200,500}_val.pkl are pickled `__main__.VRPDataset` objects (reference generate_data.py:173-190)."""

    def find_class(self, module, name):
        if name == 'VRPDataset':
            return VRPDataset
        return super().find_class(module, name)


def save_dataset(dataset, filename):
    d = os.path.split(filename)[0]
    if d and not os.path.isdir(d):
        os.makedirs(d)
    if os.path.splitext(filename)[1] != '.pkl':
        filename += '.pkl'
    with open(filename, 'wb') as f:
        pickle.dump(dataset, f, pickle.HIGHEST_PROTOCOL)
