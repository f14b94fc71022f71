"""Synthetic TSP instances (gaocrr/ELG TSP/generate_data.py): `uniform` is the reference's exact draw
(generate_data.py:10-13); `cluster` / `mixed` follow the same distributions with vectorised draws."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

from elg_amd.CVRP.generate_data import _gauss_clusters


def generate_tsp_data(batch_size, problem_size, distribution):
    kind = distribution['data_type']
    if isinstance(kind, (list, np.ndarray)):
        kind = kind[0]
    if kind == 'uniform':
        return torch.rand(size=(batch_size, problem_size, 2))
    lo, hi = distribution['lower'], distribution['upper']
    if kind == 'cluster':
        centers = lo + (hi - lo) * torch.rand(batch_size, distribution['n_cluster'], 2)
        return _gauss_clusters(batch_size, problem_size, centers, distribution['std'])
    if kind == 'mixed':
        pts = torch.rand(batch_size, problem_size, 2)
        centers = lo + (hi - lo) * torch.rand(batch_size, distribution['n_cluster_mix'], 2)
        half = problem_size // 2
        clustered = _gauss_clusters(batch_size, half, centers, distribution['std'])
        where = torch.argsort(torch.rand(batch_size, problem_size), dim=1)[:, :half]
        return pts.scatter_(1, where[:, :, None].expand(-1, -1, 2), clustered)
    raise KeyError(kind)


class TSPDataset(Dataset):
    """pkl holding a tensor / ndarray / list of (N,2) coordinate arrays, or generated on the fly."""

    def __init__(self, filename=None, size=100, num_samples=10000, offset=0, distribution=None):
        super().__init__()
        if filename is not None:
            assert os.path.splitext(filename)[1] == '.pkl'
            with open(filename, 'rb') as f:
                data = pickle.load(f)
            self.data = [torch.as_tensor(np.asarray(row), dtype=torch.float) for row in data[offset:offset + num_samples]]
        else:
            d = generate_tsp_data(num_samples, size, distribution or {'data_type': 'uniform'})
            self.data = [d[i] for i in range(num_samples)]
        self.size = len(self.data)

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        return self.data[idx]
