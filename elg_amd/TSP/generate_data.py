"""Synthetic TSP instances (gaocrr/ELG TSP/generate_data.py): `uniform` is the reference's exact draw
(generate_data.py:10-13); `cluster` / `mixed` follow the same distributions with vectorised draws."""
from __future__ import annotations

import os
import pickle
import sys

import numpy as np
import torch
from torch.utils.data import Dataset

if __package__ in (None, ""):                      # `cd elg_amd/TSP && python generate_data.py`, as the reference is run
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

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


def save_dataset(dataset, filename):
    """reference TSP/generate_data.py:64-72: the (count, N, 2) coordinate tensor, pickled."""
    d = os.path.split(filename)[0]
    if d and not os.path.isdir(d):
        os.makedirs(d)
    if os.path.splitext(filename)[1] != '.pkl':
        filename += '.pkl'
    with open(filename, 'wb') as f:
        pickle.dump(dataset, f, pickle.HIGHEST_PROTOCOL)


def main(argv=None):
    """`python generate_data.py`: the dataset writer of the reference's __main__ block (TSP/generate_data.py:101-126) -- by
    default its run (1000 / 1000 / 100 uniform validation instances of size 100 / 200 / 500 -> data/tsp_{N}_val.pkl, unseeded
    there; --seed makes it reproducible)."""
    import argparse
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--problem-size", type=int, nargs="+", default=[100, 200, 500])
    ap.add_argument("--data-size", type=int, nargs="+", default=[1000, 1000, 100])
    ap.add_argument("--data-type", choices=["uniform", "cluster", "mixed"], default="uniform")
    ap.add_argument("--kind", choices=["val", "test"], default="val",
                    help="val: data/tsp_{N}_val.pkl; test: data/tsp_{type}{N}_test.pkl (the reference's two name patterns)")
    ap.add_argument("--out-dir", default="data")
    a = ap.parse_args(argv)
    if len(a.problem_size) != len(a.data_size):
        ap.error("--problem-size and --data-size need the same number of entries")
    if a.seed is not None:
        torch.manual_seed(a.seed)
        np.random.seed(a.seed)
    dist = {"data_type": a.data_type, "n_cluster": 3, "n_cluster_mix": 1, "lower": 0.2, "upper": 0.8, "std": 0.07}
    written = []
    for n, count in zip(a.problem_size, a.data_size):
        name = f"tsp_{n}_val.pkl" if a.kind == "val" else f"tsp_{a.data_type}{n}_test.pkl"
        path = os.path.join(a.out_dir, name)
        save_dataset(generate_tsp_data(count, n, dist), path)
        written.append(path)
        print(f"{path}: {count} instances of size {n} ({a.data_type})")
    return written


if __name__ == "__main__":
    main()
