"""`python test.py` of the reference (gaocrr/ELG TSP/test.py): greedy tour length of a pickled TSP test set with and
without 8-fold augmentation.  The evaluation loop lives in elg_amd/evaluate.py."""
from __future__ import annotations

import os
import sys

from torch.utils.data import DataLoader

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd import evaluate as ev
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.TSP.generate_data import TSPDataset
from elg_amd.TSP.utils import rollout


def test(dataloader, model, env, aug_factor):
    """-> (augmented cost, plain cost), averaged over the loader's instances."""
    return ev.evaluate_loader(dataloader, model, env, aug_factor, rollout, lambda batch: batch.shape[0])


if __name__ == "__main__":
    cfg, device = ev.load_run_config()
    run = cfg['params']
    net = ev.build_model(TSPModel, cfg, device)
    loader = DataLoader(TSPDataset(cfg['test_filename'], num_samples=run['test_size']), batch_size=run['test_batch_size'])
    test(loader, net, TSPEnv(multi_width=run['multiple_width'], device=device), run['aug_factor'])
