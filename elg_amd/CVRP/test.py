"""`python test.py` of the reference (gaocrr/ELG CVRP/test.py:14-86): greedy cost of a pickled CVRP test set with and
without 8-fold augmentation, through the fused HIP rollout.  The evaluation loop lives in elg_amd/evaluate.py."""
from __future__ import annotations

import os
import sys

from torch.utils.data import DataLoader

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from elg_amd import evaluate as ev
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import VRPDataset
from elg_amd.CVRP.utils import rollout


def test(dataloader, model, env, aug_factor):
    """-> (augmented cost, plain cost), averaged over the loader's instances."""
    return ev.evaluate_loader(dataloader, model, env, aug_factor, rollout, lambda batch: batch['loc'].shape[0])


if __name__ == "__main__":
    cfg, device = ev.load_run_config()
    run = cfg['params']
    net = ev.build_model(CVRPModel, cfg, device)
    loader = DataLoader(VRPDataset(cfg['test_filename'], num_samples=run['test_size']), batch_size=run['test_batch_size'])
    test(loader, net, CVRPEnv(multi_width=run['multiple_width'], device=device), run['aug_factor'])
