"""Known-answer test: the best-known routes of all 104 VRPLIB X / XXL instances, pushed through the
oracle's parser + rounded route length, reproduce the `.sol` Cost exactly -- the only golden
vectors the reference repository itself holds for this path."""
import os

import numpy as np
import torch

import golden_util as gu
from oracle import elg_oracle as orc


def _instances():
    for sub in ("X", "XXL"):
        d = os.path.join(gu.GOLDEN_DIR, "vrplib", sub)
        for f in sorted(os.listdir(d)):
            if f.endswith(".vrp"):
                yield os.path.join(d, f), os.path.join(d, f[:-4] + ".sol")


def sol_tour(sol):
    tour = [0]
    for r in sol["routes"]:
        tour += r + [0]
    return tour


def test_known_answers_oracle():
    fx = gu.load_golden("vrplib_known_answers.npz")
    ref = dict(zip([str(n) for n in fx["names"]], fx["ref_costs"]))
    n = 0
    for vrp, solp in _instances():
        inst, sol = orc.read_vrp(vrp), orc.read_sol(solp)
        xy = torch.tensor(inst["node_coord"], dtype=torch.float32)[None]
        t = torch.tensor(sol_tour(sol), dtype=torch.long)[None, None]
        c = float(orc.route_length(xy, t, rounding=True)[0, 0])
        name = os.path.basename(vrp)[:-4]
        assert c == sol["cost"] == ref[name], name
        # every customer exactly once, capacity respected
        flat = [x for r in sol["routes"] for x in r]
        assert sorted(flat) == list(range(1, len(inst["demand"])))
        for r in sol["routes"]:
            assert inst["demand"][r].sum() <= inst["capacity"]
        n += 1
    assert n == 104
