"""Vrp-Set-XXL (reference CVRP/test_vrplib.py:40,73-75, config `vrplib_set: XXL`): instances with 3 001 - 7 001 nodes run through
VRPLib_Tester (x8 augmentation, pomo 1000, greedy) on the N1 > 1024 kernel: feasible tours, integer costs on the raw
coordinates, never below the best-known cost."""
import os
import time

import numpy as np
import pytest
import torch

import golden_util as gu
import gpu_common as gc
from oracle import elg_oracle as orc

pytestmark = pytest.mark.gpu
DEV = gc.DEV
HERE = os.path.dirname(os.path.abspath(__file__))
XXL = os.path.join(gu.GOLDEN_DIR, "vrplib", "XXL")


@pytest.mark.parametrize("name", ["Leuven1", "Antwerp2"])
def test_xxl_instance_through_the_tester(name, tmp_path, monkeypatch):
    import yaml
    from elg_amd import vrplib_io
    from elg_amd.CVRP import test_vrplib as tv
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.utils import rollout
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(HERE), "elg_amd", "CVRP", "config.yml")))
    cfg["load_checkpoint"] = None
    cfg["vrplib_set"] = "XXL"
    model = gc.load_model("cvrp", 21, cfg["model_params"])
    monkeypatch.chdir(tmp_path)
    os.makedirs("VRPLib/Vrp-Set-XXL")
    for ext in (".vrp", ".sol"):
        os.symlink(os.path.join(XXL, name + ext), os.path.join("VRPLib/Vrp-Set-XXL", name + ext))
    tester = tv.VRPLib_Tester(cfg, model=model)
    t0 = time.time()
    results, summary = tester.test_on_vrplib()
    dt = time.time() - t0
    rec = results[0]["record"][-1]
    best, optimal = rec["best_cost"], results[0]["optimal"]
    print(f"{name}: N = {rec['scale']}, best {best:.0f} vs best-known {optimal} (random-init weights), {dt:.1f} s")
    assert best == round(best) and best >= optimal
    # feasibility of the tours of one augmentation (every customer once, capacity respected), and cost = tour length
    inst = vrplib_io.read_instance(os.path.join(XXL, name + ".vrp"))
    env = CVRPEnv(min(rec["scale"], 1000), DEV)
    env.load_vrplib_problem(inst, aug_factor=1)
    rs, _, _ = env.reset()
    with torch.no_grad():
        model.pre_forward(rs)
        sol, _, rew = rollout(model, env, "greedy")
    acts = sol[0].cpu().numpy()
    dem = (np.asarray(inst["demand"], dtype=np.float32) / np.float32(inst["capacity"]))[1:]
    orc.check_feasible(acts[::97], dem)
    coords = np.asarray(inst["node_coord"], dtype=np.float64)
    for m in (0, 501):
        tour = acts[m]
        seg = coords[tour[1:]] - coords[tour[:-1]]
        ref = np.rint(np.sqrt((seg.astype(np.float32) ** 2).sum(1, dtype=np.float32))).sum()
        assert abs(float(-rew[0, m]) - ref) <= 1e-6 * ref + 2
    gc.record_parity(f"xxl/{name}/seconds_aug8_pomo1000", dt)
