"""The callers either side of the path with fixtures of the real reference (tools/make_golden_r02.py): train.validate()'s
test_rollout, VRPLib_Tester / TSPLib_Tester.test_on_one_ins with their summaries -- seeded weights, greedy, so the costs
are deterministic functions of the weights and must match (reference CVRP/train.py:22-80, CVRP/test_vrplib.py:45-145,
TSP/train.py:20-78, TSP/test_tsplib.py:63-162)."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
import gpu_common as gc

pytestmark = pytest.mark.gpu
DEV = gc.DEV
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cvrp_validate_path_matches_reference():
    from torch.utils.data import DataLoader
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.train import test_rollout
    from elg_amd.CVRP.utils import seed_everything
    fx = gu.load_golden("r02_cvrp_validate.npz")
    model = gc.load_model("cvrp", int(fx["wseed"]), dict(gu.CVRP_MODEL_PARAMS))
    for key in ("vrp100_val", "vrp200_val", "vrp_cluster100"):
        rows = [dict(loc=torch.from_numpy(l), demand=torch.from_numpy(d), depot=torch.from_numpy(p))
                for l, d, p in zip(fx[key + "/loc"], fx[key + "/demand"], fx[key + "/depot"])]
        env = CVRPEnv(multi_width=100, device=DEV)
        seed_everything(77)                      # as the generator: the POMO starts are a random subset when N > multi_width
        cost = test_rollout(DataLoader(rows, batch_size=len(rows)), env, model)
        ref = float(fx[key + "/cost"])
        gc.record_parity(f"validate/cvrp/{key}/rel", abs(cost - ref) / ref)
        assert abs(cost - ref) <= 2e-6 * ref, (key, cost, ref)


def test_tsp_validate_path_matches_reference():
    from torch.utils.data import DataLoader
    from elg_amd.TSP.TSPEnv import TSPEnv
    from elg_amd.TSP.train import test_rollout
    from elg_amd.TSP.utils import seed_everything
    fx = gu.load_golden("r02_tsp_validate.npz")
    model = gc.load_model("tsp", int(fx["wseed"]), dict(gu.TSP_MODEL_PARAMS))
    for key in ("tsp_100_val", "tsp_200_val", "tsp_cluster100"):
        rows = [torch.from_numpy(x) for x in fx[key + "/xy"]]
        env = TSPEnv(multi_width=100, device=DEV)
        seed_everything(77)
        cost = test_rollout(DataLoader(rows, batch_size=len(rows)), env, model)
        ref = float(fx[key + "/cost"])
        gc.record_parity(f"validate/tsp/{key}/rel", abs(cost - ref) / ref)
        assert abs(cost - ref) <= 2e-6 * ref, (key, cost, ref)


def test_vrplib_tester_matches_reference(tmp_path, monkeypatch):
    """VRPLib_Tester on three X instances: best_cost, gap and the bucket summary of the reference's tester."""
    import yaml
    from elg_amd.CVRP import test_vrplib as tv
    fx = gu.load_golden("r02_cvrp_vrplib.npz")
    names = [str(n) for n in fx["names"]]
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(HERE), "elg_amd", "CVRP", "config.yml")))
    cfg["load_checkpoint"] = None
    cfg["vrplib_set"] = "X"
    model = gc.load_model("cvrp", int(fx["wseed"]), cfg["model_params"])
    monkeypatch.chdir(tmp_path)
    os.makedirs("VRPLib/Vrp-Set-X")
    src = os.path.join(gu.GOLDEN_DIR, "vrplib", "X")
    for n in names:
        for ext in (".vrp", ".sol"):
            os.symlink(os.path.join(src, n + ext), os.path.join("VRPLib/Vrp-Set-X", n + ext))
    tester = tv.VRPLib_Tester(cfg, model=model)
    results, summary = tester.test_on_vrplib()
    got = {r["instance"]: r["record"][-1] for r in results}
    for i, n in enumerate(names):
        assert got[n]["best_cost"] == float(fx["best_cost"][i]), (n, got[n]["best_cost"], float(fx["best_cost"][i]))
        assert got[n]["scale"] == int(fx["scale"][i])
        assert abs(got[n]["gap"] - float(fx["gap"][i])) < 1e-12
    assert abs(summary["<200"] - float(fx["summary_lt200"])) < 1e-9 and abs(summary["total"] - float(fx["summary_total"])) < 1e-9
    assert os.path.exists("test_results/" + cfg["name"] + "_vrplib.json")


def test_tsplib_tester_matches_reference():
    import pickle
    import yaml
    from elg_amd.TSP import test_tsplib as tt
    fx = gu.load_golden("r02_tsp_tsplib.npz")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(HERE), "elg_amd", "TSP", "config.yml")))
    cfg["load_checkpoint"] = None
    model = gc.load_model("tsp", int(fx["wseed"]), cfg["model_params"])
    tester = tt.TSPLib_Tester(cfg, model=model)
    best = []
    for i, n in enumerate(str(x) for x in fx["names"]):
        inst = pickle.load(open(os.path.join(gu.GOLDEN_DIR, "tsplib", n + ".pkl"), "rb"))
        rec = {}
        tester.test_on_one_ins(n, rec, inst)
        assert rec["best_cost"] == float(fx["best_cost"][i]), (n, rec["best_cost"], float(fx["best_cost"][i]))
        assert rec["scale"] == int(fx["scale"][i])
        best.append(rec["best_cost"])
    total = 100 * ((np.array(best) - fx["optimal"]) / fx["optimal"]).mean()
    assert abs(total - float(fx["total_gap"])) < 1e-9
