"""Host-side logic on CPU: weight folds against the oracle's unfolded math, state-dict / seeded-init
compatibility with the reference, data generation, VRPLIB reader."""
import math
import os
import random

import numpy as np
import torch

import golden_util as gu
from oracle import elg_oracle as orc
from elg_amd import _lib as L
from elg_amd import engine as eng
import gpu_common as gc


def _weights(problem, seed, mp):
    return {k: torch.from_numpy(v) for k, v in gu.golden_weights(problem, seed, mp, True, 1.0).items()}


def _sub(P, prefix):
    return {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}


def _folded_local(loc, feats, smask, nf):
    """The kernel's folded evaluation of the local policy (csrc/elg_rollout.h::local_policy), in torch."""
    Ls = feats.shape[0]
    la = loc[L.LOC_LA:L.LOC_LA + 12].view(4, 3)[:, :nf]
    lt = loc[L.LOC_LT:L.LOC_LT + 256].view(64, 4)[:Ls]
    lAv = loc[L.LOC_LAV:L.LOC_LAV + 96].view(32, 3)[:, :nf]
    lcv = loc[L.LOC_LCV:L.LOC_LCV + 2048].view(64, 32)[:Ls]
    lWc = loc[L.LOC_LWC:L.LOC_LWC + 1024].view(32, 32)
    lbc = loc[L.LOC_LBC:L.LOC_LBC + 32]
    lWe = loc[L.LOC_LWE:L.LOC_LWE + 96].view(32, 3)[:, :nf]
    lpe = loc[L.LOC_LPE:L.LOC_LPE + 2048].view(64, 32)[:Ls]
    sc = feats @ la.T + lt
    sc = sc.masked_fill(smask[:, None], float("-inf"))
    al = torch.softmax(sc, dim=0)                                  # (L,4)
    F = al.T @ feats                                               # (4,nf)
    P = (al.repeat_interleave(8, dim=1) * lcv).sum(0)              # (32)
    op = P + (lAv * F.repeat_interleave(8, dim=0)).sum(1)
    g = lWc @ op + lbc
    return lpe @ g + feats @ (lWe.T @ g)


def test_fold_local_tables_matches_unfolded_math():
    for problem, mp, nf in (("cvrp", gu.CVRP_MODEL_PARAMS, 3), ("tsp", gu.TSP_MODEL_PARAMS, 2)):
        P = _weights(problem, 3, mp)
        cfg = orc.ModelCfg.from_model_params(mp, problem)
        pre = "decoder.local_policies.0." if problem == "cvrp" else "decoder.local_policy_0."
        Ls = cfg.local_size + (1 if problem == "cvrp" else 0)
        loc = gc.fold_local_tables(_sub(P, pre), nf, Ls)
        assert loc.numel() == L.LOC_SIZE
        torch.manual_seed(1)
        feats = torch.rand(1, 1, Ls, nf)
        smask = torch.zeros(1, 1, Ls, dtype=torch.bool)
        smask[0, 0, Ls - 5:] = True
        feats[0, 0, Ls - 5:] = 0
        ref = orc._local_policy(P, cfg, pre, feats, smask)[0, 0]
        got = _folded_local(loc, feats[0, 0], smask[0, 0], nf)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=2e-6)


def test_fold_decoder_tables_matches_decoder_math():
    mp = gu.CVRP_MODEL_PARAMS
    P = _weights("cvrp", 4, mp)
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    torch.manual_seed(2)
    enc = torch.randn(2, 11, 128)
    t = gc.fold_decoder_tables(_sub(P, "decoder."), enc, L.PROBLEM_CVRP)
    kh, vh = orc.set_kv(P, cfg, enc)
    np.testing.assert_allclose(t["K"].view(2, 11, 8, 16).transpose(1, 2).numpy(), kh.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(t["V"].view(2, 11, 8, 16).transpose(1, 2).numpy(), vh.numpy(), rtol=1e-5, atol=1e-6)
    o = torch.randn(2, 5, 128)
    g = o @ P["decoder.multi_head_combine.weight"].T + P["decoder.multi_head_combine.bias"]
    s_ref = g @ enc.transpose(1, 2) / math.sqrt(128)
    s_got = o @ t["PK"].transpose(1, 2) + t["pb"][:, None, :]
    np.testing.assert_allclose(s_got.numpy(), s_ref.numpy(), rtol=2e-4, atol=2e-5)
    cur = torch.tensor([[0, 3, 7], [1, 1, 10]])
    load = torch.rand(2, 3)
    h = enc[torch.arange(2)[:, None], cur]
    q_ref = torch.cat([h, load[:, :, None]], 2) @ P["decoder.Wq_last.weight"].T
    q_got = t["Q1"][torch.arange(2)[:, None], cur] + load[:, :, None] * t["wl"]
    np.testing.assert_allclose(q_got.numpy(), q_ref.numpy(), rtol=2e-4, atol=2e-5)


def test_state_dict_layout_and_seeded_init_match_reference():
    """Same seed -> bit-identical default weights and the same POMO start draw as the reference
    (fixture generated from the real reference by tools/make_golden.py)."""
    from elg_amd.CVRP.CVRPModel import CVRPModel
    from elg_amd.CVRP.utils import seed_everything
    fx = gu.load_golden("cvrp_init_seed924.npz")
    seed_everything(924)
    m = CVRPModel(**dict(gu.CVRP_MODEL_PARAMS))
    m.decoder.add_local_policy("cpu")
    starts = random.sample(range(0, 100), 100)
    sd = m.state_dict()
    assert list(sd.keys()) == list(gu.model_param_shapes("cvrp", gu.CVRP_MODEL_PARAMS).keys())
    for k, v in sd.items():
        a = v.numpy().astype(np.float64).reshape(-1)
        assert a.sum() == float(fx["sum/" + k]) and np.abs(a).sum() == float(fx["abs/" + k]), k
        assert np.array_equal(v.numpy().reshape(-1)[:4], fx["head/" + k]), k
    assert starts == list(fx["starts"])
    assert sum(p.numel() for p in m.parameters()) == 1258304


def test_generate_data_and_vrplib_reader():
    from elg_amd.CVRP.generate_data import generate_vrp_data, CAPACITIES
    from elg_amd import vrplib_io
    dist = dict(data_type="uniform", n_cluster=3, n_cluster_mix=1, lower=0.2, upper=0.8, std=0.07)
    torch.manual_seed(5)
    d = generate_vrp_data(4, 100, dist)
    torch.manual_seed(5)                     # the reference's draw order: depot, nodes, demand
    dep, loc = torch.rand(4, 1, 2), torch.rand(4, 100, 2)
    dem = torch.randint(1, 10, (4, 100)).float() / 50.0
    assert torch.equal(d["depot"], dep) and torch.equal(d["loc"], loc) and torch.equal(d["demand"], dem)
    for kind in ("cluster", "mixed"):
        x = generate_vrp_data(3, 50, dict(dist, data_type=kind))
        assert x["loc"].shape == (3, 50, 2) and x["depot"].shape == (3, 1, 2)
        assert (x["loc"] >= 0).all() and (x["loc"] <= 1).all()
        assert torch.all(x["demand"] * CAPACITIES[50] == torch.round(x["demand"] * CAPACITIES[50]))
    p = os.path.join(gu.GOLDEN_DIR, "vrplib", "X", "X-n101-k25")
    a, b = vrplib_io.read_instance(p + ".vrp"), orc.read_vrp(p + ".vrp")
    assert np.array_equal(a["node_coord"], b["node_coord"]) and np.array_equal(a["demand"], b["demand"])
    assert a["capacity"] == b["capacity"] == 206 and list(a["depot"]) == [0]
    s = vrplib_io.read_solution(p + ".sol")
    assert s["cost"] == 27591 and len(s["routes"]) == 26


def test_best_costs_of_the_test_entry_points():
    """elg_amd.evaluate.best_costs: best over POMO, then best over the augmentations (reference test.py:30-41)."""
    import torch
    from elg_amd import evaluate as ev
    torch.manual_seed(0)
    rewards = -torch.rand(8 * 5, 7) * 10
    plain, aug = ev.best_costs(rewards, 8, 5)
    per_aug = rewards.reshape(8, 5, 7).max(dim=2)[0]
    assert torch.equal(plain, -per_aug[0]) and torch.equal(aug, -per_aug.max(dim=0)[0])
    assert (aug <= plain).all()


def test_entry_points_import_from_their_own_directory():
    """`cd elg_amd/CVRP && python train.py` is how the reference is run: the scripts must find the package themselves."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sub, mods in (("CVRP", ("train", "test", "test_vrplib")), ("TSP", ("train", "test", "test_tsplib"))):
        for m in mods:
            env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
            r = subprocess.run([sys.executable, "-c", f"import {m}"], cwd=os.path.join(root, "elg_amd", sub), env=env,
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, (sub, m, r.stderr[-1500:])


def test_entry_points_have_no_undefined_globals():
    """Every global name a function of the entry-point modules loads exists in the module (or builtins): the config-only
    branches (`training: only_local`, checkpoints with local policies ...) are not all exercised on the CPU."""
    import builtins
    import dis
    import importlib
    import types
    missing = []
    for name in ("elg_amd.CVRP.train", "elg_amd.CVRP.test", "elg_amd.CVRP.test_vrplib", "elg_amd.CVRP.utils",
                 "elg_amd.CVRP.CVRPModel", "elg_amd.CVRP.models", "elg_amd.CVRP.CVRPEnv", "elg_amd.TSP.train", "elg_amd.TSP.test",
                 "elg_amd.TSP.test_tsplib", "elg_amd.TSP.utils", "elg_amd.TSP.TSPModel", "elg_amd.TSP.models", "elg_amd.TSP.TSPEnv",
                 "elg_amd.engine", "elg_amd.encoder", "elg_amd.parallel", "elg_amd.optim"):
        mod = importlib.import_module(name)

        with open(mod.__file__) as f:
            top = compile(f.read(), mod.__file__, "exec")
        stored = {i.argval for i in dis.get_instructions(top) if i.opname in ("STORE_NAME", "STORE_GLOBAL", "IMPORT_NAME")}

        def walk(code):
            for ins in dis.get_instructions(code):
                if (ins.opname == "LOAD_GLOBAL" or (ins.opname == "LOAD_NAME" and code is top)) \
                        and not hasattr(mod, ins.argval) and not hasattr(builtins, ins.argval) and ins.argval not in stored:
                    missing.append((name, code.co_name, ins.argval))
            for c in code.co_consts:
                if isinstance(c, types.CodeType):
                    walk(c)
        walk(top)
    assert not missing, missing


def test_train_batch_must_divide_over_the_ranks(monkeypatch):
    from elg_amd import parallel
    from elg_amd.CVRP import train as ctrain
    monkeypatch.setattr(parallel, "world_info", lambda: (0, 3, 0))

    class _Env:
        def __init__(self, **kw):
            pass
    monkeypatch.setattr(ctrain, "CVRPEnv", _Env)
    import pytest
    with pytest.raises(ValueError, match="not divisible"):
        ctrain.train(model=None, training="only_global", T=10, start_steps=0, train_steps=1, mixed=False, train_batch_size=64,
                     problem_size=20, distribution={}, multiple_width=20, lr=1e-4, device="cuda:0", logger=None, scale_norm=True,
                     fileLogger=None, dir_path=".", log_step=10)


def _stats_of(xy):
    xy = np.asarray(xy, dtype=np.float64)
    c = xy.mean(1, keepdims=True)
    d = np.sqrt(((xy[:, :, None, :] - xy[:, None, :, :]) ** 2).sum(-1))
    d2 = d + np.eye(xy.shape[1])[None] * 1e9
    return dict(mean_x=xy[:, :, 0].mean(1), mean_y=xy[:, :, 1].mean(1), std_x=xy[:, :, 0].std(1), std_y=xy[:, :, 1].std(1),
                spread=np.sqrt(((xy - c) ** 2).sum(-1)).mean(1), nn=d2.min(-1).mean(1), lo=xy.min((1, 2)), hi=xy.max((1, 2)))


def test_cluster_and_mixed_generators_follow_the_reference_distributions():
    """The vectorised `cluster` / `mixed` generators against per-instance statistics of the reference's loops
    (generate_data.py:16-72, 2000 instances, tools/make_golden_r02.py): two-sample Kolmogorov-Smirnov on every statistic
    (centroid, per-axis spread, mean distance to the centroid, mean nearest-neighbour distance, extremes, demand)."""
    from scipy.stats import ks_2samp
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.TSP.generate_data import generate_tsp_data
    dist_cfg = dict(n_cluster=3, n_cluster_mix=1, lower=0.2, upper=0.8, std=0.07)
    fx = gu.load_golden("r02_generators_cvrp.npz")
    torch.manual_seed(99)
    worst = 1.0
    for kind in ("cluster", "mixed"):
        d = generate_vrp_data(2000, 100, dict(dist_cfg, data_type=kind))
        st = _stats_of(d["loc"].numpy())
        st["demand_mean"] = d["demand"].numpy().mean(1)
        for k, v in st.items():
            p = ks_2samp(v, fx[f"cvrp_{kind}/{k}"]).pvalue
            worst = min(worst, p)
            assert p > 1e-3, (kind, k, p)
        assert d["depot"].shape == (2000, 1, 2) and float(d["depot"].min()) >= 0 and float(d["depot"].max()) <= 1
    fx = gu.load_golden("r02_generators_tsp.npz")
    d = generate_tsp_data(2000, 100, dict(dist_cfg, data_type="cluster"))
    for k, v in _stats_of(d.numpy()).items():
        p = ks_2samp(v, fx[f"tsp_cluster/{k}"]).pvalue
        assert p > 1e-3, ("tsp", k, p)
    print("smallest KS p-value", worst)


def test_unsupported_model_shapes_raise_value_errors_naming_the_supported_set():
    """The reference takes any embedding / head / local sizes (models.py:8-36,277-294); the HIP kernels are built for its
    config.yml defaults.  Anything else is refused up front with a ValueError that says what is supported."""
    import pytest
    from elg_amd.CVRP.CVRPModel import CVRPModel
    from elg_amd.TSP.TSPModel import TSPModel
    with pytest.raises(ValueError, match=r"\(128, 8, 16\)"):
        CVRPModel(**dict(gu.CVRP_MODEL_PARAMS, embedding_dim=64, head_num=4))
    with pytest.raises(ValueError, match=r"\(128, 8, 16\)"):
        TSPModel(**dict(gu.TSP_MODEL_PARAMS, qkv_dim=32))
    m = CVRPModel(**dict(gu.CVRP_MODEL_PARAMS, local_att_hidden_dim=64))
    with pytest.raises(ValueError, match=r"\(32, 4, 8\)"):
        m.decoder.add_local_policy("cpu")
    m = CVRPModel(**dict(gu.CVRP_MODEL_PARAMS, local_size=[60]))
    m.decoder.add_local_policy("cpu")                                # up to 63: the one-wavefront kernels (round 6)
    m = CVRPModel(**dict(gu.CVRP_MODEL_PARAMS, local_size=[64]))
    with pytest.raises(ValueError, match=r"1 \.\. 63"):
        m.decoder.add_local_policy("cpu")
    m = CVRPModel(**dict(gu.CVRP_MODEL_PARAMS, ensemble_size=5, local_size=[10] * 5))
    with pytest.raises(ValueError, match=r"1 \.\. 4"):
        m.decoder.add_local_policy("cpu")


def test_dataset_writer_entry_points(tmp_path):
    """`python generate_data.py` of both trees (reference CVRP/generate_data.py:173-197, TSP/generate_data.py:101-126): files
    under the reference's names, readable by the dataset classes, seeded runs reproducible, both CVRP formats equivalent."""
    from elg_amd.CVRP import generate_data as gv
    from elg_amd.TSP import generate_data as gt
    a = gv.main(["--problem-size", "20", "50", "--data-size", "6", "4", "--out-dir", str(tmp_path / "a")])
    assert [os.path.basename(p) for p in a] == ["vrp20_val.pkl", "vrp50_val.pkl"]
    b = gv.main(["--problem-size", "20", "50", "--data-size", "6", "4", "--out-dir", str(tmp_path / "b"), "--format", "tuples"])
    for pa, pb, n, cnt in zip(a, b, (20, 50), (6, 4)):
        da, db = gv.VRPDataset(pa, num_samples=cnt), gv.VRPDataset(pb, num_samples=cnt)
        assert len(da) == len(db) == cnt and da[0]['loc'].shape == (n, 2)
        for x, y in zip(da.data, db.data):                      # same seed -> same instances; demand = integer / capacity in both
            assert torch.equal(x['loc'], y['loc']) and torch.allclose(x['demand'], y['demand'], atol=1e-7)
    c = gv.main(["--problem-size", "20", "--data-size", "3", "--data-type", "cluster", "--kind", "test", "--out-dir", str(tmp_path)])
    assert os.path.basename(c[0]) == "vrp_cluster20_test.pkl"
    t1 = gt.main(["--problem-size", "30", "--data-size", "5", "--seed", "7", "--out-dir", str(tmp_path / "t1")])
    t2 = gt.main(["--problem-size", "30", "--data-size", "5", "--seed", "7", "--out-dir", str(tmp_path / "t2")])
    assert os.path.basename(t1[0]) == "tsp_30_val.pkl"
    d1, d2 = gt.TSPDataset(t1[0], num_samples=5), gt.TSPDataset(t2[0], num_samples=5)
    assert len(d1) == 5 and d1[0].shape == (30, 2) and all(torch.equal(x, y) for x, y in zip(d1.data, d2.data))
