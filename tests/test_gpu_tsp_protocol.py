"""GPU tests of the TSP drop-in protocol (elg_amd/TSP/*) against the reference's golden vectors."""
import os
import pickle

import numpy as np
import pytest
import torch

import golden_util as gu
import gpu_common as gc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(mp, wseed, gain=1.0):
    from elg_amd.TSP.TSPModel import TSPModel
    m = TSPModel(**mp)
    m.decoder.add_local_policy(DEV)
    w = gu.golden_weights("tsp", wseed, mp, True, gain)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return m.to(DEV)


def test_tsp_fused_and_stepwise_greedy_match_reference():
    from elg_amd.TSP.TSPEnv import TSPEnv
    from elg_amd.TSP.utils import rollout, check_feasible
    fx = gu.load_golden("tsp_rollout_greedy_n20.npz")
    B, N, M, wseed, pseed, local_size, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.TSP_MODEL_PARAMS)
    model = _model(mp, wseed).eval()
    env = TSPEnv(M, DEV)
    env.load_random_problems(torch.from_numpy(gu.golden_tsp_problem(pseed, B, N)))
    rs, _, _ = env.reset()
    acts = fx["actions"].astype(np.int64)
    model.draw_starts = lambda n, m: [int(x) for x in acts[0, :, 0]]
    with torch.no_grad():
        model.pre_forward(rs)
        np.testing.assert_allclose(model.encoded_nodes.cpu().numpy(), fx["enc"], rtol=2e-4, atol=5e-5)
        a, p, r = rollout(model, env, 'greedy')
        assert p is None and np.array_equal(a.cpu().numpy(), acts)
        np.testing.assert_allclose(r.cpu().numpy(), fx["reward"], rtol=1e-5)
        assert check_feasible(a[0:1])
        # the reference's step loop on this engine
        env.reset()
        state, reward, done = env.pre_step()
        tour = []
        while not done:
            cd, ct, xy = env.get_local_feature()
            sel, _ = model.one_step_rollout(state, cur_dist=cd, cur_theta=ct, xy=xy, eval_type='greedy')
            state, reward, done = env.step(sel)
            tour.append(sel.cpu())
        assert np.array_equal(torch.stack(tour, 2).numpy(), acts)
        np.testing.assert_allclose(reward.cpu().numpy(), fx["reward"], rtol=1e-5)


def test_tsp_train_step_against_reference_train():
    from elg_amd import _lib as L, engine as eng
    from elg_amd.TSP.TSPEnv import TSPEnv
    from elg_amd.TSP.train import pomo_loss
    fx = gu.load_golden("tsp_train_n20.npz")
    B, N, M, wseed, rseed = [int(x) for x in fx["meta"]]
    model = _model(dict(gu.TSP_MODEL_PARAMS), wseed).train()
    env = TSPEnv(M, DEV)
    env.load_random_problems(torch.from_numpy(fx["problems"]))
    rs, _, _ = env.reset()
    model.pre_forward(rs)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    pol = model.decoder.policy
    res = eng.rollout_forward(env.problem, pol, M, acts[0, :, 0], L.MODE_FORCED, forced=acts)
    probs = eng.chosen_probs(env.problem, pol, M, res, N)
    np.testing.assert_allclose(probs.detach().cpu().numpy(), fx["probs"], rtol=5e-4)
    J = pomo_loss(probs, torch.from_numpy(fx["rewards"]).to(DEV), True)
    assert abs(J.item() - float(fx["loss"])) < 2e-5 * max(1.0, abs(float(fx["loss"])))
    J.backward()
    named = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
    stride = int(fx["stride"])
    rms = max(float(fx[k]) / np.sqrt(named[k[len("grad/norm/"):]].numel()) for k in fx.files if k.startswith("grad/norm/"))
    for key in fx.files:
        if not key.startswith("grad/") or key.startswith("grad/norm/"):
            continue
        kind, name = key[5:].split("/", 1)
        g = named[name].numpy().astype(np.float64)
        g = g if kind == "full" else g.reshape(-1)[::stride]
        ref, atol = fx[key], 1e-3 * rms
        err = np.abs(g - ref)
        if name.startswith("decoder."):
            # exact arithmetic from the recorded actions to these gradients (no ReLU in between): every entry within
            # 1e-3 of the tensor's largest entry, no outliers
            assert err.max() <= 1e-3 * np.abs(ref).max() + atol, (name, err.max(), np.abs(ref).max())
            gc.record_parity("tsp_train_step_grad/" + name, float(err.max() / (np.abs(ref).max() + atol)))
        else:
            # encoder: a ReLU pre-activation that sits at ~0 may switch side between CPU and GPU arithmetic, which moves every
            # gradient upstream of it discretely: bound the outliers, require the bulk to agree
            bad = err > 3e-3 * np.abs(ref).max() + atol
            assert bad.mean() < 0.01 and err.max() <= 0.1 * np.abs(ref).max() + atol, name


def test_tsplib_instances_run():
    """load_tsplib_problem (isotropic scaling, x8 aug), greedy POMO = N, rounded length on raw coordinates:
    feasible tours, integer costs, never below the known optimum (random-init weights)."""
    from elg_amd.TSP.test_tsplib import TSPLib_Tester
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "elg_amd", "TSP", "config.yml")))
    cfg["load_checkpoint"] = None
    torch.manual_seed(0)
    tester = TSPLib_Tester(cfg, model=_model(dict(gu.TSP_MODEL_PARAMS), 5))
    for name in ("berlin52", "eil101", "kroA200"):
        inst = pickle.load(open(os.path.join(gu.GOLDEN_DIR, "tsplib", name + ".pkl"), "rb"))
        rec = {}
        tester.test_on_one_ins(name, rec, inst)
        assert rec["scale"] == len(inst[0]) and rec["best_cost"] == round(rec["best_cost"])
        assert rec["best_cost"] >= inst[1] and rec["gap"] < 1.5, rec
