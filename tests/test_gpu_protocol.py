"""GPU tests of the drop-in Python protocol (elg_amd/CVRP/*): step-wise env (bit-exact against the
reference's traces), fused rollout, one real training step against the reference's recorded train() step,
VRPLIB known answers + an instance end to end."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
import gpu_common as gc
from oracle import elg_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(mp, wseed, gain=1.0):
    from elg_amd.CVRP.CVRPModel import CVRPModel
    m = CVRPModel(**mp)
    m.decoder.add_local_policy(DEV)
    w = gu.golden_weights("cvrp", wseed, mp, True, gain)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return m.to(DEV)


def _fixture(tag):
    fx = gu.load_golden(f"cvrp_rollout_{tag}.npz")
    B, N, M, wseed, pseed, local_size, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    mp["local_size"] = [local_size]
    depot, loc, demand = gu.golden_cvrp_problem(pseed, B, N, float(fx["capacity"]))
    batch = dict(loc=torch.from_numpy(loc), demand=torch.from_numpy(demand), depot=torch.from_numpy(depot))
    return fx, mp, batch, B, N, M, wseed, float(fx["gain"])


@pytest.mark.parametrize("tag", ["n20", "n50", "n100"])
def test_env_step_bit_exact(tag):
    """CVRPEnv.step driven with the reference's actions: load (bit pattern), ninf_mask, finished, done flag,
    final reward; get_cur_feature against the recorded features."""
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    fx, mp, batch, B, N, M, wseed, gain = _fixture(tag)
    env = CVRPEnv(M, DEV)
    env.load_random_problems(batch)
    rs, _, done = env.reset()
    np.testing.assert_allclose(rs.dist.cpu().numpy(), fx["dist"], rtol=1e-6, atol=1e-7)
    assert env.get_cur_feature() == (None, None, None, None)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    T = acts.shape[2]
    pf = {int(t): i for i, t in enumerate(fx["pf_t"])}
    state, reward, done = env.pre_step()
    for t in range(T):
        if t in pf:
            cd, ct, rel, nd = env.get_cur_feature()
            i = pf[t]
            np.testing.assert_allclose(cd.cpu().numpy(), fx["feat_dist"][i], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(ct.cpu().numpy(), fx["feat_theta"][i], rtol=1e-5, atol=2e-6)
            ndc = nd.cpu().numpy()
            ok = np.isfinite(ndc)
            np.testing.assert_allclose(ndc[ok], fx["feat_nd"][i][ok], rtol=1e-6)
        state, reward, done = env.step(acts[:, :, t].to(DEV))
        assert np.array_equal(state.load.cpu().numpy().view(np.uint32), fx["load"][t].view(np.uint32)), f"load t={t}"
        got_mask = torch.isinf(state.ninf_mask).cpu().numpy()
        assert np.array_equal(np.packbits(got_mask.astype(np.uint8), axis=-1), fx["maskbits"][t]), f"mask t={t}"
        assert np.array_equal(state.finished.cpu().numpy(), fx["finished"][t]), f"finished t={t}"
        assert done == (t == T - 1)
    np.testing.assert_allclose(reward.cpu().numpy(), fx["reward"], rtol=1e-5)
    assert state.selected_count == T and torch.equal(env.selected_node_list.cpu(), acts)


def test_stepwise_protocol_greedy_matches_reference_loop():
    """The reference's own loop (get_cur_feature -> one_step_rollout -> step) on this engine reproduces the
    reference's greedy tours, and equals the fused rollout()."""
    import random
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.utils import rollout
    fx, mp, batch, B, N, M, wseed, gain = _fixture("greedy_n20")
    model = _model(mp, wseed, gain).eval()
    env = CVRPEnv(M, DEV)
    env.load_random_problems(batch)
    rs, _, _ = env.reset()
    acts = fx["actions"].astype(np.int64)
    with torch.no_grad():
        model.pre_forward(rs)
        np.testing.assert_allclose(model.encoded_nodes.cpu().numpy(), fx["enc"], rtol=2e-4, atol=5e-5)
        orig = model.draw_starts
        model.draw_starts = lambda n, m: [int(x) for x in acts[0, :, 1]]
        state, reward, done = env.pre_step()
        tour = []
        guard = 0
        while not done and guard < 200:
            guard += 1
            cd, ct, xy, nd = env.get_cur_feature()
            sel, p = model.one_step_rollout(state, cd, ct, xy, norm_demand=nd, eval_type='greedy')
            state, reward, done = env.step(sel)
            tour.append(sel.cpu())
        tour = torch.stack(tour, 2).numpy()
        assert np.array_equal(tour, acts)
        np.testing.assert_allclose(reward.cpu().numpy(), fx["reward"], rtol=1e-5)
        a2, p2, r2 = rollout(model, env, 'greedy')
        model.draw_starts = orig
    assert p2 is None and np.array_equal(a2.cpu().numpy(), acts)
    np.testing.assert_allclose(r2.cpu().numpy(), fx["reward"], rtol=1e-5)


def test_train_step_against_reference_train():
    """G6 through the HIP path: the reference's recorded train() step (batch, sampled actions) replayed
    teacher-forced -> loss, EVERY parameter gradient (encoder included) and the Adam update."""
    from elg_amd import _lib as L, engine as eng
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.train import pomo_loss
    from elg_amd.optim import Adam
    fx = gu.load_golden("cvrp_train_n20.npz")
    B, N, M, wseed, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    model = _model(mp, wseed).train()
    env = CVRPEnv(M, DEV)
    env.load_random_problems(dict(loc=torch.from_numpy(fx["loc"]), demand=torch.from_numpy(fx["demand"]),
                                  depot=torch.from_numpy(fx["depot"])))
    rs, _, _ = env.reset()
    model.pre_forward(rs)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    T = acts.shape[2]
    pol = model.decoder.policy
    res = eng.rollout_forward(env.problem, pol, M, acts[0, :, 1], L.MODE_FORCED, forced=acts)
    probs = eng.chosen_probs(env.problem, pol, M, res, T)
    np.testing.assert_allclose(probs.detach().cpu().numpy(), fx["probs"], rtol=5e-4)
    np.testing.assert_allclose(res.reward.cpu().numpy(), fx["rewards"], rtol=1e-5)
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    w0 = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt.zero_grad()
    J = pomo_loss(probs, torch.from_numpy(fx["rewards"]).to(DEV), True)
    assert abs(J.item() - float(fx["loss"])) < 2e-5 * max(1.0, abs(float(fx["loss"])))
    J.backward()
    named = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
    stride = int(fx["stride"])
    rms = max(float(fx[k]) / np.sqrt(named[k[len("grad/norm/"):]].numel()) for k in fx.files if k.startswith("grad/norm/"))
    atol = 1e-3 * rms
    n = 0
    for key in fx.files:
        if not key.startswith("grad/") or key.startswith("grad/norm/"):
            continue
        kind, name = key[5:].split("/", 1)
        g = named[name].numpy().astype(np.float64)
        g = g if kind == "full" else g.reshape(-1)[::stride]
        ref = fx[key]
        err = np.abs(g - ref)
        if name.startswith("decoder."):
            # exact arithmetic from the recorded actions to these gradients (no ReLU in between): every entry within
            # 1e-3 of the tensor's largest entry, no outliers
            assert err.max() <= 1e-3 * np.abs(ref).max() + atol, (name, err.max(), np.abs(ref).max())
            gc.record_parity("train_step_grad/" + name, float(err.max() / (np.abs(ref).max() + atol)))
        else:
            # encoder: an activation within rounding distance of 0 may take the other ReLU branch than in the reference's
            # fp32 evaluation, which moves every gradient upstream of it: allow 1 % of entries beyond the tight bound
            bad = err > 3e-3 * np.abs(ref).max() + atol
            assert bad.mean() < 0.01 and err.max() <= 0.1 * np.abs(ref).max() + atol, name
        n += 1
    assert n == len(named)
    opt.step()
    checked = 0
    for key in fx.files:
        if key.startswith("delta/full/"):
            name = key[len("delta/full/"):]
            gref = fx["grad/full/" + name]
            sig = np.abs(gref) > 1e-3 * np.abs(gref).max()
            if float(fx["grad/norm/" + name]) / np.sqrt(gref.size) < 1e-6 or sig.sum() == 0:
                continue
            got = (dict(model.named_parameters())[name].detach() - w0[name]).cpu().numpy()
            assert (np.abs(got - fx[key])[sig] > 2e-5).mean() < 0.02, name
            checked += 1
    assert checked >= 10


def test_fused_training_step_runs_and_learns():
    """train_step() end to end (sample -> loss -> backward -> Adam): finite, feasible, deterministic per seed."""
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.CVRP.utils import seed_everything
    from elg_amd.optim import Adam
    mp = dict(gu.CVRP_MODEL_PARAMS)
    outs = []
    for rep in range(2):
        seed_everything(11)
        model = _model(mp, 21).train()
        env = CVRPEnv(20, DEV)
        opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
        batch = generate_vrp_data(8, 20, dict(data_type="uniform"))
        J, rew = train_step(model, env, opt, batch, True)
        assert torch.isfinite(J) and torch.isfinite(rew).all()
        outs.append((J.item(), rew.cpu().clone(), model.decoder.Wq_last.weight.detach().cpu().clone()))
    # same seed -> same tours and loss; the update is equal up to the summation order of float atomics
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-7)


def test_vrplib_known_answers_hip():
    """All 104 best-known solutions through the HIP route-length kernel with rounding: exact costs."""
    from elg_amd import engine as eng, vrplib_io
    fx = gu.load_golden("vrplib_known_answers.npz")
    ref = dict(zip([str(n) for n in fx["names"]], fx["costs"]))
    n = 0
    for sub in ("X", "XXL"):
        d = os.path.join(gu.GOLDEN_DIR, "vrplib", sub)
        for f in sorted(os.listdir(d)):
            if not f.endswith(".vrp"):
                continue
            inst = vrplib_io.read_instance(os.path.join(d, f))
            sol = vrplib_io.read_solution(os.path.join(d, f[:-4] + ".sol"))
            tour = [0]
            for r in sol["routes"]:
                tour += r + [0]
            xy = torch.tensor(inst["node_coord"], dtype=torch.float32, device=DEV)[None]
            t = torch.tensor(tour, dtype=torch.long, device=DEV)[None, None]
            c = float(eng.route_length(xy, t, rounding=True)[0, 0])
            assert c == sol["cost"] == ref[f[:-4]], f
            n += 1
    assert n == 104


def test_vrplib_instance_end_to_end():
    """load_vrplib_problem (scaling, x8 augmentation) + greedy rollout + rounded unscaled reward on X-n101-k25
    with the weights of the reference run recorded in the fixture."""
    from elg_amd import vrplib_io
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.utils import rollout
    fx = gu.load_golden("cvrp_vrplib_X-n101-k25.npz")
    inst = vrplib_io.read_instance(os.path.join(gu.GOLDEN_DIR, "vrplib", "X", "X-n101-k25.vrp"))
    mp = dict(gu.CVRP_MODEL_PARAMS)
    model = _model(mp, int(fx["wseed"])).eval()
    env = CVRPEnv(100, DEV)
    env.load_vrplib_problem(inst, aug_factor=8)
    assert np.array_equal(env.depot_node_xy.cpu().numpy(), fx["scaled_xy"])
    assert np.array_equal(env.unscaled_depot_node_xy.cpu().numpy(), fx["unscaled_xy"])
    assert np.array_equal(env.depot_node_demand.cpu().numpy(), fx["demand"])
    rs, _, _ = env.reset()
    ref_acts = torch.from_numpy(fx["actions"].astype(np.int64))
    # the reference's reward of the reference's tours, through the HIP kernel: exact integers
    got = env.compute_unscaled_reward(solutions=ref_acts.to(DEV))
    assert np.array_equal(got.cpu().numpy(), fx["reward"])
    with torch.no_grad():
        model.pre_forward(rs)
        model.draw_starts = lambda n, m: [int(x) for x in ref_acts[0, :, 1]]
        acts, _, rew = rollout(model, env, 'greedy')
    T = min(acts.shape[2], ref_acts.shape[2])
    same = (acts[:, :, :T].cpu() == ref_acts[:, :, :T]).all(dim=2).float().mean().item()
    best = float(-rew.max())
    print("identical trajectories:", same, "best cost", best, "reference", float(fx["best_cost"]))
    assert same > 0.9                                  # free-running greedy: near-ties may fork a few tours
    assert abs(best - float(fx["best_cost"])) <= 0.01 * float(fx["best_cost"])
    for b in range(8):
        orc.check_feasible(acts[b].cpu().numpy(), fx["demand"][b, 1:])


def test_dataset_eval_with_augmentation():
    """test.py path: VRPDataset from the reference's pickle format, x8 augmentation, greedy;
    the augmented best is never worse than the un-augmented one and matches a direct evaluation."""
    from torch.utils.data import DataLoader
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.generate_data import VRPDataset
    from elg_amd.CVRP.test import test
    from elg_amd.CVRP.utils import rollout
    ds = VRPDataset(os.path.join(gu.GOLDEN_DIR, "data", "vrp_uniform100_first8.pkl"), num_samples=8)
    assert len(ds) == 8 and ds[0]["loc"].shape == (100, 2) and float(ds[0]["demand"].max()) <= 9 / 50 + 1e-6
    model = _model(dict(gu.CVRP_MODEL_PARAMS), 5).eval()
    env = CVRPEnv(100, DEV)
    import random
    random.seed(3)
    aug, plain = test(DataLoader(ds, batch_size=4), model, env, 8)
    assert aug <= plain + 1e-6 and 10 < aug < 60
    # un-augmented number equals a direct greedy rollout with the same POMO starts
    random.seed(3)
    tot = 0.0
    for batch in DataLoader(ds, batch_size=4):
        env.load_random_problems(batch, 8)        # same draw order as test(): one start draw per batch
        rs, _, _ = env.reset()
        with torch.no_grad():
            model.pre_forward(rs)
            _, _, r = rollout(model, env, 'greedy')
        tot += float(-r.reshape(8, 4, 100).max(dim=2)[0][0].mean())
    assert abs(tot / 2 - plain) < 1e-4


def test_global_only_then_joint_training_steps():
    """The reference trains the global policy alone for the first T steps and attaches the local policy afterwards
    (train.py:93-96): both regimes, and the switch (new optimizer over the grown parameter list), step without the
    other's tables; TSP likewise."""
    import yaml
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.CVRPModel import CVRPModel
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.TSP.TSPEnv import TSPEnv
    from elg_amd.TSP.TSPModel import TSPModel
    from elg_amd.TSP.train import train_step as tsp_step
    from elg_amd.optim import Adam
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "elg_amd", "CVRP", "config.yml")))
    torch.manual_seed(0)
    m = CVRPModel(**cfg["model_params"]).to(DEV).train()
    env = CVRPEnv(50, DEV)
    opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    w0 = m.decoder.Wq_last.weight.detach().clone()
    dist = dict(cfg["distribution"], data_type="uniform")
    for _ in range(2):
        J, _ = train_step(m, env, opt, generate_vrp_data(16, 50, dist), True)
    assert torch.isfinite(J) and not torch.equal(w0, m.decoder.Wq_last.weight.detach())
    m.decoder.add_local_policy(DEV)
    opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    l0 = m.decoder.local_policies[0].Wq.weight.detach().clone()
    J2, _ = train_step(m, env, opt, generate_vrp_data(16, 50, dist), True)
    assert torch.isfinite(J2) and not torch.equal(l0, m.decoder.local_policies[0].Wq.weight.detach())
    tcfg = yaml.safe_load(open(os.path.join(root, "elg_amd", "TSP", "config.yml")))
    t = TSPModel(**tcfg["model_params"]).to(DEV).train()
    tenv = TSPEnv(50, DEV)
    topt = Adam(t.parameters(), lr=1e-4, weight_decay=1e-6)
    J3, _ = tsp_step(t, tenv, topt, torch.rand(16, 50, 2), True)
    assert torch.isfinite(J3)
