#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REAL reference (/root/reference, Python, CPU).

Runs only in the build container (the reference never travels to the GPU box); the fixtures it
writes are data (inputs + the reference's outputs) and are committed together with this script.

    python tools/make_golden.py            # both trees (each in its own subprocess)
    python tools/make_golden.py cvrp|tsp   # one tree (the two trees define clashing module names)

Weights / problems come from tests/golden_util.py (numpy RandomState), loaded into the reference
modules with load_state_dict, so fixtures store seeds instead of state-dicts."""
import os
import random
import subprocess
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import golden_util as gu  # noqa: E402

OUT = gu.GOLDEN_DIR


def _torch():
    import torch
    torch.set_num_threads(8)
    return torch


def packbits(mask_bool):
    return np.packbits(mask_bool.astype(np.uint8), axis=-1)


def summarize_named(named, full_limit=5000, stride=5):
    """{name: tensor} -> flat dict of arrays: full for small params, strided sample + norm for large."""
    out = {"stride": np.int64(stride)}
    for n, t in named.items():
        a = t.detach().cpu().numpy().astype(np.float32)
        out["norm/" + n] = np.array(np.sqrt((a.astype(np.float64) ** 2).sum()), dtype=np.float64)
        if a.size <= full_limit:
            out["full/" + n] = a
        else:
            out["samp/" + n] = a.reshape(-1)[::stride].copy()
    return out


# =====================================================================================
# CVRP
# =====================================================================================
def gen_cvrp():
    torch = _torch()
    sys.path.insert(0, os.path.join(REF, "CVRP"))
    import CVRPEnv as ref_env_mod
    import CVRPModel as ref_model_mod
    import utils as ref_utils
    import train as ref_train
    from oracle import elg_oracle as orc

    CAP = {10: 20., 20: 30., 50: 40., 100: 50., 200: 80., 500: 100., 1000: 250.}

    def make_model(mp, wseed, gain=1.0):
        model = ref_model_mod.CVRPModel(**mp)
        model.decoder.add_local_policy("cpu")
        w = gu.golden_weights("cvrp", wseed, mp, local=True, gain=gain)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        return model

    def traced_rollout(model, env, eval_type, pf_steps=None, with_grad=False):
        """The reference's rollout loop (utils.rollout) re-driven step by step so that the env /
        decoder internals can be recorded.  Calls only reference methods."""
        rec = dict(load=[], mask=[], finished=[], cur=[], pf=[], pf_t=[], feat_dist=[], feat_theta=[], feat_nd=[])
        env.reset()
        actions, probs = [], []
        state, reward, done = env.pre_step()
        probs_holder = {}
        orig_fwd = model.decoder.forward

        def hooked(*a, **k):
            p = orig_fwd(*a, **k)
            probs_holder["p"] = p.detach().clone()
            return p
        model.decoder.forward = hooked
        t = 0
        ctx = torch.enable_grad() if with_grad else torch.no_grad()
        with ctx:
            while not done:
                cur_dist, cur_theta, xy, norm_demand = env.get_cur_feature()
                want = pf_steps is None or t in pf_steps
                if t >= 2 and want:
                    rec["feat_dist"].append(cur_dist.detach().clone().numpy())
                    rec["feat_theta"].append(cur_theta.detach().clone().numpy())
                    rec["feat_nd"].append(norm_demand.detach().clone().numpy())
                selected, p1 = model.one_step_rollout(state, cur_dist, cur_theta, xy, norm_demand=norm_demand,
                                                      eval_type=eval_type)
                if t >= 2 and want:
                    rec["pf"].append(probs_holder["p"].numpy())
                    rec["pf_t"].append(t)
                state, reward, done = env.step(selected)
                actions.append(selected)
                probs.append(p1)
                rec["load"].append(state.load.detach().clone().numpy())
                rec["mask"].append(packbits(torch.isinf(state.ninf_mask).numpy()))
                rec["finished"].append(state.finished.clone().numpy())
                t += 1
        model.decoder.forward = orig_fwd
        act = torch.stack(actions, 1).transpose(1, 2)
        out = dict(actions=act.numpy().astype(np.int16), reward=reward.detach().numpy(),
                   load=np.stack(rec["load"]), maskbits=np.stack(rec["mask"]), finished=np.stack(rec["finished"]),
                   pf_t=np.array(rec["pf_t"], dtype=np.int32))
        if rec["pf"]:
            out["pf"] = np.stack(rec["pf"])
            fd = np.stack(rec["feat_dist"]); ft = np.stack(rec["feat_theta"]); fn = np.stack(rec["feat_nd"])
            out["feat_dist"] = fd
            out["feat_theta"] = ft
            out["feat_nd"] = np.where(np.isfinite(fn), fn, 0).astype(np.float32)
        if eval_type == "sample":
            out["sel_prob"] = torch.stack(probs, 1).detach().numpy()
        return out

    def rollout_fixture(tag, B, N, M, wseed, pseed, gain, local_size, pf_steps, eval_type="sample", rseed=5):
        mp = dict(gu.CVRP_MODEL_PARAMS)
        mp["local_size"] = [local_size]
        model = make_model(mp, wseed, gain)
        depot, loc, demand = gu.golden_cvrp_problem(pseed, B, N, CAP[N])
        env = ref_env_mod.CVRPEnv(multi_width=M, device="cpu")
        env.load_random_problems(dict(loc=torch.from_numpy(loc), demand=torch.from_numpy(demand),
                                      depot=torch.from_numpy(depot)))
        reset_state, _, _ = env.reset()
        ref_utils.seed_everything(rseed)
        with torch.no_grad():
            model.pre_forward(reset_state)
        enc = model.encoded_nodes.detach().numpy()
        out = traced_rollout(model, env, eval_type, pf_steps)
        out.update(enc=enc, dist=env.dist.numpy(),
                   meta=np.array([B, N, M, wseed, pseed, local_size, rseed], dtype=np.int64), gain=np.float64(gain),
                   capacity=np.float64(CAP[N]))
        # sanity: the reference's own utils.rollout gives the same actions under the same seeds
        ref_utils.seed_everything(rseed)
        env2 = ref_env_mod.CVRPEnv(multi_width=M, device="cpu")
        env2.load_random_problems(dict(loc=torch.from_numpy(loc), demand=torch.from_numpy(demand),
                                       depot=torch.from_numpy(depot)))
        rs2, _, _ = env2.reset()
        with torch.no_grad():
            model.pre_forward(rs2)
            a2, p2, r2 = ref_utils.rollout(model, env2, eval_type)
        assert (a2.numpy() == out["actions"]).all(), "traced rollout diverged from utils.rollout"
        assert np.array_equal(r2.numpy(), out["reward"])
        np.savez_compressed(os.path.join(OUT, f"cvrp_rollout_{tag}.npz"), **out)
        print(tag, "T =", out["actions"].shape[2], {k: v.shape for k, v in out.items() if hasattr(v, "shape")})

    rollout_fixture("n20", B=2, N=20, M=20, wseed=11, pseed=21, gain=1.0, local_size=40, pf_steps=None)
    rollout_fixture("n20k8", B=2, N=20, M=20, wseed=12, pseed=22, gain=2.0, local_size=8, pf_steps=None)
    rollout_fixture("n50", B=2, N=50, M=50, wseed=13, pseed=23, gain=2.0, local_size=40,
                    pf_steps={2, 3, 10, 25, 40, 52, 56, 58, 60, 62, 64, 66})
    rollout_fixture("n100", B=1, N=100, M=100, wseed=14, pseed=24, gain=1.0, local_size=40,
                    pf_steps={2, 30, 70, 110, 125, 130})
    rollout_fixture("greedy_n20", B=4, N=20, M=20, wseed=15, pseed=25, gain=1.0, local_size=40, pf_steps={2, 9},
                    eval_type="greedy")

    # ---------------- one real train() step (train.py:83-125) ----------------
    def train_fixture(tag, B, N, M, wseed, rseed):
        mp = dict(gu.CVRP_MODEL_PARAMS)
        model = make_model(mp, wseed, 1.0)
        w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        cap = {}
        orig_gen, orig_roll = ref_train.generate_vrp_data, ref_train.rollout

        def gen(**kw):
            b = orig_gen(**kw)
            cap["batch"] = {k: v.clone() for k, v in b.items()}
            return b

        def roll(**kw):
            s, p, r = orig_roll(**kw)
            cap["actions"], cap["probs"], cap["rewards"] = s.clone(), p.detach().clone(), r.detach().clone()
            return s, p, r
        ref_train.generate_vrp_data, ref_train.rollout = gen, roll
        ref_utils.seed_everything(rseed)
        dist_cfg = dict(data_type="uniform", n_cluster=3, n_cluster_mix=1, lower=0.2, upper=0.8, std=0.07)
        ref_train.train(model=model, training="joint", T=10 ** 9, start_steps=0, train_steps=0, mixed=False,
                        train_batch_size=B, problem_size=N, distribution=dist_cfg, multiple_width=M, lr=1e-4,
                        device="cpu", logger=None, scale_norm=True, fileLogger=None, dir_path=None, log_step=10 ** 9)
        ref_train.generate_vrp_data, ref_train.rollout = orig_gen, orig_roll
        # loss value restated from the captured tensors exactly as train.py:114-121 does
        rewards, probs = cap["rewards"], cap["probs"]
        adv = rewards - rewards.mean(dim=1)[:, None]
        J = (-adv * probs.log().sum(dim=1) / adv.max(dim=1)[0][:, None]).mean()
        grads = {n: p.grad for n, p in model.named_parameters()}
        delta = {n: (p.detach() - w0[n]) for n, p in model.named_parameters()}
        out = dict(loc=cap["batch"]["loc"].numpy(), demand=cap["batch"]["demand"].numpy(),
                   depot=cap["batch"]["depot"].numpy(), actions=cap["actions"].numpy().astype(np.int16),
                   probs=probs.numpy(), rewards=rewards.numpy(), loss=np.float64(J.item()),
                   meta=np.array([B, N, M, wseed, rseed], dtype=np.int64))
        out.update({"grad/" + k: v for k, v in summarize_named(grads).items() if k != "stride"})
        out.update({"delta/" + k: v for k, v in summarize_named(delta).items() if k != "stride"})
        out["stride"] = summarize_named({})["stride"]
        np.savez_compressed(os.path.join(OUT, f"cvrp_train_{tag}.npz"), **out)
        print("train", tag, "loss", J.item(), "T", cap["actions"].shape)

    train_fixture("n20", B=4, N=20, M=20, wseed=16, rseed=7)

    # ---------------- seeded default initialisation (drop-in: same seed -> same weights) ----------------
    ref_utils.seed_everything(924)
    m0 = ref_model_mod.CVRPModel(**dict(gu.CVRP_MODEL_PARAMS))
    m0.decoder.add_local_policy("cpu")
    starts0 = random.sample(range(0, 100), 100)                      # the draw CVRPModel.py:47 would make next
    init = {}
    for k, v in m0.state_dict().items():
        a = v.numpy().astype(np.float64).reshape(-1)
        init["sum/" + k] = np.float64(a.sum()); init["abs/" + k] = np.float64(np.abs(a).sum())
        init["head/" + k] = v.numpy().reshape(-1)[:4].copy()
    np.savez_compressed(os.path.join(OUT, "cvrp_init_seed924.npz"), starts=np.array(starts0), **init)

    # ---------------- aug8 (utils.py:69-87) ----------------
    x = torch.from_numpy(np.random.RandomState(3).uniform(size=(3, 7, 2)).astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "aug8.npz"), x=x.numpy(), y=ref_utils.augment_xy_data_by_8_fold(x).numpy())

    # ---------------- VRPLIB known answers + one instance end to end ----------------
    names, costs, ref_costs, dims = [], [], [], []
    for sub in ("Vrp-Set-X", "Vrp-Set-XXL"):
        d = os.path.join(REF, "CVRP", "VRPLib", sub)
        for f in sorted(os.listdir(d)):
            if not f.endswith(".vrp"):
                continue
            inst = orc.read_vrp(os.path.join(d, f))
            sol = orc.read_sol(os.path.join(d, f[:-4] + ".sol"))
            env = ref_env_mod.CVRPEnv(multi_width=1, device="cpu")
            env.load_vrplib_problem(inst, aug_factor=1)
            tour = [0]
            for r in sol["routes"]:
                tour += r + [0]
            t = torch.tensor(tour, dtype=torch.long)[None, None, :]
            c = -env.compute_unscaled_reward(solutions=t, rounding=True)[0, 0].item()
            names.append(f[:-4]); costs.append(sol["cost"]); ref_costs.append(c); dims.append(len(inst["demand"]))
    assert costs == ref_costs, "reference reward does not reproduce the .sol costs"
    np.savez_compressed(os.path.join(OUT, "vrplib_known_answers.npz"), names=np.array(names), costs=np.array(costs),
                        ref_costs=np.array(ref_costs), dims=np.array(dims))
    print("vrplib known answers:", len(names), "all equal")

    inst = orc.read_vrp(os.path.join(REF, "CVRP", "VRPLib", "Vrp-Set-X", "X-n101-k25.vrp"))
    mp = dict(gu.CVRP_MODEL_PARAMS)
    model = make_model(mp, 17, 1.0)
    env = ref_env_mod.CVRPEnv(multi_width=100, device="cpu")
    env.load_vrplib_problem(inst, aug_factor=8)
    rs, _, _ = env.reset()
    model.eval(); model.requires_grad_(False)
    ref_utils.seed_everything(9)
    model.pre_forward(rs)
    with torch.no_grad():
        sol, _, rew = ref_utils.rollout(model, env, "greedy")
    aug_reward = rew.reshape(8, 1, 100)
    best = -aug_reward.max(dim=2)[0].max(dim=0)[0].float()
    np.savez_compressed(os.path.join(OUT, "cvrp_vrplib_X-n101-k25.npz"), scaled_xy=env.depot_node_xy.numpy(),
                        unscaled_xy=env.unscaled_depot_node_xy.numpy(), demand=env.depot_node_demand.numpy(),
                        actions=sol.numpy().astype(np.int16), reward=rew.numpy(), best_cost=np.float64(best.item()),
                        wseed=np.int64(17), rseed=np.int64(9), enc=model.encoded_nodes.numpy())
    print("vrplib X-n101-k25 best", best.item(), "T", sol.shape)


# =====================================================================================
# TSP
# =====================================================================================
def gen_tsp():
    torch = _torch()
    sys.modules["wandb"] = types.ModuleType("wandb")          # TSP/train.py:7 imports it unconditionally
    sys.path.insert(0, os.path.join(REF, "TSP"))
    import TSPEnv as ref_env_mod
    import TSPModel as ref_model_mod
    import utils as ref_utils
    import train as ref_train

    def make_model(mp, wseed, gain=1.0):
        model = ref_model_mod.TSPModel(**mp)
        model.decoder.add_local_policy("cpu")
        w = gu.golden_weights("tsp", wseed, mp, local=True, gain=gain)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        return model

    def rollout_fixture(tag, B, N, M, wseed, pseed, gain, local_size, pf_steps, eval_type="sample", rseed=5):
        mp = dict(gu.TSP_MODEL_PARAMS)
        mp["local_size"] = [local_size]
        model = make_model(mp, wseed, gain)
        xy = gu.golden_tsp_problem(pseed, B, N)
        env = ref_env_mod.TSPEnv(multi_width=M, device="cpu")
        env.load_random_problems(torch.from_numpy(xy))
        reset_state, _, _ = env.reset()
        ref_utils.seed_everything(rseed)
        holder = {}
        orig_fwd = model.decoder.forward

        def hooked(*a, **k):
            p = orig_fwd(*a, **k)
            holder["p"] = p.detach().clone()
            return p
        model.decoder.forward = hooked
        pf, pf_t, actions, probs = [], [], [], []
        with torch.no_grad():
            model.pre_forward(reset_state)
            enc = model.encoded_nodes.numpy().copy()
            state, reward, done = env.pre_step()
            t = 0
            while not done:
                cur_dist, cur_theta, xy_ = env.get_local_feature()
                sel, p1 = model.one_step_rollout(state, cur_dist=cur_dist, cur_theta=cur_theta, xy=xy_,
                                                 eval_type=eval_type)
                if t >= 1 and (pf_steps is None or t in pf_steps):
                    pf.append(holder["p"].numpy()); pf_t.append(t)
                state, reward, done = env.step(sel)
                actions.append(sel); probs.append(p1)
                t += 1
        model.decoder.forward = orig_fwd
        act = torch.stack(actions, 1).transpose(1, 2)
        out = dict(actions=act.numpy().astype(np.int16), reward=reward.numpy(), pf=np.stack(pf),
                   pf_t=np.array(pf_t, dtype=np.int32), enc=enc,
                   meta=np.array([B, N, M, wseed, pseed, local_size, rseed], dtype=np.int64), gain=np.float64(gain))
        if eval_type == "sample":
            out["sel_prob"] = torch.stack(probs, 1).numpy()
        np.savez_compressed(os.path.join(OUT, f"tsp_rollout_{tag}.npz"), **out)
        print("tsp", tag, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})

    rollout_fixture("n20", B=2, N=20, M=20, wseed=31, pseed=41, gain=1.0, local_size=30, pf_steps=None)
    rollout_fixture("n50", B=2, N=50, M=50, wseed=32, pseed=42, gain=2.0, local_size=30,
                    pf_steps={1, 2, 10, 19, 20, 21, 30, 45, 48, 49})
    rollout_fixture("greedy_n20", B=4, N=20, M=20, wseed=33, pseed=43, gain=1.0, local_size=30, pf_steps={1, 7},
                    eval_type="greedy")

    def train_fixture(tag, B, N, M, wseed, rseed):
        mp = dict(gu.TSP_MODEL_PARAMS)
        model = make_model(mp, wseed, 1.0)
        w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        cap = {}
        orig_gen, orig_roll = ref_train.generate_tsp_data, ref_train.rollout

        def gen(**kw):
            b = orig_gen(**kw)
            cap["batch"] = b.clone()
            return b

        def roll(**kw):
            s, p, r = orig_roll(**kw)
            cap["actions"], cap["probs"], cap["rewards"] = s.clone(), p.detach().clone(), r.detach().clone()
            return s, p, r
        ref_train.generate_tsp_data, ref_train.rollout = gen, roll
        ref_utils.seed_everything(rseed)
        dist_cfg = dict(data_type="uniform", n_cluster=3, n_cluster_mix=1, lower=0.2, upper=0.8, std=0.07)
        ref_train.train(model=model, training="joint", T=10 ** 9, start_steps=0, train_steps=0, mixed=False,
                        train_batch_size=B, problem_size=N, distribution=dist_cfg, multiple_width=M, lr=1e-4,
                        device="cpu", logger=None, scale_norm=True, fileLogger=None, dir_path=None, log_step=10 ** 9)
        ref_train.generate_tsp_data, ref_train.rollout = orig_gen, orig_roll
        rewards, probs = cap["rewards"], cap["probs"]
        adv = rewards - rewards.mean(dim=1)[:, None]
        J = -adv * probs.log().sum(dim=1)
        nf = adv.max(dim=1)[0][:, None]
        if (nf != 0.).all():
            J = J / nf
        J = J.mean()
        grads = {n: p.grad for n, p in model.named_parameters()}
        delta = {n: (p.detach() - w0[n]) for n, p in model.named_parameters()}
        out = dict(problems=cap["batch"].numpy(), actions=cap["actions"].numpy().astype(np.int16), probs=probs.numpy(),
                   rewards=rewards.numpy(), loss=np.float64(J.item()), meta=np.array([B, N, M, wseed, rseed], dtype=np.int64))
        out.update({"grad/" + k: v for k, v in summarize_named(grads).items() if k != "stride"})
        out.update({"delta/" + k: v for k, v in summarize_named(delta).items() if k != "stride"})
        out["stride"] = summarize_named({})["stride"]
        np.savez_compressed(os.path.join(OUT, f"tsp_train_{tag}.npz"), **out)
        print("tsp train", tag, "loss", J.item())

    train_fixture("n20", B=4, N=20, M=20, wseed=34, rseed=8)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "all":
        for w in ("cvrp", "tsp"):
            subprocess.check_call([sys.executable, "-B", __file__, w], env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    elif which == "cvrp":
        gen_cvrp()
    elif which == "tsp":
        gen_tsp()
    else:
        raise SystemExit("usage: make_golden.py [cvrp|tsp|all]")
