"""Config-reachable model variants through the product path, against fixtures of the real reference
(tools/make_golden_r02b.py): `training: only_local` = CVRPModel_local (reference CVRPModel.py:78-131) and
model_params['euclidean'] = True (models.py:95-125, TSP/models.py:67-75).  Logit tolerance as in test_gpu_logits.py."""
import numpy as np
import pytest
import torch

import golden_util as gu
import gpu_common as gc
from oracle import elg_oracle as orc
from elg_amd import _lib as L
from elg_amd import engine as eng
from test_oracle_golden import LOGIT_RTOL, local_only_setup, logit_errors

pytestmark = pytest.mark.gpu
DEV = gc.DEV


def _local_model(mp, P):
    from elg_amd.CVRP.CVRPModel import CVRPModel_local
    model = CVRPModel_local(**mp)
    model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)     # the reference's checkpoint keys
    return model.to(DEV)


def _env(mp, xy, dem, M):
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    env = CVRPEnv(multi_width=M, device=DEV)
    env.load_random_problems(dict(depot=xy[:, :1].clone(), loc=xy[:, 1:].clone(), demand=dem[:, 1:].clone()))
    return env


@pytest.mark.parametrize("tag,variant", [("greedy", 0), ("sample", 0), ("greedy", 1)])
def test_local_only_model_logits(tag, variant):
    """The reference's CVRPModel_local tours teacher-forced through the engine: clipped + masked logits of every decode
    step (1e-4 of the clip), the mask exactly, rewards; sampled tours: chosen probabilities."""
    fx, mp, cfg, P, xy, dem, B, N, M = local_only_setup()
    model = _local_model(mp, P).eval()
    env = _env(mp, xy, dem, M)
    rs, _, _ = env.reset()
    acts = torch.from_numpy(fx[f"{tag}_actions"].astype(np.int64))
    T = acts.shape[2]
    with torch.no_grad():
        model.pre_forward(rs)
        res = eng.rollout_forward(env.problem, model.policy, M, acts[0, :, 1], L.MODE_FORCED, forced=acts, dump_T=T,
                                  dump="logits", variant=variant)
    got = res.full_probs.cpu().numpy()                               # (B, M, T, N1)
    ref = fx[f"{tag}_logits"]                                        # (T-2, B, M, N1)
    live = (np.arange(T)[None, None, :] < res.tlen.cpu().numpy()[:, :, None])
    worst = 0.0
    for i in range(ref.shape[0]):
        t = i + 2
        rows = live[:, :, t]
        g, r = got[:, :, t, :][rows], ref[i][rows]
        assert np.array_equal(np.isfinite(g), np.isfinite(r)), t
        worst = max(worst, logit_errors(g, r, np.isfinite(r), cfg.logit_clipping))
    assert worst <= LOGIT_RTOL, worst
    np.testing.assert_allclose(res.reward.cpu().numpy(), fx[f"{tag}_reward"], rtol=1e-5)
    if tag == "sample":
        p = res.probs[:, :T].cpu().numpy()
        lv = np.transpose(live, (0, 2, 1))
        np.testing.assert_allclose(p[lv], fx["sample_probs"][lv], rtol=5e-4, atol=1e-9)
    gc.record_parity(f"local_only_{tag}_v{variant}_logits_over_clip", worst)


def test_local_only_model_free_running_and_protocol():
    """Greedy free-running rollout (utils.rollout, one launch) and the reference's step-wise loop (one_step_rollout) both
    reproduce the reference's own greedy tours of CVRPModel_local."""
    import random
    from elg_amd.CVRP.utils import rollout
    fx, mp, cfg, P, xy, dem, B, N, M = local_only_setup()
    model = _local_model(mp, P).eval()
    env = _env(mp, xy, dem, M)
    acts = fx["greedy_actions"].astype(np.int64)
    starts = [int(a) for a in acts[0, :, 1]]
    model.draw_starts = staticmethod(lambda n, m: starts)            # the reference drew these with Python's random
    with torch.no_grad():
        rs, _, _ = env.reset()
        model.pre_forward(rs)
        a, p, r = rollout(model, env, 'greedy')
    assert np.array_equal(a.cpu().numpy(), acts)
    np.testing.assert_allclose(r.cpu().numpy(), fx["greedy_reward"], rtol=1e-5)


def test_local_only_training_gradients():
    """`training: only_local`: sampled rollout -> POMO loss -> backward; gradients of every local-policy parameter against
    the oracle's autograd on the same tours (the decoder tables are zeros and carry no gradient)."""
    from elg_amd.CVRP.utils import rollout
    from elg_amd.CVRP.train import pomo_loss
    fx, mp, cfg, P, xy, dem, B, N, M = local_only_setup()
    model = _local_model(mp, P).train()
    env = _env(mp, xy, dem, M)
    rs, _, _ = env.reset()
    model.pre_forward(rs)
    torch.manual_seed(3)
    acts, probs, rew = rollout(model, env, 'sample')
    assert probs.requires_grad
    rew_n = rew + 0.3 * torch.randn(B, M, device=rew.device)        # keep the advantage away from rounding noise
    J = pomo_loss(probs, rew_n, True)
    J.backward()
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = orc.rollout_cvrp(Po, cfg, xy, dem, M, starts=acts[0, :, 1].cpu(), forced=acts.cpu(), local_only=True)
    To = out["probs"].shape[1]
    np.testing.assert_allclose(probs.detach().cpu().numpy()[:, :To], out["probs"].detach().numpy(), rtol=5e-4, atol=1e-9)
    Jo = orc.pomo_loss(out["probs"], rew_n.cpu(), True)
    Jo.backward()
    assert abs(float(J.detach()) - float(Jo.detach())) <= 2e-4 * max(1.0, abs(float(Jo.detach())))
    got = dict(model.named_parameters())
    for k, v in Po.items():
        g, r = got[k].grad.cpu(), v.grad
        err = float((g - r).abs().max())
        assert err <= 2e-3 * float(r.abs().max()) + 1e-6, (k, err, float(r.abs().max()))


@pytest.mark.parametrize("problem,variant", [("cvrp", 0), ("cvrp", 1), ("tsp", 0), ("tsp", 1)])
def test_euclidean_local_features(problem, variant):
    """model_params['euclidean'] = True: local-policy output is not observable on its own through the ABI, the score before
    the clip and the logits are -- both against the reference's, cooperative and per-wavefront kernels."""
    lg = gu.load_golden(f"r02_{problem}_euclidean.npz")
    if problem == "cvrp":
        fx, cfg, P, xy, dem, B, N, M = gc.cvrp_fixture("n20")
        kind, t_first = L.PROBLEM_CVRP, 1
    else:
        fx, cfg, P, xy, B, N, M = gc.tsp_fixture("n20")
        dem, kind, t_first = None, L.PROBLEM_TSP, 0
    cfg.euclidean = True
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    T = acts.shape[2]
    enc = orc.encoder_forward(P, cfg, xy, dem)
    prob = gc.make_problem(xy, dem, kind)
    pol = gc.make_policy(P, cfg, enc.to(DEV), kind)
    assert pol.euclidean
    worst = {}
    for dump, key, scale in (("scores", "pre_clip", None), ("logits", "logits", cfg.logit_clipping)):
        res = eng.rollout_forward(prob, pol, M, acts[0, :, t_first], L.MODE_FORCED, forced=acts, dump_T=T, dump=dump,
                                  variant=variant)
        got = res.full_probs.cpu().numpy()
        for i, t in enumerate(lg["steps"]):
            ref = lg[key][i]
            open_ = np.isfinite(lg["logits"][i])
            e = logit_errors(got[:, :, int(t), :], ref, open_, scale)
            worst[key] = max(worst.get(key, 0.0), e)
    assert max(worst.values()) <= LOGIT_RTOL, worst
    gc.record_parity(f"euclidean_{problem}_v{variant}", max(worst.values()))


# ---------------------------------------------------------------------------------------------------------------------
# local_size above 47 (round 6): the reference takes any K (models.py:8-36); here the one-wavefront kernels (one slot per lane,
# up to 64) run the rollout and the replay backward trains -- fixtures of the real reference at local_size 50 and 63
# ---------------------------------------------------------------------------------------------------------------------
def _wide_model(mp, P):
    from elg_amd.CVRP.CVRPModel import CVRPModel
    model = CVRPModel(**mp)
    model.decoder.add_local_policy(DEV)
    model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    return model.to(DEV)


@pytest.mark.parametrize("variant", [0, 1, 2], ids=["by_shape", "wave_per_trajectory", "xl"])
@pytest.mark.parametrize("tag", ["k50", "k63"])
def test_local_size_above_47_logits_and_tours(tag, variant):
    from test_oracle_golden import wide_slots_setup
    fx, mp, cfg, P, xy, dem, B, N, M = wide_slots_setup(tag)
    model = _wide_model(mp, P).eval()
    env = _env(mp, xy, dem, M)
    acts = torch.from_numpy(fx[f"{tag}_greedy_actions"].astype(np.int64))
    T = acts.shape[2]
    with torch.no_grad():
        rs, _, _ = env.reset()
        model.pre_forward(rs)
        pol = model.decoder.policy
        assert pol.wide_slots and pol.K == int(tag[1:])
        worst = {}
        for dump, key, scale in (("scores", "pre_clip", None), ("logits", "logits", cfg.logit_clipping)):
            res = eng.rollout_forward(env.problem, pol, M, acts[0, :, 1], L.MODE_FORCED, forced=acts, dump_T=T, dump=dump, variant=variant)
            assert res.kernel_id == (L.KERNEL_XL if variant == 2 else L.KERNEL_WAVE)       # never a 48-slot matrix kernel
            got = res.full_probs.cpu().numpy()
            tl = res.tlen.cpu().numpy()
            for i, t in enumerate(fx[f"{tag}_steps"]):
                live = (int(t) < tl)[:, :, None]                     # (a finished trajectory decodes nothing: its dump rows stay zero)
                ref = fx[f"{tag}_{key}"][i]
                open_ = np.isfinite(fx[f"{tag}_logits"][i]) & live
                assert open_.any()
                if key == "logits":
                    assert np.array_equal(np.isfinite(got[:, :, int(t), :]) & live, open_), int(t)
                worst[key] = max(worst.get(key, 0.0), logit_errors(got[:, :, int(t), :], ref, open_, scale))
        assert max(worst.values()) <= LOGIT_RTOL, worst
        gc.record_parity(f"local_size_{tag}_variant{variant}_logits", max(worst.values()))
        g = eng.rollout_forward(env.problem, pol, M, acts[0, :, 1], L.MODE_GREEDY, variant=variant)
        assert int(g.tlen.max()) == T and torch.equal(g.actions[:, :, :T].cpu(), acts.int())
        np.testing.assert_allclose(g.reward.cpu().numpy(), fx[f"{tag}_greedy_reward"], rtol=1e-5)
        if variant == 0:                                             # the step-wise protocol (CVRPEnv.step + one_step_rollout)
            env.reset()
            state, _, done = env.pre_step()
            t = 0
            starts = [int(a) for a in acts[0, :, 1]]
            model.draw_starts = staticmethod(lambda n, m: starts)
            while not done and t < 12:
                sel, _ = model.one_step_rollout(state, *env.get_cur_feature(), eval_type='greedy')
                assert np.array_equal(sel.cpu().numpy(), acts[:, :, t].numpy()), t
                state, rew, done = env.step(sel)
                t += 1


@pytest.mark.parametrize("tag", ["k50", "k63"])
def test_local_size_above_47_training_gradients(tag):
    """The reference's REINFORCE step on its own sampled tours through the engine's replay backward (no saved rows above 47)."""
    from elg_amd.CVRP.train import pomo_loss
    from test_oracle_golden import wide_slots_setup
    fx, mp, cfg, P, xy, dem, B, N, M = wide_slots_setup(tag)
    model = _wide_model(mp, P).train()
    env = _env(mp, xy, dem, M)
    rs, _, _ = env.reset()
    model.pre_forward(rs)
    pol = model.decoder.policy
    sacts = torch.from_numpy(fx[f"{tag}_sample_actions"].astype(np.int64))
    T = sacts.shape[2]
    res = eng.rollout_forward(env.problem, pol, M, sacts[0, :, 1], L.MODE_FORCED, forced=sacts, train=True)
    assert res.rows is None and res.kernel_id == L.KERNEL_WAVE
    probs = eng.chosen_probs(env.problem, pol, M, res, T)
    np.testing.assert_allclose(probs.detach().cpu().numpy(), fx[f"{tag}_sample_probs"], rtol=5e-4, atol=1e-9)
    J = pomo_loss(probs, torch.from_numpy(fx[f"{tag}_sample_reward"]).to(DEV), True)
    assert abs(float(J.detach()) - float(fx[f"{tag}_loss"])) <= 2e-4 * max(1.0, abs(float(fx[f"{tag}_loss"])))
    J.backward()
    got = dict(model.named_parameters())
    worst = 0.0
    for n in [k[len(f"{tag}_grad_"):] for k in fx.files if k.startswith(f"{tag}_grad_")]:
        ref = fx[f"{tag}_grad_{n}"]
        err = float(np.abs(got[n].grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-6)
        worst = max(worst, err)
        assert err <= 2e-3, (n, err)
    gc.record_parity(f"local_size_{tag}_grad_over_max", worst)


def test_local_size_63_train_step_runs():
    """The product train_step at local_size 63 (host-synchronised path + replay backward) moves the local policy's parameters."""
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.optim import Adam
    from test_oracle_golden import wide_slots_setup
    fx, mp, cfg, P, xy, dem, B, N, M = wide_slots_setup("k63")
    model = _wide_model(mp, P).train()
    env = CVRPEnv(multi_width=M, device=DEV)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    torch.manual_seed(0)
    J, rew = train_step(model, env, opt, generate_vrp_data(3, N, {"data_type": "uniform"}))
    assert torch.isfinite(J).item()
    assert all(float((v.detach() - before[k]).abs().max()) > 0 for k, v in model.named_parameters() if k.startswith("decoder.local_policies."))


@pytest.mark.parametrize("K", [48, 63])
def test_tsp_local_size_above_47_against_the_oracle(K):
    """TSP with local_size 48 / 63 (no depot slot: K slots, the 64-lane limit is K <= 64; kept <= 63 like CVRP): the oracle's own
    greedy tours teacher-forced through the one-wavefront kernels -- chosen probabilities, rewards; the engine's free-running
    greedy rollout reproduces the tours; sampled rollout = its forced replay."""
    mp = dict(gu.TSP_MODEL_PARAMS)
    mp["local_size"] = [K]
    cfg = orc.ModelCfg.from_model_params(mp, "tsp")
    P = gc.weights("tsp", 17, mp, 1.0)
    B, N, M = 2, 100, 16
    xy = torch.from_numpy(gu.golden_tsp_problem(33, B, N))
    enc = orc.encoder_forward(P, cfg, xy)
    prob = gc.make_problem(xy, None, L.PROBLEM_TSP)
    pol = gc.make_policy(P, cfg, enc.to(DEV), L.PROBLEM_TSP)
    assert pol.wide_slots
    starts = torch.arange(M)
    out = orc.rollout_tsp(P, cfg, xy, M, starts=starts, mode="greedy", enc=enc)
    acts = out["actions"].int()
    res = eng.rollout_forward(prob, pol, M, starts, L.MODE_FORCED, forced=acts)
    assert res.kernel_id == L.KERNEL_WAVE
    T = acts.shape[2]
    np.testing.assert_allclose(res.probs[:, :T].cpu().numpy(), out["probs"].numpy(), rtol=5e-4, atol=1e-9)
    np.testing.assert_allclose(res.reward.cpu().numpy(), out["reward"].numpy(), rtol=1e-5)
    g = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY)
    agree = float((g.actions[:, :, :T].cpu() == acts).float().mean())
    assert agree > 0.995, agree                                   # (free-running arg-max: a near-tie may flip a step)
    s1 = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=5)
    s2 = eng.rollout_forward(prob, pol, M, starts, L.MODE_FORCED, forced=s1.actions)
    np.testing.assert_allclose(s1.probs.cpu().numpy(), s2.probs.cpu().numpy(), rtol=1e-6, atol=0)
