"""model_params['ensemble_size'] > 1 (reference CVRP/models.py:296-298: one local_policy_att per member, each with its own
local_size; :409-413: the mean of the members' outputs) through the product path, against a fixture of the real reference
with two members (tools/make_golden_r03.py; a: local_size [12, 6], b: [6, 12] -- the second member looks further than the
distance penalty).  Logit tolerance as in test_gpu_logits.py; gradients of the reference's REINFORCE step."""
import numpy as np
import pytest
import torch

import gpu_common as gc
from elg_amd import _lib as L
from elg_amd import engine as eng
from test_oracle_golden import LOGIT_RTOL, ensemble_setup, logit_errors

pytestmark = pytest.mark.gpu
DEV = gc.DEV


def _model(mp, P):
    from elg_amd.CVRP.CVRPModel import CVRPModel
    model = CVRPModel(**mp)
    model.decoder.add_local_policy(DEV)
    assert len(model.decoder.local_policies) == mp["ensemble_size"]
    model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)     # the reference's checkpoint keys
    return model.to(DEV)


def _env(xy, dem, M):
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    env = CVRPEnv(multi_width=M, device=DEV)
    env.load_random_problems(dict(depot=xy[:, :1].clone(), loc=xy[:, 1:].clone(), demand=dem[:, 1:].clone()))
    return env


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ensemble_logits_and_greedy_tours(tag):
    fx, mp, cfg, P, xy, dem, B, N, M = ensemble_setup(tag)
    model = _model(mp, P).eval()
    env = _env(xy, dem, M)
    acts = torch.from_numpy(fx[f"{tag}_greedy_actions"].astype(np.int64))
    T = acts.shape[2]
    with torch.no_grad():
        rs, _, _ = env.reset()
        model.pre_forward(rs)
        pol = model.decoder.policy
        assert pol.ens == 2 and tuple(pol.Ks) == tuple(mp["local_size"]) and pol.loc.numel() == 2 * L.LOC_SIZE
        worst = {}
        for dump, key, scale in (("scores", "pre_clip", None), ("logits", "logits", cfg.logit_clipping)):
            res = eng.rollout_forward(env.problem, pol, M, acts[0, :, 1], L.MODE_FORCED, forced=acts, dump_T=T, dump=dump)
            got = res.full_probs.cpu().numpy()
            for i, t in enumerate(fx[f"{tag}_steps"]):
                ref = fx[f"{tag}_{key}"][i]
                open_ = np.isfinite(fx[f"{tag}_logits"][i])
                if key == "logits":
                    assert np.array_equal(np.isfinite(got[:, :, int(t), :]), open_), int(t)
                worst[key] = max(worst.get(key, 0.0), logit_errors(got[:, :, int(t), :], ref, open_, scale))
        assert max(worst.values()) <= LOGIT_RTOL, worst
        gc.record_parity(f"ensemble2_{tag}_logits", max(worst.values()))
        # free-running: the reference's own greedy tours and rewards, fused rollout and the step-wise protocol
        from elg_amd.CVRP.utils import rollout
        starts = [int(a) for a in acts[0, :, 1]]
        model.draw_starts = staticmethod(lambda n, m: starts)
        a, _, r = rollout(model, env, 'greedy')
        assert np.array_equal(a.cpu().numpy(), acts.numpy())
        np.testing.assert_allclose(r.cpu().numpy(), fx[f"{tag}_greedy_reward"], rtol=1e-5)
        env.reset()
        state, _, done = env.pre_step()
        t = 0
        while not done:
            sel, _ = model.one_step_rollout(state, *env.get_cur_feature()[:3], eval_type='greedy')
            assert np.array_equal(sel.cpu().numpy(), acts[:, :, t].numpy()), t
            state, rew, done = env.step(sel)
            t += 1
        assert t == T


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ensemble_training_gradients(tag):
    """The reference's REINFORCE step (train.py:107-123) on its own sampled tours, replayed through the engine's backward:
    chosen probabilities, loss, every decoder and local-policy gradient (both members)."""
    from elg_amd.CVRP.train import pomo_loss
    fx, mp, cfg, P, xy, dem, B, N, M = ensemble_setup(tag)
    model = _model(mp, P).train()
    env = _env(xy, dem, M)
    rs, _, _ = env.reset()
    model.pre_forward(rs)
    pol = model.decoder.policy
    sacts = torch.from_numpy(fx[f"{tag}_sample_actions"].astype(np.int64))
    T = sacts.shape[2]
    res = eng.rollout_forward(env.problem, pol, M, sacts[0, :, 1], L.MODE_FORCED, forced=sacts, train=True)
    assert res.rows is None                                           # an ensemble trains through the replay backward
    assert int(res.tlen.max()) == T
    probs = eng.chosen_probs(env.problem, pol, M, res, T)
    np.testing.assert_allclose(probs.detach().cpu().numpy(), fx[f"{tag}_sample_probs"], rtol=5e-4, atol=1e-9)
    np.testing.assert_allclose(res.reward.cpu().numpy(), fx[f"{tag}_sample_reward"], rtol=1e-5)
    J = pomo_loss(probs, torch.from_numpy(fx[f"{tag}_sample_reward"]).to(DEV), True)
    assert abs(float(J.detach()) - float(fx[f"{tag}_loss"])) <= 2e-4 * max(1.0, abs(float(fx[f"{tag}_loss"])))
    J.backward()
    got = dict(model.named_parameters())
    names = [k[len(f"{tag}_grad_"):] for k in fx.files if k.startswith(f"{tag}_grad_")]
    worst = 0.0
    for n in names:
        ref = fx[f"{tag}_grad_{n}"]
        g = got[n].grad.cpu().numpy()
        err = float(np.abs(g - ref).max()) / max(float(np.abs(ref).max()), 1e-6)
        worst = max(worst, err)
        assert err <= 2e-3, (n, err)
    gc.record_parity(f"ensemble2_{tag}_grad_over_max", worst)


def test_ensemble_train_step_runs():
    """The product train_step takes the replay branch for an ensemble and moves every member's parameters."""
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.optim import Adam
    fx, mp, cfg, P, xy, dem, B, N, M = ensemble_setup("b")
    model = _model(mp, P).train()
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    env = CVRPEnv(multi_width=M, device=DEV)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    torch.manual_seed(0)
    J, rew = train_step(model, env, opt, generate_vrp_data(4, N, {"data_type": "uniform"}))
    assert torch.isfinite(J).item()
    for k, v in model.named_parameters():
        if k.startswith("decoder.local_policies."):
            assert float((v.detach() - before[k]).abs().max()) > 0, k
