"""GPU parity of the backward replay kernels: gradients of a loss over the chosen-node probabilities
w.r.t. the decoder / local-policy parameters and the encoder output, against the oracle's autograd
(which itself is pinned on the reference's train() step, tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import elg_oracle as orc

pytestmark = pytest.mark.gpu


def _imports():
    import gpu_common as gc
    from elg_amd import _lib as L
    from elg_amd import engine as eng
    return gc, L, eng


def _grad_check(got: dict, ref: dict, rtol=2e-3):
    rms = max(float(v.norm()) / np.sqrt(v.numel()) for v in ref.values())
    atol = 2e-3 * rms
    worst = {}
    for k, r in ref.items():
        g = got[k].detach().cpu()
        err = float((g - r).abs().max())
        lim = rtol * float(r.abs().max()) + atol
        worst[k] = err / lim
        assert err <= lim, f"{k}: max abs err {err:.3e} > {lim:.3e} (ref max {float(r.abs().max()):.3e})"
    return worst


def _run(problem, tag, loss_kind, geometry=None, train=False, precision=0, bf16_bounds=None):
    gc, L, eng = _imports()
    if problem == "cvrp":
        fx, cfg, P, xy, dem, B, N, M = gc.cvrp_fixture(tag)
        kind = L.PROBLEM_CVRP
        lp_prefix, nfeat, nslots = "decoder.local_policies.0.", 3, cfg.local_size + 1
    else:
        fx, cfg, P, xy, B, N, M = gc.tsp_fixture(tag)
        dem = None
        kind = L.PROBLEM_TSP
        lp_prefix, nfeat, nslots = "decoder.local_policy_0.", 2, cfg.local_size
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    T = acts.shape[2]
    torch.manual_seed(3)
    W = torch.randn(B, T, M)
    # perturb the rewards: an instance whose trajectories all found the same tour has advantage == rounding
    # noise, and the reference's scale_norm then divides by ~0 -- ill-conditioned
    rew = torch.from_numpy(fx["reward"]) + 0.3 * torch.randn(B, M)

    def loss_fn(probs):
        if loss_kind == "weighted":
            return (probs * W.to(probs.device)).sum()
        return orc.pomo_loss(probs, rew.to(probs.device), guard_zero=(problem == "tsp"))

    # ---------------- oracle (CPU autograd)
    Po = {k: v.clone().requires_grad_(k.startswith("decoder.")) for k, v in P.items()}
    with torch.no_grad():
        enc0 = orc.encoder_forward(P, cfg, xy, dem) if problem == "cvrp" else orc.encoder_forward(P, cfg, xy)
    enc_o = enc0.clone().requires_grad_(True)
    if problem == "cvrp":
        out = orc.rollout_cvrp(Po, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, enc=enc_o)
    else:
        out = orc.rollout_tsp(Po, cfg, xy, M, starts=acts[0, :, 0], forced=acts, enc=enc_o)
    Jo = loss_fn(out["probs"])
    Jo.backward()
    ref = {k: v.grad.clone() for k, v in Po.items() if v.grad is not None}
    ref["enc"] = enc_o.grad.clone()

    # ---------------- engine
    Pg = {k: v.clone().to(gc.DEV).requires_grad_(k.startswith("decoder.")) for k, v in P.items()}
    enc_g = enc0.clone().to(gc.DEV).requires_grad_(True)
    tables = gc.fold_decoder_tables(gc.sub(Pg, "decoder."), enc_g, kind)
    loc = gc.fold_local_tables(gc.sub(Pg, lp_prefix), nfeat, nslots)
    pol = eng.Policy(tables, loc, cfg.local_size, cfg.xi, cfg.logit_clipping, 1.0 / cfg.ensemble_size, True, True)
    prob = gc.make_problem(xy, dem, kind)
    starts = acts[0, :, 1] if problem == "cvrp" else acts[0, :, 0]
    res = eng.rollout_forward(prob, pol, M, starts, L.MODE_FORCED, forced=acts, geometry=geometry, train=train, precision=precision)
    assert (res.rows is not None) == train
    pr = eng.chosen_probs(prob, pol, M, res, T, geometry=geometry)
    if precision == 1:
        # The mode's own oracle (oracle/elg_oracle.py precision="bf16", float64): forward values from the bf16-rounded operands,
        # derivatives of the f32 formulas at those values (what elg_decoder_bwd mode 3 computes), rounded values taken from the
        # engine's own f32 tables.
        assert res.rows.precision == 1
        P64 = {k: v.double().clone().requires_grad_(k.startswith("decoder.")) for k, v in P.items()}
        enc64 = enc0.double().clone().requires_grad_(True)
        t64 = orc.fold_tables(P64, cfg, enc64)
        tv = {k: (None if v is None else v.detach().float().cpu()) for k, v in tables.items()}
        if problem == "cvrp":
            ob = orc.rollout_cvrp(P64, cfg, xy.double(), dem.double(), M, starts=acts[0, :, 1], forced=acts, enc=enc64, tables=t64,
                                  tables_val=tv, precision="bf16")
        else:
            ob = orc.rollout_tsp(P64, cfg, xy.double(), M, starts=acts[0, :, 0], forced=acts, enc=enc64, tables=t64, tables_val=tv,
                                 precision="bf16")
        W64, rew64 = W.double(), rew.double()
        Jb = (ob["probs"] * W64).sum() if loss_kind == "weighted" else orc.pomo_loss(ob["probs"], rew64, guard_zero=(problem == "tsp"))
        Jb.backward()
        refb = {k: v.grad.clone() for k, v in P64.items() if v.grad is not None}
        refb["enc"] = enc64.grad.clone()
        rel_p = ((pr.detach().cpu().double() - ob["probs"].detach()).abs() / ob["probs"].detach()).flatten()
        pe, p_frac = float(rel_p.max()), float((rel_p <= 2e-3).double().mean())
        pe32 = float(((pr.detach().cpu() - out["probs"].detach()).abs() / out["probs"].detach()).max())
        print(problem, tag, f"bf16 mode: chosen probabilities vs the bf16 oracle: {p_frac:.4f} within 2e-3, worst {pe:.2e} (vs the f32 oracle {pe32:.2e})")
        gc.record_parity(f"bf16_mode/{problem}_{tag}_chosen_prob_rel_vs_bf16_oracle", pe)
        gc.record_parity(f"bf16_mode/{problem}_{tag}_chosen_prob_fraction_within_2e-3_of_bf16_oracle", p_frac)
        gc.record_parity(f"bf16_mode/{problem}_{tag}_chosen_prob_rel_vs_f32_oracle", pe32)
        Jg = loss_fn(pr)
        Jg.backward()
        got = {k: v.grad for k, v in Pg.items() if v.grad is not None}
        got["enc"] = enc_g.grad
        assert set(got) == set(refb)
        prob_rel, grad_rel = bf16_bounds
        # (a score on a bf16 rounding boundary moves by up to 4e-4 -- test_gpu_logits -- and the clip's factor 50 makes that 2 % of a
        # single probability: almost all of them agree to 2e-3, the rest are bounded)
        assert p_frac >= 0.99 and pe <= prob_rel, f"chosen probabilities off the bf16 oracle: {p_frac:.4f} within 2e-3, worst {pe:.3e}"
        # every gradient entry within grad_rel of the tensor's largest entry (+ the f32 floor of _grad_check), and the tensor clearly
        # closer to the bf16 oracle than to the f32 one
        rms = max(float(v.norm()) / np.sqrt(v.numel()) for v in refb.values())
        worst, closer = {}, {}
        for k, r in refb.items():
            g = got[k].detach().cpu().double()
            worst[k] = float((g - r).abs().max()) / (float(r.abs().max()) + 2.0 * rms)
            closer[k] = float((g - r).norm() / (g - ref[k].double()).norm().clamp_min(1e-30))
        gc.record_parity(f"bf16_mode/{problem}_{tag}_grad_err_over_max_vs_bf16_oracle", max(worst.values()))
        gc.record_parity(f"bf16_mode/{problem}_{tag}_grad_l2_to_bf16_oracle_over_l2_to_f32_oracle", max(closer.values()))
        print(problem, tag, "bf16 mode, gradient error / tensor max:", {k.split(".")[-1] if "." in k else k: f"{v:.1e}" for k, v in worst.items()},
              " L2 distance to the bf16 oracle / to the f32 oracle:", f"{max(closer.values()):.2f}")
        bad = {k: v for k, v in worst.items() if v > grad_rel}
        assert not bad, f"gradients off the bf16 oracle: {bad} (bound {grad_rel})"
        assert max(closer.values()) < 0.5, closer
        return
    np.testing.assert_allclose(pr.detach().cpu().numpy(), out["probs"].detach().numpy(), rtol=5e-4)
    Jg = loss_fn(pr)
    assert abs(Jg.item() - Jo.item()) <= 2e-4 * max(1.0, abs(Jo.item()))
    Jg.backward()
    got = {k: v.grad for k, v in Pg.items() if v.grad is not None}
    got["enc"] = enc_g.grad
    assert set(got) == set(ref)
    worst = _grad_check(got, ref)
    print(problem, tag, loss_kind, "worst err/limit:", {k.split(".")[-2] + "." + k.split(".")[-1] if "." in k else k: round(v, 3)
                                                        for k, v in worst.items()})


@pytest.mark.parametrize("tag", ["n20", "n20k8", "n50"])
@pytest.mark.parametrize("loss_kind", ["weighted", "pomo"])
@pytest.mark.parametrize("train", [False, True], ids=["replay", "saved_rows"])
def test_cvrp_backward(tag, loss_kind, train):
    """replay: rows recomputed by rollout_bwd_kernel; saved_rows: rows written by the training forward."""
    _run("cvrp", tag, loss_kind, train=train)


def test_cvrp_backward_global_memory_variant():
    _run("cvrp", "n20", "weighted", geometry=(8, 2, 0))


@pytest.mark.parametrize("train", [False, True], ids=["replay", "saved_rows"])
def test_cvrp_backward_n100(train):
    _run("cvrp", "n100", "pomo", train=train)


@pytest.mark.parametrize("tag", ["n20", "n50"])
@pytest.mark.parametrize("train", [False, True], ids=["replay", "saved_rows"])
def test_tsp_backward(tag, train):
    _run("tsp", tag, "pomo", train=train)


@pytest.mark.parametrize("problem,tag,bounds", [("cvrp", "n100", (6e-2, 1e-2)), ("tsp", "n50", (6e-2, 1e-2)), ("cvrp", "n50", (6e-2, 1e-2))])
def test_bf16_mode_training_gradients(problem, tag, bounds):
    """The bf16 throughput mode end to end (elg_rollout_args.precision = 1 forward -> elg_decoder_bwd mode 3, whose score
    recompute rounds q and K as the forward did): chosen probabilities and REINFORCE gradients against the oracle's bf16
    restatement in float64 (same rounded operands, derivatives of the f32 formulas at the forward's values) -- bounds =
    (worst relative error of a chosen probability, worst gradient entry error over the tensor's largest entry); 99 % of the chosen
    probabilities within 2e-3; every gradient tensor at least twice as close (L2) to the bf16 oracle as to the f32 one.  A backward
    that is inconsistent with its forward (the f32 backward on a bf16 forward, a head's sign) is off by per cents to factors."""
    _run(problem, tag, "pomo", train=True, precision=1, bf16_bounds=bounds)


@pytest.mark.parametrize("problem,N,M", [("cvrp", 200, 8), ("tsp", 150, 6), ("cvrp", 300, 4), ("tsp", 530, 3)])
def test_backward_large_instance_replay(problem, N, M):
    """N1 > 128 (no saved-row training forward for these sizes): the engine's own sampled tours, gradients through the
    replay backward (rollout_bwd_kernel + the row contractions) against the oracle's autograd on the same forced tours."""
    gc, L, eng = _imports()
    B = 1
    if problem == "cvrp":
        mp = dict(gu.CVRP_MODEL_PARAMS)
        cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
        depot, loc_xy, demand = gu.golden_cvrp_problem(61 + N, B, N, 50.0)
        xy = torch.from_numpy(np.concatenate([depot, loc_xy], 1))
        dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1))
        kind, starts = L.PROBLEM_CVRP, torch.arange(1, M + 1)
        lp_prefix, nfeat, nslots = "decoder.local_policies.0.", 3, cfg.local_size + 1
    else:
        mp = dict(gu.TSP_MODEL_PARAMS)
        cfg = orc.ModelCfg.from_model_params(mp, "tsp")
        xy, dem = torch.from_numpy(gu.golden_tsp_problem(61 + N, B, N)), None
        kind, starts = L.PROBLEM_TSP, torch.arange(M)
        lp_prefix, nfeat, nslots = "decoder.local_policy_0.", 2, cfg.local_size
    P = gc.weights(problem, 21, mp, 1.0)
    with torch.no_grad():
        enc0 = orc.encoder_forward(P, cfg, xy, dem)
    prob = gc.make_problem(xy, dem, kind)
    Pg = {k: v.clone().to(gc.DEV).requires_grad_(k.startswith("decoder.")) for k, v in P.items()}
    enc_g = enc0.clone().to(gc.DEV).requires_grad_(True)
    tables = gc.fold_decoder_tables(gc.sub(Pg, "decoder."), enc_g, kind)
    loc = gc.fold_local_tables(gc.sub(Pg, lp_prefix), nfeat, nslots)
    pol = eng.Policy(tables, loc, cfg.local_size, cfg.xi, cfg.logit_clipping, 1.0 / cfg.ensemble_size, True, True)
    with torch.no_grad():
        res = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=9)
    T = int(res.tlen.max().item())
    acts = res.actions[:, :, :T].cpu().long()
    torch.manual_seed(4)
    rew = res.reward.cpu() + 0.3 * torch.randn(B, M)

    def loss_fn(probs):
        return orc.pomo_loss(probs, rew.to(probs.device), guard_zero=(problem == "tsp"))
    Po = {k: v.clone().requires_grad_(k.startswith("decoder.")) for k, v in P.items()}
    enc_o = enc0.clone().requires_grad_(True)
    if problem == "cvrp":
        out = orc.rollout_cvrp(Po, cfg, xy, dem, M, starts=starts, forced=acts, enc=enc_o)
    else:
        out = orc.rollout_tsp(Po, cfg, xy, M, starts=starts, forced=acts, enc=enc_o)
    Jo = loss_fn(out["probs"])
    Jo.backward()
    ref = {k: v.grad.clone() for k, v in Po.items() if v.grad is not None}
    ref["enc"] = enc_o.grad.clone()
    pr = eng.chosen_probs(prob, pol, M, res, T)
    np.testing.assert_allclose(pr.detach().cpu().numpy(), out["probs"].detach().numpy(), rtol=5e-4)
    Jg = loss_fn(pr)
    assert abs(Jg.item() - Jo.item()) <= 2e-4 * max(1.0, abs(Jo.item()))
    Jg.backward()
    got = {k: v.grad for k, v in Pg.items() if v.grad is not None}
    got["enc"] = enc_g.grad
    assert set(got) == set(ref)
    _grad_check(got, ref)
