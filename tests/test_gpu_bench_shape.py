"""The bench configuration itself under the parity lens (VERDICT r03 items 8 / "soft spots"):

* the PRODUCT-DEFAULT training step at the bench shape's row count -- one CVRP-100 instance, pomo 100, the whole sampled
  construction (T ~ 115 decode steps, ~11 000 decode rows) through model.pre_forward -> rollout -> POMO loss -> backward --
  against the oracle evaluated in FLOAT64 on the same tours: every decoder / local-policy gradient within 1e-4 of the tensor's
  maximum with the default f32 backward (elg_decoder_bwd mode 0), within 1e-3 with the split-bf16 backward (mode 2);
* the bench launch geometry (64 instances x pomo 100 -> tiles = 4, 256 workgroups of 25 lockstep trajectories) diffed against
  the oracle's probabilities on two of its instances.

Worst observed values go to gpurun_out/parity_r06.json (copied to profiles/)."""
import random

import numpy as np
import pytest
import torch

import golden_util as gu
import gpu_common as gc
from oracle import elg_oracle as orc

pytestmark = pytest.mark.gpu
DEV = gc.DEV


def _cvrp100(seed, B):
    depot, loc, demand = gu.golden_cvrp_problem(seed, B, 100, 50.0)
    batch = dict(depot=torch.from_numpy(depot), loc=torch.from_numpy(loc), demand=torch.from_numpy(demand))
    xy = torch.from_numpy(np.concatenate([depot, loc], 1))
    dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1))
    return batch, xy, dem


@pytest.mark.parametrize("mode,bound", [(0, 1e-4), (2, 1e-3)], ids=["f32_default", "split_bf16_fast"])
def test_train_step_gradients_at_the_bench_row_count_vs_float64(mode, bound, monkeypatch):
    from elg_amd import engine as eng
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.train import pomo_loss
    from elg_amd.CVRP.utils import rollout
    monkeypatch.setattr(eng, "BWD_MFMA_MODE", mode)
    mp = dict(gu.CVRP_MODEL_PARAMS)
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    B, M = 1, 100
    batch, xy, dem = _cvrp100(2024, B)
    model = gc.load_model("cvrp", 23, mp, 1.0).train()
    env = CVRPEnv(multi_width=M, device=DEV)
    env.load_random_problems(batch)
    rs, _, _ = env.reset()
    model.pre_forward(rs)
    torch.manual_seed(8)
    random.seed(8)
    acts, probs, rew = rollout(model, env, 'sample')
    assert probs.requires_grad and eng.TrainRows._cache, "the step must run over the rows the training forward saved"
    T = acts.shape[2]
    assert T * M > 9000, "the point of this test is the bench's row count"
    rew_n = rew + 0.3 * torch.randn(B, M, device=rew.device)
    J = pomo_loss(probs, rew_n, True)
    J.backward()
    got = {k: v.grad.detach().cpu().double() for k, v in model.named_parameters()}
    # ---- the oracle in float64 on the same tours (float32 inputs converted exactly)
    P = {k: v.double().requires_grad_(True) for k, v in gc.weights("cvrp", 23, mp, 1.0).items()}
    a = acts.cpu()
    out = orc.rollout_cvrp(P, cfg, xy.double(), dem.double(), M, starts=a[0, :, 1], forced=a)
    pe = float(((probs.detach().cpu().double() - out["probs"].detach()).abs() / out["probs"].detach()).max())
    assert pe <= 5e-4, f"chosen probabilities off by {pe:.2e}"
    Jo = orc.pomo_loss(out["probs"], rew_n.cpu().double(), True)
    Jo.backward()
    worst_dec, worst_enc = 0.0, 0.0
    # (the biases in front of an instance norm have an exactly-zero true gradient -- the norm removes the per-channel mean --:
    # encoder tensors are measured against max(their own maximum, 1e-3 of the largest encoder gradient))
    enc_floor = 1e-3 * max(float(p.grad.abs().max()) for k, p in P.items() if k.startswith("encoder."))
    for k, p in P.items():
        r = p.grad
        scale = float(r.abs().max()) if k.startswith("decoder.") else max(float(r.abs().max()), enc_floor)
        err = float((got[k] - r).abs().max()) / scale
        if k.startswith("decoder."):
            worst_dec = max(worst_dec, err)
            assert err <= bound, f"{k}: {err:.3e} of the tensor maximum (bound {bound:g}, mode {mode})"
        else:
            worst_enc = max(worst_enc, err)
    gc.record_parity(f"bench_rows/mode{mode}_decoder_local_grad_over_tensor_max", worst_dec)
    gc.record_parity(f"bench_rows/mode{mode}_encoder_grad_over_tensor_max", worst_enc)
    gc.record_parity("bench_rows/chosen_prob_rel_vs_float64", pe)
    print(f"mode {mode}: T = {T}, decoder/local {worst_dec:.2e}, encoder {worst_enc:.2e}, chosen probabilities {pe:.2e}")
    # the encoder's gradients pass through six ReLU layers: a unit at its kink may move single rows (see test_gpu_train_large)
    assert worst_enc <= 2e-3, worst_enc


def test_bench_launch_geometry_against_the_oracle():
    """64 instances x pomo 100, the launch bench.py times (tiles = 4: 256 workgroups, groups of 25 lockstep trajectories,
    XCD-aware unit map), sampled; the oracle teacher-forced on two of the instances (the first and one that lands on another
    XCD / tile pattern): chosen probabilities at every step, rewards, and the whole probability rows at three steps."""
    from elg_amd import _lib as L, engine as eng
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    mp = dict(gu.CVRP_MODEL_PARAMS)
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    B, M = 64, 100
    batch, xy, dem = _cvrp100(77, B)
    model = gc.load_model("cvrp", 23, mp, 1.0)
    env = CVRPEnv(multi_width=M, device=DEV)
    env.load_random_problems(batch)
    rs, _, _ = env.reset()
    with torch.no_grad():
        model.pre_forward(rs)
    pol = model.decoder.policy
    assert eng.launch_geometry(B, M, 101) == (8, 4, 1)
    starts = torch.tensor(random.Random(5).sample(range(100), M), dtype=torch.int32)
    dump_T = 40
    r = eng.rollout_forward(env.problem, pol, M, starts, L.MODE_SAMPLE, seed=4242, dump_T=dump_T)
    T = int(r.tlen.max())
    acts = r.actions[:, :, :T].cpu().long()
    P = gc.weights("cvrp", 23, mp, 1.0)
    worst_p = worst_row = 0.0
    for b in (0, 37):
        out = orc.rollout_cvrp(P, cfg, xy[b:b + 1], dem[b:b + 1], M, starts=starts.long(), forced=acts[b:b + 1], keep_probs=True)
        To = out["probs"].shape[1]
        got = r.probs[b:b + 1, :To].cpu().numpy()
        ref = out["probs"].numpy()
        worst_p = max(worst_p, float((np.abs(got - ref) / ref).max()))
        np.testing.assert_allclose(got, ref, rtol=5e-4, atol=1e-9)
        np.testing.assert_allclose(r.reward[b:b + 1].cpu().numpy(), out["reward"].numpy(), rtol=1e-5)
        tl = r.tlen[b].cpu().numpy()
        for t in (2, 17, 39):
            full_ref = out["full_probs"][t - 2][0].numpy()              # (M,N1), decode steps start at t = 2
            full_got = r.full_probs[b, :, t].cpu().numpy()
            live = t < tl
            gc.assert_same_mask(full_got[live], full_ref[live], f"instance {b} step {t}")
            keep = live[:, None] & (full_ref > 1e-30)
            worst_row = max(worst_row, float((np.abs(full_got[keep] - full_ref[keep]) / full_ref[keep]).max()))
    assert worst_row <= 5e-4, worst_row
    gc.record_parity("bench_geometry/chosen_prob_rel", worst_p)
    gc.record_parity("bench_geometry/probability_rows_rel", worst_row)
    print(f"bench geometry: chosen probabilities {worst_p:.2e}, probability rows {worst_row:.2e}")


def test_side_stream_local_backward_equals_the_inline_launch(monkeypatch):
    """engine.SIDE_LOCAL_BWD: the local-policy row backward on a side stream over half the CUs, next to the encoder's backward
    chain (the fold backward waits for it).  Same training step with the side stream forced on (threshold 0) and off: every
    parameter gradient equal up to the order of the float atomics."""
    from elg_amd import engine as eng
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.train import pomo_loss
    from elg_amd.CVRP.utils import rollout
    mp = dict(gu.CVRP_MODEL_PARAMS)
    batch, xy, dem = _cvrp100(31, 4)
    grads = {}
    for side in (True, False):
        monkeypatch.setattr(eng, "SIDE_LOCAL_BWD", side)
        monkeypatch.setattr(eng, "SIDE_LOCAL_MIN_TILES_PER_CU", 0)
        model = gc.load_model("cvrp", 23, mp, 1.0).train()
        env = CVRPEnv(multi_width=100, device=DEV)
        env.load_random_problems(batch)
        rs, _, _ = env.reset()
        model.pre_forward(rs)
        torch.manual_seed(8)
        random.seed(8)
        acts, probs, rew = rollout(model, env, 'sample')
        torch.manual_seed(9)
        J = pomo_loss(probs, rew + 0.3 * torch.randn(4, 100, device=rew.device), True)
        J.backward()
        torch.cuda.synchronize()
        assert not eng._PENDING_SIDE, "the fold backward must have consumed the side-stream event"
        grads[side] = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
    worst = 0.0
    # (the biases in front of an instance norm have an exactly-zero true gradient: measured against a floor, as above)
    floor = 1e-2 * max(float(v.abs().max()) for v in grads[False].values())
    for k, g in grads[True].items():
        ref = grads[False][k]
        err = float((g - ref).abs().max()) / max(float(ref.abs().max()), floor)
        worst = max(worst, err)
        assert err <= 2e-4, (k, err)            # (two runs of the SAME path differ by ~3e-5: float atomics)
    gc.record_parity("side_stream_local_bwd/grad_rel_diff", worst)


@pytest.mark.parametrize("problem,N", [("cvrp", 120), ("tsp", 127), ("cvrp", 112), ("tsp", 113)])
def test_product_train_step_in_the_112_to_128_node_band(problem, N):
    """112 < N + 1 <= 128: too wide for the cooperative kernel (its operand images hold 7 node tiles), still inside the
    one-instance-per-row-block encoder and the N1 <= 128 decoder backward -- the product step takes the one-wavefront training
    forward with STORED glimpse weights there (engine.rollout_forward: trA), a path the fixed-size tests reach only through
    parametrisations.  Whole product step (pre_forward -> sampled rollout -> POMO loss -> backward) against the oracle's autograd
    in float64 on the same tours: chosen probabilities 5e-4, decoder / local-policy gradients 1e-3 of the tensor's maximum."""
    from elg_amd import engine as eng
    if problem == "cvrp":
        from elg_amd.CVRP.CVRPEnv import CVRPEnv as Env
        from elg_amd.CVRP.train import pomo_loss
        from elg_amd.CVRP.utils import rollout
        mp = dict(gu.CVRP_MODEL_PARAMS)
        depot, loc, demand = gu.golden_cvrp_problem(500 + N, 1, N, 50.0)
        batch = dict(depot=torch.from_numpy(depot), loc=torch.from_numpy(loc), demand=torch.from_numpy(demand))
        xy = torch.from_numpy(np.concatenate([depot, loc], 1))
        dem = torch.from_numpy(np.concatenate([np.zeros((1, 1), np.float32), demand], 1))
    else:
        from elg_amd.TSP.TSPEnv import TSPEnv as Env
        from elg_amd.TSP.train import pomo_loss
        from elg_amd.TSP.utils import rollout
        mp = dict(gu.TSP_MODEL_PARAMS)
        xy = torch.from_numpy(gu.golden_tsp_problem(500 + N, 1, N))
        batch, dem = xy, None
    cfg = orc.ModelCfg.from_model_params(mp, problem)
    M = 12
    model = gc.load_model(problem, 29, mp, 1.0).train()
    env = Env(multi_width=M, device=DEV)
    env.load_random_problems(batch)
    rs, _, _ = env.reset()
    model.pre_forward(rs)
    torch.manual_seed(5)
    random.seed(5)
    acts, probs, rew = rollout(model, env, 'sample')
    assert probs.requires_grad
    rew_n = rew + 0.3 * torch.randn(1, M, device=rew.device)
    J = pomo_loss(probs, rew_n, True) if problem == "cvrp" else pomo_loss(probs, rew_n)
    J.backward()
    got = {k: v.grad.detach().cpu().double() for k, v in model.named_parameters()}
    P = {k: v.double().requires_grad_(True) for k, v in gc.weights(problem, 29, mp, 1.0).items()}
    a = acts.cpu().long()
    if problem == "cvrp":
        out = orc.rollout_cvrp(P, cfg, xy.double(), dem.double(), M, starts=a[0, :, 1], forced=a)
    else:
        out = orc.rollout_tsp(P, cfg, xy.double(), M, starts=a[0, :, 0], forced=a)
    ref_p = out["probs"].detach()
    pe = float(((probs.detach().cpu().double()[:, :ref_p.shape[1]] - ref_p).abs() / ref_p).max())
    assert pe <= 5e-4, f"chosen probabilities off by {pe:.2e}"
    Jo = orc.pomo_loss(out["probs"], rew_n.cpu().double(), True, guard_zero=(problem == "tsp"))
    Jo.backward()
    worst = 0.0
    for k, p in P.items():
        if not k.startswith("decoder.") or p.grad is None:
            continue
        err = float((got[k] - p.grad).abs().max()) / float(p.grad.abs().max())
        worst = max(worst, err)
        assert err <= 1e-3, f"{k}: {err:.3e} of the tensor maximum"
    gc.record_parity(f"band_112_128/{problem}_n{N}_decoder_local_grad_over_tensor_max", worst)
    gc.record_parity(f"band_112_128/{problem}_n{N}_chosen_prob_rel", pe)
    print(problem, N, f"decoder/local gradients {worst:.2e} of the tensor maximum, chosen probabilities {pe:.2e}")


def test_bf16_mode_says_where_it_runs_in_f32(monkeypatch):
    """The bf16 mode covers the cooperative kernel (N + 1 <= 112) and the streaming kernels' evaluation.  Where a launch runs in f32
    although the mode is on (113 <= N + 1 <= 128: one-wavefront kernel; a TRAINING forward above 128 nodes), the engine warns once
    per case instead of switching silently; the result is the f32 result."""
    import warnings
    from elg_amd import engine as eng
    from elg_amd.TSP.TSPEnv import TSPEnv
    from elg_amd.TSP.utils import rollout
    mp = dict(gu.TSP_MODEL_PARAMS)
    model = gc.load_model("tsp", 31, mp, 1.0).eval()
    xy = torch.from_numpy(gu.golden_tsp_problem(640, 2, 120))
    env = TSPEnv(8, gc.DEV)
    env.load_random_problems(xy)
    rs, _, _ = env.reset()
    with torch.no_grad():
        model.pre_forward(rs)
        random.seed(5)
        ref_actions, _, ref_reward = rollout(model, env, "greedy")
        monkeypatch.setattr(eng, "FWD_PRECISION", 1)
        eng._F32_NOTED.clear()
        random.seed(5)
        with pytest.warns(RuntimeWarning, match="runs in f32"):
            actions, _, reward = rollout(model, env, "greedy")
        with warnings.catch_warnings():
            warnings.simplefilter("error")                 # ... and only once
            random.seed(5)
            rollout(model, env, "greedy")
    assert torch.equal(actions, ref_actions) and torch.equal(reward, ref_reward)
