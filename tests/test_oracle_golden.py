"""Pin the oracle (oracle/elg_oracle.py) against golden vectors produced by the real reference
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import elg_oracle as orc

torch.set_num_threads(4)


def _weights(problem, seed, mp, gain, dtype=torch.float32):
    return {k: torch.from_numpy(v).to(dtype) for k, v in gu.golden_weights(problem, seed, mp, True, gain).items()}


def _cvrp_setup(fx, dtype=torch.float32):
    B, N, M, wseed, pseed, local_size, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    mp["local_size"] = [local_size]
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    P = _weights("cvrp", wseed, mp, float(fx["gain"]), dtype)
    depot, loc, demand = gu.golden_cvrp_problem(pseed, B, N, float(fx["capacity"]))
    xy = torch.from_numpy(np.concatenate([depot, loc], 1)).to(dtype)
    dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1)).to(dtype)
    return cfg, P, xy, dem, B, N, M


CVRP_TAGS = ["n20", "n20k8", "n50", "n100", "greedy_n20"]


@pytest.mark.parametrize("tag", CVRP_TAGS)
def test_cvrp_env_bit_exact(tag):
    """G1: load, feasibility mask and finished flags of every step, bit for bit (CVRPEnv.step)."""
    fx = gu.load_golden(f"cvrp_rollout_{tag}.npz")
    cfg, P, xy, dem, B, N, M = _cvrp_setup(fx)
    env = orc.CvrpEnvOracle(dem.numpy(), M)
    acts = fx["actions"].astype(np.int64)
    T = acts.shape[2]
    for t in range(T):
        done = env.step(acts[:, :, t])
        assert np.array_equal(env.load.view(np.uint32), fx["load"][t].view(np.uint32)), f"load differs at t={t}"
        assert np.array_equal(np.packbits(env.mask.astype(np.uint8), axis=-1), fx["maskbits"][t]), f"mask t={t}"
        assert np.array_equal(env.finished, fx["finished"][t]), f"finished t={t}"
        assert done == (t == T - 1)
    r = -orc.route_length(xy, torch.from_numpy(acts))
    np.testing.assert_allclose(r.numpy(), fx["reward"], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("tag", CVRP_TAGS)
def test_cvrp_encoder(tag):
    """G5: encoder output."""
    fx = gu.load_golden(f"cvrp_rollout_{tag}.npz")
    cfg, P, xy, dem, B, N, M = _cvrp_setup(fx)
    enc = orc.encoder_forward(P, cfg, xy, dem)
    np.testing.assert_allclose(enc.numpy(), fx["enc"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(orc.dist_matrix(xy).numpy(), fx["dist"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag", CVRP_TAGS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_cvrp_decode_probs(tag, dtype):
    """G2-G4: teacher-forced full probability rows + chosen probabilities (decoder + local policy)."""
    fx = gu.load_golden(f"cvrp_rollout_{tag}.npz")
    cfg, P, xy, dem, B, N, M = _cvrp_setup(fx, dtype)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_probs=True)
    assert out["actions"].shape == acts.shape
    pf_t = fx["pf_t"]
    worst = 0.0
    for i, t in enumerate(pf_t):
        ref = fx["pf"][i]
        got = out["full_probs"][t - 2].numpy()
        assert np.array_equal(ref == 0, got == 0), f"zero pattern differs at t={t}"
        nz = ref > 1e-30
        rel = np.abs(got[nz] - ref[nz]) / ref[nz]
        # tiny probabilities sit behind exp(-50..); compare those in log space
        big = ref[nz] > 1e-6
        worst = max(worst, rel[big].max())
        assert rel[big].max() < 5e-4, f"t={t}: rel err {rel[big].max()}"
        assert np.abs(np.log(got[nz]) - np.log(ref[nz])).max() < 2e-3
    if "sel_prob" in fx.files:
        np.testing.assert_allclose(out["probs"].numpy(), fx["sel_prob"], rtol=5e-4, atol=1e-9)
    np.testing.assert_allclose(out["reward"].numpy(), fx["reward"], rtol=2e-6)
    print(tag, dtype, "worst rel", worst)


def test_cvrp_greedy_free_running():
    """G7: free-running greedy construction reproduces the reference's tours."""
    fx = gu.load_golden("cvrp_rollout_greedy_n20.npz")
    cfg, P, xy, dem, B, N, M = _cvrp_setup(fx)
    acts = fx["actions"].astype(np.int64)
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], mode="greedy")
    assert np.array_equal(out["actions"].numpy(), acts)
    np.testing.assert_allclose(out["reward"].numpy(), fx["reward"], rtol=2e-6)
    for b in range(B):
        orc.check_feasible(acts[b], dem[b, 1:].numpy())


def _check_named(fx, prefix, named, rtol=2e-3):
    """Compare per-parameter tensors with the fixture's summaries.  Absolute slack is tied to the
    GLOBAL gradient scale (RMS of the largest parameter gradient): parameters whose gradient is
    analytically ~0 (e.g. a bias in front of an instance norm) carry only rounding noise."""
    stride = int(fx["stride"])
    rms = []
    for key in fx.files:
        if key.startswith(prefix + "/norm/"):
            name = key[len(prefix) + 6:]
            rms.append(float(fx[key]) / np.sqrt(named[name].numel()))
    atol = 1e-3 * max(rms)
    n_checked = 0
    for key in fx.files:
        if not key.startswith(prefix + "/"):
            continue
        kind, name = key[len(prefix) + 1:].split("/", 1)
        ref = fx[key]
        got = named[name].detach().numpy().astype(np.float64)
        if kind == "norm":
            assert abs(np.sqrt((got ** 2).sum()) - ref) <= rtol * ref + atol * np.sqrt(got.size), name
        else:
            g = got if kind == "full" else got.reshape(-1)[::stride]
            assert np.abs(g - ref).max() <= rtol * np.abs(ref).max() + atol, name
        n_checked += 1
    assert n_checked > 0


def test_cvrp_train_step():
    """G6: loss, every parameter gradient and the Adam update of one real train() step."""
    fx = gu.load_golden("cvrp_train_n20.npz")
    B, N, M, wseed, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    P = {k: v.requires_grad_(True) for k, v in _weights("cvrp", wseed, mp, 1.0).items()}
    xy = torch.from_numpy(np.concatenate([fx["depot"], fx["loc"]], 1))
    dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), fx["demand"]], 1))
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts)
    np.testing.assert_allclose(out["probs"].detach().numpy(), fx["probs"], rtol=1e-4)
    np.testing.assert_allclose(out["reward"].numpy(), fx["rewards"], rtol=2e-6)
    J = orc.pomo_loss(out["probs"], torch.from_numpy(fx["rewards"]))
    assert abs(J.item() - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    opt = torch.optim.Adam(list(P.values()), lr=1e-4, weight_decay=1e-6)
    w0 = {k: v.detach().clone() for k, v in P.items()}
    J.backward()
    _check_named(fx, "grad", {k: v.grad for k, v in P.items()})
    opt.step()
    # Adam's first step is ~lr*sign(g): compare where the gradient is not rounding noise
    # (biases in front of an instance norm have an analytically zero gradient -> random sign).
    n_checked = 0
    for key in fx.files:
        if key.startswith("delta/full/"):
            name = key[len("delta/full/"):]
            g = fx["grad/full/" + name]
            sig = np.abs(g) > 1e-3 * np.abs(g).max()
            if float(fx["grad/norm/" + name]) / np.sqrt(g.size) < 1e-6 or sig.sum() == 0:
                continue
            got = (P[name].detach() - w0[name]).numpy()
            bad = np.abs(got - fx[key])[sig] > 2e-5
            assert bad.mean() < 0.02, (name, bad.mean())
            n_checked += 1
    assert n_checked >= 10


# ---------------------------------------------------------------- TSP
TSP_TAGS = ["n20", "n50", "greedy_n20"]


def _tsp_setup(fx, dtype=torch.float32):
    B, N, M, wseed, pseed, local_size, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.TSP_MODEL_PARAMS)
    mp["local_size"] = [local_size]
    cfg = orc.ModelCfg.from_model_params(mp, "tsp")
    P = _weights("tsp", wseed, mp, float(fx["gain"]), dtype)
    xy = torch.from_numpy(gu.golden_tsp_problem(pseed, B, N)).to(dtype)
    return cfg, P, xy, B, N, M


@pytest.mark.parametrize("tag", TSP_TAGS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_tsp_decode_probs(tag, dtype):
    fx = gu.load_golden(f"tsp_rollout_{tag}.npz")
    cfg, P, xy, B, N, M = _tsp_setup(fx, dtype)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    out = orc.rollout_tsp(P, cfg, xy, M, starts=acts[0, :, 0], forced=acts, keep_probs=True)
    if dtype == torch.float32:
        np.testing.assert_allclose(out["enc"].numpy(), fx["enc"], rtol=1e-4, atol=2e-5)
    for i, t in enumerate(fx["pf_t"]):
        ref = fx["pf"][i]
        got = out["full_probs"][t - 1].numpy()
        assert np.array_equal(ref == 0, got == 0), f"zero pattern t={t}"
        big = ref > 1e-6
        assert (np.abs(got[big] - ref[big]) / ref[big]).max() < 5e-4, f"t={t}"
    if "sel_prob" in fx.files:
        np.testing.assert_allclose(out["probs"].numpy(), fx["sel_prob"], rtol=5e-4, atol=1e-9)
    np.testing.assert_allclose(out["reward"].numpy(), fx["reward"], rtol=2e-6)


def test_tsp_greedy_free_running():
    fx = gu.load_golden("tsp_rollout_greedy_n20.npz")
    cfg, P, xy, B, N, M = _tsp_setup(fx)
    acts = fx["actions"].astype(np.int64)
    out = orc.rollout_tsp(P, cfg, xy, M, starts=acts[0, :, 0], mode="greedy")
    assert np.array_equal(out["actions"].numpy(), acts)


def test_tsp_train_step():
    fx = gu.load_golden("tsp_train_n20.npz")
    B, N, M, wseed, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.TSP_MODEL_PARAMS)
    cfg = orc.ModelCfg.from_model_params(mp, "tsp")
    P = {k: v.requires_grad_(True) for k, v in _weights("tsp", wseed, mp, 1.0).items()}
    xy = torch.from_numpy(fx["problems"])
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    out = orc.rollout_tsp(P, cfg, xy, M, starts=acts[0, :, 0], forced=acts)
    np.testing.assert_allclose(out["probs"].detach().numpy(), fx["probs"], rtol=1e-4)
    J = orc.pomo_loss(out["probs"], torch.from_numpy(fx["rewards"]), guard_zero=True)
    assert abs(J.item() - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    J.backward()
    _check_named(fx, "grad", {k: v.grad for k, v in P.items()})


# ---------------------------------------------------------------- misc
def test_aug8():
    fx = gu.load_golden("aug8.npz")
    assert np.array_equal(orc.aug8(torch.from_numpy(fx["x"])).numpy(), fx["y"])


def test_vrplib_instance_end_to_end():
    """G8: per-axis scaling, 8-fold aug, greedy construction and rounded unscaled reward on X-n101-k25."""
    import os
    fx = gu.load_golden("cvrp_vrplib_X-n101-k25.npz")
    inst = orc.read_vrp(os.path.join(gu.GOLDEN_DIR, "vrplib", "X", "X-n101-k25.vrp"))
    sxy = orc.vrplib_scale(inst["node_coord"])
    xy8 = orc.aug8(sxy)
    assert np.array_equal(xy8.numpy(), fx["scaled_xy"])
    raw8 = orc.aug8(torch.tensor(inst["node_coord"], dtype=torch.float32)[None])
    assert np.array_equal(raw8.numpy(), fx["unscaled_xy"])
    dem = torch.tensor(inst["demand"] / inst["capacity"], dtype=torch.float32)[None].repeat(8, 1)
    assert np.array_equal(dem.numpy(), fx["demand"])
    mp = dict(gu.CVRP_MODEL_PARAMS)
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    P = _weights("cvrp", int(fx["wseed"]), mp, 1.0)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    enc = orc.encoder_forward(P, cfg, xy8, dem)
    np.testing.assert_allclose(enc.numpy(), fx["enc"], rtol=1e-4, atol=2e-5)
    rew = -orc.route_length(raw8, acts, rounding=True)
    assert np.array_equal(rew.numpy(), fx["reward"])
    assert float(-rew.max()) == float(fx["best_cost"])
    # teacher-forced: the oracle's greedy choice agrees with the reference's at (almost) every step
    out = orc.rollout_cvrp(P, cfg, xy8[:2], dem[:2], 100, starts=acts[0, :, 1], forced=acts[:2], keep_probs=True,
                           enc=enc[:2], max_steps=40)
    agree = []
    for i, p in enumerate(out["full_probs"]):
        agree.append((p.argmax(-1) == acts[:2, :, i + 2]).float().mean().item())
    assert min(agree) > 0.98, agree


# ------------------------------------------------------------------------------------------------------------------
# round 2: decoder internals -- the tolerance north_star states, on the logits themselves
# ------------------------------------------------------------------------------------------------------------------
# north_star: "within 1e-4 rel on logits".  Scores (before the clip, O(1)): |got - ref| <= 1e-4 max(|ref|, 1) on every open
# node.  Clipped logits clip * tanh(s), range +-clip: |got - ref| <= 1e-4 * clip (a bound relative to |logit| itself is
# void where the logit crosses 0 -- two fp32 evaluations of the reference differ by more than that there).
LOGIT_RTOL = 1e-4


def logit_errors(got, ref, open_, scale=None):
    """Worst error of `got` against `ref` over the open nodes, in units of max(|ref|, 1), or of `scale` if given."""
    den = np.maximum(np.abs(ref[open_]), 1.0) if scale is None else scale
    d = np.abs(got[open_] - ref[open_]) / den
    return float(d.max()) if d.size else 0.0


@pytest.mark.parametrize("problem,tag", [("cvrp", "n50"), ("cvrp", "n20k8"), ("cvrp", "n100"), ("tsp", "n50"), ("tsp", "n20")])
def test_oracle_decoder_internals(problem, tag):
    """Pointer score before the penalty, local-policy output, score before the clip and the clipped logits of the
    reference's decoder (tools/make_golden_r02.py hooks), teacher-forced, against the oracle: 1e-4 on open nodes."""
    lg = gu.load_golden(f"r02_{problem}_logits_{tag}.npz")
    fx = gu.load_golden(f"{problem}_rollout_{str(lg['src'])}.npz")
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    if problem == "cvrp":
        cfg, P, xy, dem, B, N, M = _cvrp_setup(fx, torch.float32)
        out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_parts=True)
        t0 = 2
    else:
        cfg, P, xy, B, N, M = _tsp_setup(fx, torch.float32)
        out = orc.rollout_tsp(P, cfg, xy, M, starts=acts[0, :, 0], forced=acts, keep_parts=True)
        t0 = 1
    worst = {}
    for i, t in enumerate(lg["steps"]):
        parts = out["parts"][int(t) - t0]
        ref_logits = lg["logits"][i]
        open_ = np.isfinite(ref_logits)
        s = parts["s"].numpy()
        u = parts["u"].numpy() if "u" in parts else np.zeros_like(s)
        pen = parts["pen"].numpy() if "pen" in parts else np.zeros_like(s)
        got = {"pre_clip": s, "local": u, "score_scaled": s - pen - u / cfg.ensemble_size,
               "logits": cfg.logit_clipping * np.tanh(s)}
        for k, g in got.items():
            e = logit_errors(g, lg[k][i], open_, cfg.logit_clipping if k == "logits" else None)
            worst[k] = max(worst.get(k, 0.0), e)
            assert e <= LOGIT_RTOL, (k, int(t), e)
    print(problem, tag, {k: f"{v:.2e}" for k, v in worst.items()})


# ---------------------------------------------------------------------------------------------------------------------
# config-reachable model variants: fixtures of tools/make_golden_r02b.py
# ---------------------------------------------------------------------------------------------------------------------
def local_only_setup(dtype=torch.float32):
    fx = gu.load_golden("r02_cvrp_local_only.npz")
    B, N, M, wseed, pseed, K = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    mp["local_size"] = [K]
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    w = gu.golden_weights("cvrp", wseed, mp, local=True, gain=float(fx["gain"]))
    pre = "decoder.local_policies.0."
    P = {"local_policy." + k[len(pre):]: torch.from_numpy(v).to(dtype) for k, v in w.items() if k.startswith(pre)}
    depot, loc, demand = gu.golden_cvrp_problem(pseed, B, N, 30.0)
    xy = torch.from_numpy(np.concatenate([depot, loc], 1)).to(dtype)
    dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1)).to(dtype)
    return fx, mp, cfg, P, xy, dem, B, N, M


@pytest.mark.parametrize("tag", ["greedy", "sample"])
def test_oracle_local_only_model(tag):
    """CVRPModel_local (reference CVRPModel.py:78-131): logits of every decode step of the reference's own rollout, its
    greedy tours and the chosen probabilities of its sampled tours."""
    fx, mp, cfg, P, xy, dem, B, N, M = local_only_setup()
    acts = torch.from_numpy(fx[f"{tag}_actions"].astype(np.int64))
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_parts=True, keep_probs=True,
                           local_only=True)
    ref_logits = fx[f"{tag}_logits"]
    assert len(out["parts"]) == ref_logits.shape[0]
    worst = 0.0
    for i in range(ref_logits.shape[0]):
        open_ = np.isfinite(ref_logits[i])
        got = cfg.logit_clipping * np.tanh(out["parts"][i]["s"].numpy())
        e = logit_errors(got, ref_logits[i], open_, cfg.logit_clipping)
        worst = max(worst, e)
        assert e <= LOGIT_RTOL, (i, e)
        assert np.array_equal(np.isfinite(out["full_probs"][i].log().numpy()), open_)       # same mask
    np.testing.assert_allclose(out["reward"].numpy(), fx[f"{tag}_reward"], rtol=1e-5)
    if tag == "greedy":
        free = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], local_only=True)
        assert torch.equal(free["actions"], acts)
    else:
        np.testing.assert_allclose(out["probs"].numpy(), fx["sample_probs"], rtol=5e-4, atol=1e-9)
    print("local-only", tag, f"worst logit error {worst:.2e}")


@pytest.mark.parametrize("problem", ["cvrp", "tsp"])
def test_oracle_euclidean_local_features(problem):
    """model_params['euclidean'] = True (reference models.py:95-125, TSP/models.py:67-75): local-policy output, score before
    the clip and logits at teacher-forced steps."""
    lg = gu.load_golden(f"r02_{problem}_euclidean.npz")
    fx = gu.load_golden(f"{problem}_rollout_n20.npz")
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    if problem == "cvrp":
        cfg, P, xy, dem, B, N, M = _cvrp_setup(fx, torch.float32)
        cfg.euclidean = True
        out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_parts=True)
        t0 = 2
    else:
        cfg, P, xy, B, N, M = _tsp_setup(fx, torch.float32)
        cfg.euclidean = True
        out = orc.rollout_tsp(P, cfg, xy, M, starts=acts[0, :, 0], forced=acts, keep_parts=True)
        t0 = 1
    for i, t in enumerate(lg["steps"]):
        parts = out["parts"][int(t) - t0]
        open_ = np.isfinite(lg["logits"][i])
        s = parts["s"].numpy()
        for k, g in {"pre_clip": s, "local": parts["u"].numpy(), "logits": cfg.logit_clipping * np.tanh(s)}.items():
            e = logit_errors(g, lg[k][i], open_, cfg.logit_clipping if k == "logits" else None)
            assert e <= LOGIT_RTOL, (k, int(t), e)
    # and it is a different model: the polar features give other local scores
    cfg.euclidean = False
    if problem == "cvrp":
        polar = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_parts=True)
    else:
        polar = orc.rollout_tsp(P, cfg, xy, M, starts=acts[0, :, 0], forced=acts, keep_parts=True)
    i, t = 1, int(lg["steps"][1])
    assert np.abs(polar["parts"][t - t0]["u"].numpy() - lg["local"][i]).max() > 1e-2


def ensemble_setup(tag, dtype=torch.float32, requires_grad=False):
    """The r03_cvrp_ensemble.npz configuration `tag` (tools/make_golden_r03.py): oracle config, weights, problem."""
    fx = gu.load_golden("r03_cvrp_ensemble.npz")
    B, N, M, wseed, pseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    mp["local_size"] = [int(k) for k in fx[f"{tag}_sizes"]]
    mp["ensemble_size"] = len(mp["local_size"])
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    P = _weights("cvrp", wseed, mp, float(fx["gain"]), dtype)
    if requires_grad:
        P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    depot, loc, demand = gu.golden_cvrp_problem(pseed, B, N, 30.0)
    xy = torch.from_numpy(np.concatenate([depot, loc], 1)).to(dtype)
    dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1)).to(dtype)
    return fx, mp, cfg, P, xy, dem, B, N, M


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_ensemble_of_local_policies(tag):
    """ensemble_size = 2 (reference models.py:296-298,409-413), local_size [12, 6] and [6, 12]: the members' summed output,
    the score before the clip and the logits at teacher-forced steps of the reference's own greedy tours; then the gradients
    of one REINFORCE step (train.py:112-121) on the reference's sampled tours."""
    fx, mp, cfg, P, xy, dem, B, N, M = ensemble_setup(tag)
    acts = torch.from_numpy(fx[f"{tag}_greedy_actions"].astype(np.int64))
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_parts=True, keep_probs=True)
    for i, t in enumerate(fx[f"{tag}_steps"]):
        parts = out["parts"][int(t) - 2]
        open_ = np.isfinite(fx[f"{tag}_logits"][i])
        s = parts["s"].numpy()
        for k, g in {"pre_clip": s, "local_sum": parts["u"].numpy(), "logits": cfg.logit_clipping * np.tanh(s)}.items():
            e = logit_errors(g, fx[f"{tag}_{k}"][i], open_, cfg.logit_clipping if k == "logits" else None)
            assert e <= LOGIT_RTOL, (k, int(t), e)
    # the reference's greedy choices are the oracle's arg-max at every decoded step
    agree = np.mean([(p.argmax(-1) == acts[:, :, i + 2]).float().mean().item() for i, p in enumerate(out["full_probs"])])
    assert agree == 1.0
    np.testing.assert_allclose(out["reward"].numpy(), fx[f"{tag}_greedy_reward"], rtol=1e-5)
    # ---- one REINFORCE step on the reference's sampled tours: probabilities, loss, every decoder / local gradient
    fx, mp, cfg, P, xy, dem, B, N, M = ensemble_setup(tag, torch.float64, requires_grad=True)
    sacts = torch.from_numpy(fx[f"{tag}_sample_actions"].astype(np.int64))
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=sacts[0, :, 1], forced=sacts)
    np.testing.assert_allclose(out["probs"].detach().numpy(), fx[f"{tag}_sample_probs"], rtol=2e-4, atol=1e-7)
    J = orc.pomo_loss(out["probs"], torch.from_numpy(fx[f"{tag}_sample_reward"]).double())
    assert abs(J.item() - float(fx[f"{tag}_loss"])) <= 1e-4 * max(1.0, abs(float(fx[f"{tag}_loss"])))
    J.backward()
    names = [k[len(f"{tag}_grad_"):] for k in fx.files if k.startswith(f"{tag}_grad_")]
    assert any(n.startswith("decoder.local_policies.1.") for n in names)
    for n in names:
        ref = fx[f"{tag}_grad_{n}"]
        got = P[n].grad.numpy()
        assert np.abs(got - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-6), n


def wide_slots_setup(tag, dtype=torch.float32, requires_grad=False):
    """The r06_cvrp_wide_slots.npz configuration `tag` (tools/make_golden_r06.py): local_size 50 (k50) / 63 (k63) on CVRP-100."""
    fx = gu.load_golden("r06_cvrp_wide_slots.npz")
    B, N, M, wseed, pseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    mp["local_size"] = [int(tag[1:])]
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    P = _weights("cvrp", wseed, mp, float(fx["gain"]), dtype)
    if requires_grad:
        P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    depot, loc, demand = gu.golden_cvrp_problem(pseed, B, N, 50.0)
    xy = torch.from_numpy(np.concatenate([depot, loc], 1)).to(dtype)
    dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1)).to(dtype)
    return fx, mp, cfg, P, xy, dem, B, N, M


@pytest.mark.parametrize("tag", ["k50", "k63"])
def test_oracle_local_size_above_47(tag):
    """local_size 50 / 63 (the reference takes any K: models.py:8-36, 55-120): local-policy output, score before the clip and logits
    at teacher-forced steps of the reference's own greedy tours (early steps with all K neighbours, late ones with K_eff < K), the
    greedy choices, then the gradients of one REINFORCE step on the reference's sampled tours."""
    fx, mp, cfg, P, xy, dem, B, N, M = wide_slots_setup(tag)
    acts = torch.from_numpy(fx[f"{tag}_greedy_actions"].astype(np.int64))
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_parts=True, keep_probs=True)
    for i, t in enumerate(fx[f"{tag}_steps"]):
        parts = out["parts"][int(t) - 2]
        open_ = np.isfinite(fx[f"{tag}_logits"][i])
        s = parts["s"].numpy()
        for k, g in {"pre_clip": s, "local": parts["u"].numpy(), "logits": cfg.logit_clipping * np.tanh(s)}.items():
            e = logit_errors(g, fx[f"{tag}_{k}"][i], open_, cfg.logit_clipping if k == "logits" else None)
            assert e <= LOGIT_RTOL, (k, int(t), e)
    agree = np.mean([(p.argmax(-1) == acts[:, :, i + 2]).float().mean().item() for i, p in enumerate(out["full_probs"])])
    assert agree == 1.0
    np.testing.assert_allclose(out["reward"].numpy(), fx[f"{tag}_greedy_reward"], rtol=1e-5)
    fx, mp, cfg, P, xy, dem, B, N, M = wide_slots_setup(tag, torch.float64, requires_grad=True)
    sacts = torch.from_numpy(fx[f"{tag}_sample_actions"].astype(np.int64))
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=sacts[0, :, 1], forced=sacts)
    np.testing.assert_allclose(out["probs"].detach().numpy(), fx[f"{tag}_sample_probs"], rtol=2e-4, atol=1e-7)
    J = orc.pomo_loss(out["probs"], torch.from_numpy(fx[f"{tag}_sample_reward"]).double())
    assert abs(J.item() - float(fx[f"{tag}_loss"])) <= 1e-4 * max(1.0, abs(float(fx[f"{tag}_loss"])))
    J.backward()
    for n in [k[len(f"{tag}_grad_"):] for k in fx.files if k.startswith(f"{tag}_grad_")]:
        ref = fx[f"{tag}_grad_{n}"]
        got = P[n].grad.numpy()
        assert np.abs(got - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-6), n


# ---------------------------------------------------------------------------------------------------------------------
# the folded decode and the bf16 restatement (round 5): the oracle of the engine's bf16 throughput mode
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("problem,tag", [("cvrp", "n50"), ("cvrp", "n100"), ("tsp", "n20")])
def test_oracle_folded_decode_against_reference_internals(problem, tag):
    """fold_tables + the folded glimpse / pointer (what precision="bf16" is defined on) against the REFERENCE's recorded score
    before the clip and logits at the f32 bar, and against the oracle's unfolded path: the regrouping changes nothing."""
    lg = gu.load_golden(f"r02_{problem}_logits_{tag}.npz")
    fx = gu.load_golden(f"{problem}_rollout_{str(lg['src'])}.npz")
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    if problem == "cvrp":
        cfg, P, xy, dem, B, N, M = _cvrp_setup(fx, torch.float32)
        enc = orc.encoder_forward(P, cfg, xy, dem)
        kw = dict(starts=acts[0, :, 1], forced=acts, enc=enc, keep_parts=True)
        run = lambda **k: orc.rollout_cvrp(P, cfg, xy, dem, M, **kw, **k)
        t0 = 2
    else:
        cfg, P, xy, B, N, M = _tsp_setup(fx, torch.float32)
        enc = orc.encoder_forward(P, cfg, xy)
        kw = dict(starts=acts[0, :, 0], forced=acts, enc=enc, keep_parts=True)
        run = lambda **k: orc.rollout_tsp(P, cfg, xy, M, **kw, **k)
        t0 = 1
    plain, folded = run(), run(tables=orc.fold_tables(P, cfg, enc))
    for i, t in enumerate(lg["steps"]):
        open_ = np.isfinite(lg["logits"][i])
        s = folded["parts"][int(t) - t0]["s"].numpy()
        assert logit_errors(s, lg["pre_clip"][i], open_) <= LOGIT_RTOL
        assert logit_errors(cfg.logit_clipping * np.tanh(s), lg["logits"][i], open_, cfg.logit_clipping) <= LOGIT_RTOL
        assert logit_errors(s, plain["parts"][int(t) - t0]["s"].numpy(), open_) <= 2e-5
    np.testing.assert_allclose(folded["probs"].numpy(), plain["probs"].numpy(), rtol=2e-4)


def test_oracle_bf16_restatement_is_a_value_substitution(monkeypatch):
    """precision="bf16": (i) with the rounding switched off it IS the folded f32 path, values and gradients (the substitution
    plumbing adds nothing of its own); (ii) with it on, the scores move by the size of a bf16 rounding (between 1e-5 and 1e-1 of
    max(|s|, 1)), the gradient stays finite and close to the f32 one; (iii) the rounded values may come from other tables
    (tables_val) without touching the derivative's graph."""
    fx = gu.load_golden("cvrp_rollout_n20.npz")
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    cfg, P, xy, dem, B, N, M = _cvrp_setup(fx, torch.float64)

    def grads(precision, tv=False):
        Pd = {k: v.clone().requires_grad_(k.startswith("decoder.")) for k, v in P.items()}
        enc = orc.encoder_forward(P, cfg, xy, dem).detach().requires_grad_(True)
        t = orc.fold_tables(Pd, cfg, enc)
        tvd = {k: (None if v is None else v.detach().float()) for k, v in t.items()} if tv else None
        out = orc.rollout_cvrp(Pd, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, enc=enc, keep_parts=True, tables=t,
                               tables_val=tvd, precision=precision)
        torch.log(out["probs"]).sum().backward()
        return out, {**{k: v.grad for k, v in Pd.items() if v.grad is not None}, "enc": enc.grad}
    o32, g32 = grads("f32")
    with monkeypatch.context() as m:
        m.setattr(orc, "_bf16", lambda x: x.detach())
        oid, gid = grads("bf16")
    assert torch.allclose(oid["probs"], o32["probs"], rtol=1e-12, atol=0)
    for k in g32:
        assert torch.allclose(gid[k], g32[k], rtol=1e-9, atol=1e-12), k
    ob, gb = grads("bf16")
    ot, gt = grads("bf16", tv=True)
    dev = max(float(((a["s"] - b["s"])[torch.isfinite(a["s"])].abs() / a["s"][torch.isfinite(a["s"])].abs().clamp_min(1.0)).max())
              for a, b in zip(o32["parts"], ob["parts"]))
    assert 1e-5 < dev < 1e-1, dev
    for k in g32:
        assert torch.isfinite(gb[k]).all() and torch.isfinite(gt[k]).all()
        rel = float((gb[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30))
        assert rel < 0.2, (k, rel)
    # float64 tables rounded through f32 (what tables_val carries) round to the same bf16 values except at f32 rounding boundaries
    assert float((ot["probs"] - ob["probs"]).abs().max()) < 1e-2
