"""GPU parity of the forward construction kernel (through the C ABI) against the golden vectors of
the reference and against the oracle.  Bar: env masks / indices bit-exact, probabilities within
5e-4 relative (= 1e-4 relative on the clipped logits, whose range is +-50), rewards 1e-5."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import elg_oracle as orc

pytestmark = pytest.mark.gpu

PROB_RTOL = 5e-4


def _imports():
    import gpu_common as gc
    from elg_amd import _lib as L
    from elg_amd import engine as eng
    return gc, L, eng


def test_library_loads_on_gpu():
    gc, L, eng = _imports()
    assert torch.cuda.is_available()
    assert b"gfx950" in L.lib().elg_version()


def test_aug8_and_dist_and_route_length():
    gc, L, eng = _imports()
    fx = gu.load_golden("aug8.npz")
    out = eng.aug8(torch.from_numpy(fx["x"]).to(gc.DEV))
    assert np.array_equal(out.cpu().numpy(), fx["y"])
    xy = torch.rand(3, 37, 2)
    d = eng.dist_matrix(xy.to(gc.DEV)).cpu()
    assert torch.equal(d, orc.dist_matrix(xy))            # every op rounded once, correctly rounded sqrt: bit-exact
    tour = torch.stack([torch.stack([torch.randperm(37) for _ in range(5)]) for _ in range(3)])
    got = eng.route_length(xy.to(gc.DEV), tour.to(gc.DEV)).cpu()
    np.testing.assert_allclose(got.numpy(), orc.route_length(xy, tour).numpy(), rtol=1e-6)


def test_nbr_tables():
    gc, L, eng = _imports()
    torch.manual_seed(0)
    for N in (21, 101, 200, 500):
        xy = torch.rand(2, N, 2)
        xy[0, 5] = xy[0, 3]          # exact tie: order must fall back to the node index
        nb = eng.nbr_tables(xy.to(gc.DEV))
        d = orc.dist_matrix(xy)
        order = torch.argsort(d, dim=-1, stable=True)
        got_idx = nb.idx.cpu().long()
        got_d = nb.dist.cpu()
        # distances ascending and consistent with the indices
        assert (got_d[:, :, 1:] >= got_d[:, :, :-1]).all()
        # bit-exact distances (a 1-ulp difference reorders near-equidistant neighbours and changes a k-NN set), hence the
        # reference's order exactly: (distance, node index)
        assert torch.equal(torch.gather(d, 2, got_idx), got_d)
        assert torch.equal(got_idx, order)
        th = orc.make_theta_fn(xy)(torch.arange(N)[None].expand(2, N))
        np.testing.assert_allclose(nb.theta.cpu().numpy(), torch.gather(th, 2, got_idx).numpy(), rtol=1e-5, atol=2e-6)


CVRP_TAGS = ["n20", "n20k8", "n50", "n100", "greedy_n20"]


@pytest.mark.parametrize("tag", CVRP_TAGS)
@pytest.mark.parametrize("geom", ["auto", "w8_global"])
def test_cvrp_teacher_forced_probs(tag, geom):
    """Teacher-forced with the reference's recorded actions: whole probability rows, chosen probs, reward,
    step counts.  Compared with the golden vectors (reference) at the stored steps and with the oracle at all."""
    gc, L, eng = _imports()
    fx, cfg, P, xy, dem, B, N, M = gc.cvrp_fixture(tag)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    T = acts.shape[2]
    enc = orc.encoder_forward(P, cfg, xy, dem)
    prob = gc.make_problem(xy, dem, L.PROBLEM_CVRP)
    pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_CVRP)
    geometry = None if geom == "auto" else (8, 2, 0)
    res = eng.rollout_forward(prob, pol, M, acts[0, :, 1], L.MODE_FORCED, forced=acts, dump_T=T, geometry=geometry)
    torch.cuda.synchronize()
    assert np.array_equal(res.actions[:, :, :T].cpu().numpy(), acts.numpy())
    assert (res.tlen.cpu() <= T).all() and res.tlen.max().item() == T
    full = res.full_probs.cpu().numpy()                 # (B,M,T,N1)
    worst = 0.0
    for i, t in enumerate(fx["pf_t"]):
        ref = fx["pf"][i]
        got = full[:, :, t, :]
        live = fx["finished"][t - 1] == 0                # finished rows are not decoded by the engine
        gc.assert_same_mask(got[live], ref[live], f"t={t}")
        e = gc.rel_err_probs(got[live], ref[live])
        worst = max(worst, e)
        assert e < PROB_RTOL, f"t={t}: rel err {e}"
    if "sel_prob" in fx.files:
        np.testing.assert_allclose(res.probs[:, :T].cpu().numpy(), fx["sel_prob"], rtol=PROB_RTOL, atol=1e-9)
    np.testing.assert_allclose(res.reward.cpu().numpy(), fx["reward"], rtol=1e-5)
    # oracle at every step
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=acts[0, :, 1], forced=acts, keep_probs=True, enc=enc)
    for t in range(2, T):
        ref = out["full_probs"][t - 2].numpy()
        live = fx["finished"][t - 1] == 0
        e = gc.rel_err_probs(full[:, :, t, :][live], ref[live])
        assert e < PROB_RTOL, f"oracle t={t}: {e}"
    print(tag, geom, "worst rel err vs reference", worst)


def test_cvrp_greedy_free_running():
    gc, L, eng = _imports()
    fx, cfg, P, xy, dem, B, N, M = gc.cvrp_fixture("greedy_n20")
    acts = fx["actions"].astype(np.int64)
    T = acts.shape[2]
    enc = orc.encoder_forward(P, cfg, xy, dem)
    prob = gc.make_problem(xy, dem, L.PROBLEM_CVRP)
    pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_CVRP)
    res = eng.rollout_forward(prob, pol, M, torch.from_numpy(acts[0, :, 1]), L.MODE_GREEDY)
    assert res.tlen.max().item() == T
    assert np.array_equal(res.actions[:, :, :T].cpu().numpy(), acts)
    np.testing.assert_allclose(res.reward.cpu().numpy(), fx["reward"], rtol=1e-5)


def test_cvrp_sampling_self_consistent():
    """Free-running sampling: the engine's own actions, replayed by the oracle, give the same chosen
    probabilities and rewards; tours are feasible; no zero-probability node is ever drawn."""
    gc, L, eng = _imports()
    fx, cfg, P, xy, dem, B, N, M = gc.cvrp_fixture("n50")
    enc = orc.encoder_forward(P, cfg, xy, dem)
    prob = gc.make_problem(xy, dem, L.PROBLEM_CVRP)
    pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_CVRP)
    starts = torch.randperm(N)[:M]
    res = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=1234)
    T = int(res.tlen.max().item())
    acts = res.actions[:, :, :T].cpu().long()
    for b in range(B):
        orc.check_feasible(acts[b].numpy(), dem[b, 1:].numpy())
    out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=starts, forced=acts, enc=enc)
    assert out["actions"].shape[2] == T
    got = res.probs[:, :T].cpu().numpy()
    assert (got > 0).all()
    np.testing.assert_allclose(got, out["probs"].numpy(), rtol=PROB_RTOL, atol=1e-9)
    np.testing.assert_allclose(res.reward.cpu().numpy(), out["reward"].numpy(), rtol=1e-5)
    # a different seed gives different tours, the same seed the same tours
    res2 = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=1234)
    res3 = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=99)
    assert torch.equal(res.actions, res2.actions)
    assert not torch.equal(res.actions, res3.actions)


TSP_TAGS = ["n20", "n50", "greedy_n20"]


@pytest.mark.parametrize("tag", TSP_TAGS)
def test_tsp_teacher_forced_probs(tag):
    gc, L, eng = _imports()
    fx, cfg, P, xy, B, N, M = gc.tsp_fixture(tag)
    acts = torch.from_numpy(fx["actions"].astype(np.int64))
    T = acts.shape[2]
    enc = orc.encoder_forward(P, cfg, xy)
    prob = gc.make_problem(xy, None, L.PROBLEM_TSP)
    pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_TSP)
    res = eng.rollout_forward(prob, pol, M, acts[0, :, 0], L.MODE_FORCED, forced=acts, dump_T=T)
    full = res.full_probs.cpu().numpy()
    for i, t in enumerate(fx["pf_t"]):
        ref = fx["pf"][i]
        got = full[:, :, t, :]
        gc.assert_same_mask(got, ref, f"t={t}")
        e = gc.rel_err_probs(got, ref)
        assert e < PROB_RTOL, f"t={t}: {e}"
    if "sel_prob" in fx.files:
        np.testing.assert_allclose(res.probs[:, :T].cpu().numpy(), fx["sel_prob"], rtol=PROB_RTOL, atol=1e-9)
    np.testing.assert_allclose(res.reward.cpu().numpy(), fx["reward"], rtol=1e-5)
    assert (res.tlen.cpu() == N).all()


def test_tsp_greedy_free_running():
    gc, L, eng = _imports()
    fx, cfg, P, xy, B, N, M = gc.tsp_fixture("greedy_n20")
    acts = fx["actions"].astype(np.int64)
    enc = orc.encoder_forward(P, cfg, xy)
    prob = gc.make_problem(xy, None, L.PROBLEM_TSP)
    pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_TSP)
    res = eng.rollout_forward(prob, pol, M, torch.from_numpy(acts[0, :, 0]), L.MODE_GREEDY)
    assert np.array_equal(res.actions[:, :, :N].cpu().numpy(), acts)
    np.testing.assert_allclose(res.reward.cpu().numpy(), fx["reward"], rtol=1e-5)
