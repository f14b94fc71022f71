"""The cooperative lockstep rollout kernel (csrc/elg_fwd.hip::rollout_fwd_coop_kernel: glimpse / pointer / local policy
of a workgroup's trajectories on the matrix cores) against the one-wavefront-per-trajectory kernel it replaces
(debug bit 3) and against the oracle, over the launch shapes that change its control flow: fewer than 16 / more than 16 /
more than 32 trajectories per workgroup (one or two MFMA row tiles, several lockstep groups), ragged last tiles,
single instances.  Same decisions (teacher-forced probabilities 2e-4 rel, greedy tours equal where the margin is not a
rounding tie) and the same saved training rows."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import elg_oracle as orc

pytestmark = pytest.mark.gpu


def _setup(kind, N, B, seed):
    import gpu_common as gc
    from elg_amd import _lib as L
    if kind == "cvrp":
        mp = dict(gu.CVRP_MODEL_PARAMS)
        cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
        P = gc.weights("cvrp", 3, mp, 1.0)
        depot, loc, demand = gu.golden_cvrp_problem(seed, B, N, 30.0 if N <= 30 else 50.0)
        xy = torch.from_numpy(np.concatenate([depot, loc], 1))
        dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1))
        enc = orc.encoder_forward(P, cfg, xy, dem)
        prob = gc.make_problem(xy, dem, L.PROBLEM_CVRP)
        pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_CVRP)
        return P, cfg, xy, dem, enc, prob, pol
    mp = dict(gu.TSP_MODEL_PARAMS)
    cfg = orc.ModelCfg.from_model_params(mp, "tsp")
    P = gc.weights("tsp", 4, mp, 1.0)
    xy = torch.from_numpy(gu.golden_tsp_problem(seed, B, N))
    enc = orc.encoder_forward(P, cfg, xy)
    prob = gc.make_problem(xy, None, L.PROBLEM_TSP)
    pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_TSP)
    return P, cfg, xy, None, enc, prob, pol


@pytest.mark.parametrize("kind,N,B,M,tiles", [("cvrp", 100, 2, 40, 1), ("cvrp", 100, 1, 13, 1), ("cvrp", 100, 3, 50, 2),
                                              ("cvrp", 100, 2, 37, 4), ("cvrp", 63, 2, 17, 1), ("cvrp", 111, 1, 20, 2),
                                              ("tsp", 100, 2, 40, 1), ("tsp", 100, 1, 9, 1), ("tsp", 30, 3, 30, 2)])
def test_coop_matches_per_wave_kernel(kind, N, B, M, tiles):
    from elg_amd import _lib as L, engine as eng
    P, cfg, xy, dem, enc, prob, pol = _setup(kind, N, B, 900 + N + M)
    off = 1 if kind == "cvrp" else 0
    starts = torch.randperm(N, generator=torch.Generator().manual_seed(M))[:M] + off
    geom = (8, tiles, 1)
    a = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=11, geometry=geom)
    T = int(a.tlen.max().item())
    acts = a.actions[:, :, :T].contiguous()
    # same sampled tours teacher-forced through both kernels: probabilities, rewards, step counts
    c = eng.rollout_forward(prob, pol, M, starts, L.MODE_FORCED, forced=acts, geometry=geom, dump_T=min(T, 12))
    w = eng.rollout_forward(prob, pol, M, starts, L.MODE_FORCED, forced=acts, geometry=geom, dump_T=min(T, 12), variant=1)
    assert torch.equal(c.actions, w.actions) and torch.equal(c.tlen, w.tlen) and torch.equal(a.tlen, c.tlen)
    np.testing.assert_allclose(c.probs.cpu().numpy(), w.probs.cpu().numpy(), rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(a.probs.cpu().numpy(), c.probs.cpu().numpy(), rtol=1e-6, atol=0)     # sampled = forced replay
    np.testing.assert_allclose(c.reward.cpu().numpy(), w.reward.cpu().numpy(), rtol=1e-6)
    fc, fw = c.full_probs.cpu().numpy(), w.full_probs.cpu().numpy()
    assert np.array_equal(fc == 0, fw == 0)                                                        # identical masks
    np.testing.assert_allclose(fc, fw, rtol=2e-4, atol=1e-12)
    # and against the oracle
    if kind == "cvrp":
        out = orc.rollout_cvrp(P, cfg, xy, dem, M, starts=starts, forced=acts.cpu().long(), enc=enc)
    else:
        out = orc.rollout_tsp(P, cfg, xy, M, starts=starts, forced=acts.cpu().long(), enc=enc)
    np.testing.assert_allclose(c.probs[:, :T].cpu().numpy(), out["probs"].numpy(), rtol=5e-4, atol=1e-9)
    np.testing.assert_allclose(c.reward.cpu().numpy(), out["reward"].numpy(), rtol=1e-5)
    # greedy: identical tours except where a rounding tie flips a choice
    g1 = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY, geometry=geom)
    g2 = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY, geometry=geom, variant=1)
    same = (g1.actions == g2.actions).all(-1).float().mean().item()
    assert same >= 0.9, same


@pytest.mark.parametrize("kind", ["cvrp", "tsp"])
def test_coop_training_rows_match_per_wave_kernel(kind):
    """TRAIN variant: every row the backward consumes (softmax Jacobian rows, q, o, load, slot codes, slot features) is
    what the per-wavefront kernel saves.  (The glimpse weights are not saved by the cooperative kernel: the backward
    rebuilds them from q, K and the saved mask words -- covered by the gradient equality test below.)"""
    from elg_amd import _lib as L, engine as eng
    N, B, M = 100, 2, 37
    P, cfg, xy, dem, enc, prob, pol = _setup(kind, N, B, 77)
    off = 1 if kind == "cvrp" else 0
    starts = torch.arange(M) + off
    a = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=5, train=True)
    T = int(a.tlen.max().item())
    acts = a.actions[:, :, :T].contiguous()
    R = T * M
    keep = {}
    for tag, dbg in (("coop", 0), ("wave", 1)):
        r = eng.rollout_forward(prob, pol, M, starts, L.MODE_FORCED, forced=acts, train=True, variant=dbg)
        rows = r.rows
        keep[tag] = dict(PC=rows.PC[:, :R].clone(), Csel=rows.Csel[:, :R].clone(),
                         Q=rows.Q[:, :R].clone(), O=rows.O[:, :R].clone(), Load=rows.Load[:, :R].clone(),
                         Slot=rows.Slot[:, :R].clone(), F=rows.F[:, :R].clone(), tlen=r.tlen.clone())
    assert torch.equal(keep["coop"]["tlen"], keep["wave"]["tlen"])
    # rows of decoded steps only (both kernels leave the others untouched)
    tt = torch.arange(T, device="cuda:0")[None, :, None]
    t0 = 1 if kind == "tsp" else 2
    valid = ((tt >= t0) & (tt < keep["coop"]["tlen"][:, None, :])).reshape(B, R)
    assert torch.equal(keep["coop"]["Slot"][valid], keep["wave"]["Slot"][valid])
    assert torch.equal(keep["coop"]["F"][valid], keep["wave"]["F"][valid])
    assert torch.equal(keep["coop"]["Q"][valid], keep["wave"]["Q"][valid])
    assert torch.equal(keep["coop"]["Load"][valid], keep["wave"]["Load"][valid])
    for k, tol in (("O", 2e-5), ("PC", 3e-4), ("Csel", 3e-4)):
        x, y = keep["coop"][k][valid].cpu().numpy(), keep["wave"][k][valid].cpu().numpy()
        np.testing.assert_allclose(x, y, rtol=tol, atol=tol * np.abs(y).max())


@pytest.mark.parametrize("recompute", [False, True], ids=["stored_weights", "mask_recompute"])
@pytest.mark.parametrize("kind,N", [("cvrp", 110), ("tsp", 108), ("cvrp", 100)])
def test_saved_rows_backward_equals_replay_backward(kind, N, recompute, monkeypatch):
    """Gradients w.r.t. every folded table through the rows saved by the cooperative training forward (MFMA glimpse /
    local-policy backward kernels) = gradients through the replay kernels, also for 104 < N1 <= 112 where the
    per-wavefront kernels read their tables from L2."""
    from elg_amd import _lib as L, engine as eng
    # stored weights: the per-wavefront forward (variant 1; what 112 < N1 <= 128 uses) saves the glimpse weights; the
    # cooperative forward saves mask rows + normalisers and the backward recomputes the weights
    dbg = 0 if recompute else 1
    B, M = 2, 23
    P, cfg, xy, dem, enc, prob, pol = _setup(kind, N, B, 5)
    off = 1 if kind == "cvrp" else 0
    starts = torch.arange(M) + off
    a = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=2)
    T = int(a.tlen.max().item())
    acts = a.actions[:, :, :T].contiguous()
    torch.manual_seed(0)
    Wt = torch.randn(B, T, M, device="cuda:0")
    grads = {}
    for tag, train in (("rows", True), ("replay", False)):
        tabs = {k: (v.detach().clone().requires_grad_(True) if v is not None else None) for k, v in pol.tables.items()}
        loc = pol.loc.detach().clone().requires_grad_(True)
        p2 = eng.Policy(tabs, loc, pol.K, pol.xi, pol.clip, pol.inv_ens, pol.has_local, pol.has_penalty)
        res = eng.rollout_forward(prob, p2, M, starts, L.MODE_FORCED, forced=acts, train=train, variant=dbg if train else 0)
        if train:
            assert res.rows.use_mask == recompute
        probs = eng.chosen_probs(prob, p2, M, res, T)
        (probs * Wt).sum().backward()
        grads[tag] = {k: v.grad.clone() for k, v in tabs.items() if v is not None}
        grads[tag]["loc"] = loc.grad.clone()
    for k, r in grads["replay"].items():
        g = grads["rows"][k]
        err = (g - r).abs().max().item()
        assert err <= 2e-4 * r.abs().max().item() + 1e-6, (k, err, r.abs().max().item())


@pytest.mark.parametrize("kind,N,B,M", [("cvrp", 100, 3, 50), ("tsp", 50, 2, 40)])
def test_lean_production_instantiation_equals_the_generic_one(kind, N, B, M):
    """rollout_fwd_coop_kernel<..., LEAN>: the launch a training step / an evaluation makes (own Philox stream, all outputs, the
    reference's default model flags) runs an instantiation with every test branch (teacher forcing, external uniforms,
    probability dumps, ablation flags) compiled out.  Same seed through it and through the generic instantiation (forced by a
    one-step probability dump): identical tours, chosen probabilities, rewards, and identical saved training rows."""
    from elg_amd import _lib as L, engine as eng
    P, cfg, xy, dem, enc, prob, pol = _setup(kind, N, B, 4200 + N)
    off = 1 if kind == "cvrp" else 0
    starts = torch.randperm(N, generator=torch.Generator().manual_seed(M))[:M] + off
    for train in (False, True):
        lean = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=77, train=train)
        rows_lean = None
        if train:
            r = lean.rows
            rows_lean = {k: getattr(r, k).clone() for k in ("PC", "Csel", "Q", "O", "Lse", "Mask", "Slot", "F", "Load")}
        gen = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=77, train=train, dump_T=1)
        assert torch.equal(lean.actions, gen.actions) and torch.equal(lean.tlen, gen.tlen)
        assert torch.equal(lean.probs, gen.probs) and torch.equal(lean.reward, gen.reward)
        if train:
            T = int(gen.tlen.max())
            live = T * M                                            # rows of the steps that were decoded (time-major)
            for k, v in rows_lean.items():
                if k == "Load" and kind == "tsp":
                    continue                                        # (a TSP trajectory has no load; the backward never reads the row)
                assert torch.equal(v[:, :live], getattr(gen.rows, k)[:, :live]), k
    g1 = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY)
    g2 = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY, dump_T=1)
    assert torch.equal(g1.actions, g2.actions) and torch.equal(g1.reward, g2.reward)


@pytest.mark.parametrize("kind,N,B,M,tiles", [("cvrp", 100, 3, 100, 4), ("cvrp", 100, 2, 100, 2), ("cvrp", 50, 3, 50, 1), ("tsp", 100, 2, 100, 4),
                                              ("cvrp", 20, 4, 7, 1)])
def test_split_group_kernel_is_bit_identical(kind, N, B, M, tiles):
    """rollout_fwd_coop2_kernel (elg_rollout_args.variant = 4; round 6): the cooperative kernel as two independent 4-wave groups per
    workgroup, each with its own tile of <= 16 trajectories and its own LDS-counter barriers, so that the two waves of a SIMD sit in
    different phases.  Every product is formed by the same instruction sequence on the same operands: sampled
    tours, chosen probabilities, rewards and the saved training rows are equal to the lockstep kernel's bit for bit, in both
    arithmetic modes, with one or several tiles per group and with a group that has no tile (M = 7)."""
    from elg_amd import _lib as L, engine as eng
    P, cfg, xy, dem, enc, prob, pol = _setup(kind, N, B, 5100 + N)
    off = 1 if kind == "cvrp" else 0
    starts = torch.randperm(N, generator=torch.Generator().manual_seed(M))[:M] + off
    geom = (8, tiles, 1)
    for precision in (0, 1):
        for train in (False, True):
            ref = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=91, train=train, geometry=geom, precision=precision)
            assert ref.kernel_id == L.KERNEL_COOP
            rows_ref = None
            if train:
                rows_ref = {k: getattr(ref.rows, k).clone() for k in ("PC", "Csel", "Q", "O", "Lse", "Mask", "Slot", "F", "Load")}
            for variant, kid in ((4, L.KERNEL_COOP_SPLIT), (5, L.KERNEL_COOP_WIDE)):
                got = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=91, train=train, geometry=geom, precision=precision,
                                          variant=variant)
                assert got.kernel_id == kid
                assert torch.equal(ref.actions, got.actions) and torch.equal(ref.tlen, got.tlen), variant
                assert torch.equal(ref.probs, got.probs) and torch.equal(ref.reward, got.reward), variant
                if train:
                    live = int(got.tlen.max()) * M
                    for k, v in rows_ref.items():
                        if k == "Load" and kind == "tsp":
                            continue
                        assert torch.equal(v[:, :live], getattr(got.rows, k)[:, :live]), (k, precision, variant)
        g1 = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY, geometry=geom, precision=precision)
        for variant in (4, 5):
            g2 = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY, geometry=geom, precision=precision, variant=variant)
            assert torch.equal(g1.actions, g2.actions) and torch.equal(g1.reward, g2.reward), variant


@pytest.mark.parametrize("kind,N,B,M", [("cvrp", 4, 1, 1), ("cvrp", 5, 3, 4), ("tsp", 4, 1, 4), ("tsp", 6, 2, 1), ("cvrp", 10, 1, 10)])
def test_smallest_shapes_all_cooperative_forms(kind, N, B, M):
    """Edge shapes: the smallest instances the cooperative kernels take (N1 >= 4), one trajectory, one instance -- the lockstep
    kernel, both split-group forms and the one-wavefront kernel make the same greedy tours with the same rewards, a sampled rollout
    equals its forced replay, and every tour is feasible (each customer once; TSP: a permutation)."""
    from elg_amd import _lib as L, engine as eng
    P, cfg, xy, dem, enc, prob, pol = _setup(kind, N, B, 7700 + 10 * N + M)
    if pol.K > N - 1:
        pol.K = N - 1                                           # (local_size cannot exceed the customers there are)
    off = 1 if kind == "cvrp" else 0
    starts = torch.randperm(N, generator=torch.Generator().manual_seed(M))[:M] + off
    ref = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY, variant=1)
    assert ref.kernel_id == L.KERNEL_WAVE
    for variant, kid in ((0, L.KERNEL_COOP), (4, L.KERNEL_COOP_SPLIT), (5, L.KERNEL_COOP_WIDE)):
        g = eng.rollout_forward(prob, pol, M, starts, L.MODE_GREEDY, variant=variant)
        assert g.kernel_id == kid
        assert torch.equal(g.tlen, ref.tlen), variant
        T = int(ref.tlen.max())
        assert torch.equal(g.actions[:, :, :T], ref.actions[:, :, :T]), variant
        np.testing.assert_allclose(g.reward.cpu().numpy(), ref.reward.cpu().numpy(), rtol=1e-6)
        s = eng.rollout_forward(prob, pol, M, starts, L.MODE_SAMPLE, seed=3, variant=variant)
        f = eng.rollout_forward(prob, pol, M, starts, L.MODE_FORCED, forced=s.actions, variant=variant)
        np.testing.assert_allclose(s.probs.cpu().numpy(), f.probs.cpu().numpy(), rtol=1e-6, atol=0)
        acts = s.actions.cpu().numpy()
        for b in range(B):
            for m in range(M):
                tour = acts[b, m, :int(s.tlen[b, m])]
                cust = tour[tour > 0] if kind == "cvrp" else tour
                assert sorted(cust.tolist()) == list(range(off, N + off)), (variant, b, m, tour)
