"""Training-step glue kernels (csrc/elg_train.hip) against the plain torch formulas they replace:
 * elg_pomo_loss   -- reference CVRP/train.py:112-121 / TSP/train.py:107-118 (value 1e-5 rel, gradient 1e-5 rel)
 * elg_rows_prep   -- the cotangent rows of the decoder backward (bit-exact: same fp32 operations per element)
 * elg_adam_step   -- torch.optim.Adam(weight_decay) over several steps (1e-6 abs on O(1) weights), checkpoints."""
import numpy as np
import pytest
import torch

import gpu_common as gpc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_loss(probs, rewards, scale_norm, guard, with_scale=False):
    adv = rewards - rewards.mean(dim=1)[:, None]
    J = -adv * probs.log().sum(dim=1)
    if scale_norm:
        nf = adv.max(dim=1)[0][:, None]
        if not guard or bool((nf != 0.).all()):
            J = J / nf
    # the mean cancels heavily (advantages sum to zero): errors are judged against the size of the terms
    return (J.mean(), J.abs().mean().item()) if with_scale else J.mean()


@pytest.mark.parametrize("B,T,M,scale,guard", [(4, 30, 20, True, False), (3, 70, 300, True, True), (2, 11, 7, False, False)])
def test_pomo_loss_matches_formula(B, T, M, scale, guard):
    from elg_amd import engine as eng
    g = torch.Generator().manual_seed(B + T)
    big = torch.rand(B, T + 5, M, generator=g).clamp_min(1e-3).to(DEV)
    probs = big[:, :T, :].clone().requires_grad_(True)                  # also exercised strided below
    rew = -torch.rand(B, M, generator=g).to(DEV) * 10
    ref, mag = _torch_loss(probs.double(), rew.double(), scale, guard, True)
    gref, = torch.autograd.grad(ref, probs)
    J = eng.pomo_loss(probs, rew, scale, guard)
    got, = torch.autograd.grad(J, probs)
    assert abs(J.item() - ref.item()) <= 1e-5 * mag
    np.testing.assert_allclose(got.cpu().numpy(), gref.cpu().numpy(), rtol=2e-5,
                               atol=1e-6 * gref.abs().max().item())   # advantage = r - mean cancels in fp32
    Js = eng.pomo_loss(big[:, :T, :], rew, scale, guard)                # non-contiguous view (B stride != T*M)
    assert abs(Js.item() - ref.item()) <= 1e-5 * mag


def test_pomo_loss_tsp_zero_normaliser_guard():
    """TSP/train.py:113-116: if any instance has max advantage 0 (all its tours equal), nothing is scaled."""
    from elg_amd import engine as eng
    g = torch.Generator().manual_seed(1)
    probs = torch.rand(3, 9, 6, generator=g).clamp_min(1e-2).to(DEV)
    rew = -torch.rand(3, 6, generator=g).to(DEV)
    rew[1] = -2.0
    ref, mag = _torch_loss(probs.double(), rew.double(), True, True, True)
    J = eng.pomo_loss(probs, rew, True, True)
    assert torch.isfinite(J) and abs(J.item() - ref.item()) <= 1e-5 * mag


@pytest.mark.parametrize("tsp,recompute,mode,N1", [(False, False, 0, 23), (True, False, 0, 23), (False, True, 0, 23),
                                                    (False, True, 0, 101), (False, True, 1, 101), (True, True, 1, 23),
                                                    (False, True, 2, 101), (True, True, 2, 101), (False, True, 1, 37), (False, True, 2, 112),
                                                    (True, True, 2, 65)])
def test_decoder_bwd_matches_torch(tsp, recompute, mode, N1):
    """elg_decoder_bwd on synthetic saved rows against the dense torch algebra it stands for (include/elg_hip.h):
    dl = w (Csel [n == a] - PC), dO = dl PK, dPK = dl^T O, dpb = sum dl, dU = dl[slot] / ens, glimpse backward
    (dK, dV, dQ) from dO, and dQ1 / dQ2 / dwl = the query-gather backward of dQ."""
    import ctypes as C
    from elg_amd import _lib as L, engine as eng
    B, T, M, Tcap, H = 2, 9, 5, 12, 8
    if N1 > 64:
        T, M, Tcap = 12, 19, 14                  # several 16-row tiles per (instance, head), a ragged last one
    R, Rcap = T * M, Tcap * M
    g = torch.Generator().manual_seed(3)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(DEV)
    gp, pv = rnd(B, T, M), torch.rand(B, T, M, generator=g).to(DEV)
    tlen = torch.randint(4, T + 1, (B, M), generator=g, dtype=torch.int32).to(DEV)
    acts = torch.randint(0, N1, (B, M, Tcap), generator=g, dtype=torch.int32).to(DEV)
    PC, Csel, Q, O = rnd(B, Rcap, N1), rnd(B, Rcap), rnd(B, Rcap, 128), rnd(B, Rcap, 128)
    Load = torch.rand(B, Rcap, generator=g).to(DEV)
    Slot = torch.randint(-1, N1, (B, Rcap, 48), generator=g, dtype=torch.int32).to(DEV)
    K, V, PK = rnd(B, N1, 128), rnd(B, N1, 128), rnd(B, N1, 128)
    closed = (torch.rand(B, Rcap, N1, generator=g) < 0.3).to(DEV)
    closed[:, :, 1] = False
    # glimpse weights of the rows: softmax over the open nodes of q.K / 4 per head
    S = torch.einsum("brhd,bnhd->bhrn", Q.view(B, Rcap, H, 16), K.view(B, N1, H, 16)) / 4
    S = S.masked_fill(closed[:, None], float("-inf"))
    A = torch.softmax(S, dim=-1).contiguous()
    bits = torch.zeros(B, Rcap, 2, dtype=torch.int64, device=DEV)
    for n in range(N1):
        bits[:, :, n // 64] |= closed[:, :, n].long() << (n % 64)
    if N1 < 64:
        bits[:, :, 0] |= (-1 << N1)                     # nodes past N1 closed
        bits[:, :, 1] = -1
    else:
        bits[:, :, 1] |= (-1 << (N1 - 64))
    lse = (torch.logsumexp(S, dim=-1) * 1.4426950408889634).permute(0, 2, 1).contiguous()        # (B,Rcap,8), log2 units
    t0, inv = (1 if tsp else 2), 0.5
    nt = 5 if tsp else 4
    flat = torch.zeros(nt * B * N1 * 128 + B * N1 + 128, device=DEV)
    blk = B * N1 * 128
    dK, dV, dPK, dQ1 = (flat[i * blk:(i + 1) * blk].view(B, N1, 128) for i in range(4))
    dQ2 = flat[4 * blk:5 * blk].view(B, N1, 128) if tsp else None
    dpb = flat[nt * blk:nt * blk + B * N1].view(B, N1)
    dwl = flat[nt * blk + B * N1:]
    rowDU = torch.full((B, R, 48), float("nan"), device=DEV)
    dO = torch.zeros(B, R, 128, device=DEV)             # scratch: only the rows of the live decode steps are written
    ixP = torch.empty(B, R, dtype=torch.int32, device=DEV)
    ixF = torch.empty(B, R, dtype=torch.int32, device=DEV)
    rowW = torch.empty(B, R, 4, device=DEV)
    a = L.DecoderBwdArgs()
    a.problem, a.B, a.M, a.N1, a.T, a.Tcap_actions = (L.PROBLEM_TSP if tsp else L.PROBLEM_CVRP), B, M, N1, T, Tcap
    a.first_decode_step, a.inv_ens, a.Rcap = t0, inv, Rcap
    p = eng._ptr
    a.gprob, a.pval, a.tlen, a.actions = p(gp), p(pv), p(tlen), p(acts)
    a.trPC, a.trCsel, a.trQ, a.trO, a.trLoad, a.trSlot = p(PC), p(Csel), p(Q), p(O), (None if tsp else p(Load)), p(Slot)
    a.trA, a.trMask, a.trLse = (None, p(bits), p(lse)) if recompute else (p(A), None, None)
    a.Kmat, a.Vmat, a.PK = p(K), p(V), p(PK)
    a.dK, a.dV, a.dPK, a.dpb, a.dQ1, a.dQ2, a.dwl = p(dK), p(dV), p(dPK), p(dpb), p(dQ1), p(dQ2), (None if tsp else p(dwl))
    a.rowDU, a.dO, a.idx_prev, a.idx_first, a.rowW = p(rowDU), p(dO), p(ixP), p(ixF), p(rowW)
    a.mfma_mode = mode                      # 0: f32 MFMAs; 1 .. 3: split-bf16 products of the glimpse backward
    L.check(L.lib().elg_decoder_bwd(C.byref(a), eng._stream()), "elg_decoder_bwd")
    # ---- the dense algebra in torch (double precision)
    fl = acts[:, :, :T].long()
    tt = torch.arange(T, device=DEV)[None, :, None]
    valid = (tt >= t0) & (tt < tlen[:, None, :])
    W = (gp * pv * valid).reshape(B, R).double()
    sel = fl.permute(0, 2, 1).reshape(B, R)
    dl = PC[:, :R].double() * (-W)[:, :, None]
    dl.scatter_add_(2, sel[:, :, None], (W * Csel[:, :R].double())[:, :, None])
    slot = Slot[:, :R].long()
    refU = torch.gather(dl, 2, slot.clamp(min=0)) * (slot >= 0) * inv
    Kd, Vd, PKd, Qd, Od, Ad = K.double(), V.double(), PK.double(), Q[:, :R].double(), O[:, :R].double(), A[:, :, :R].double()
    dOr = dl @ PKd                                                       # (B,R,128)
    dPKr = dl.transpose(1, 2) @ Od
    dOh = dOr.view(B, R, H, 16).permute(0, 2, 1, 3)
    dA = dOh @ Vd.view(B, N1, H, 16).permute(0, 2, 3, 1)                 # (B,H,R,N1)
    dotO = (dOh * Od.view(B, R, H, 16).permute(0, 2, 1, 3)).sum(-1, keepdim=True)
    dS = Ad * (dA - dotO) / 4
    dQr = (dS @ Kd.view(B, N1, H, 16).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, R, 128)
    dKr = (dS.transpose(2, 3) @ Qd.view(B, R, H, 16).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, N1, 128)
    dVr = (Ad.transpose(2, 3) @ dOh).permute(0, 2, 1, 3).reshape(B, N1, 128)
    prev = torch.cat([torch.zeros(B, 1, M, dtype=torch.long, device=DEV), fl.permute(0, 2, 1)[:, :-1]], dim=1).reshape(B, R)
    dQ1r = torch.zeros(B, N1, 128, dtype=torch.float64, device=DEV).scatter_add_(1, prev[:, :, None].expand(B, R, 128), dQr)

    # f32 MFMAs: rounding of f32 sums.  Split-bf16: the operands carry 16 significand bits, the score product of mode 2 carries 24
    tol = {0: 2e-5, 1: 1e-4, 2: 6e-5}[mode]
    worst = {}

    def close(got, ref, what):
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        worst[what] = err
        assert err < (tol if what in ("dK", "dV", "dQ1", "dQ2", "dwl") else 2e-5), (what, err)
    close(dO, dOr, "dO"); close(dPK, dPKr, "dPK"); close(dpb, dl.sum(1), "dpb"); close(rowDU, refU, "dU")
    close(dK, dKr, "dK"); close(dV, dVr, "dV"); close(dQ1, dQ1r, "dQ1")
    assert torch.equal(ixP.long(), prev)
    if tsp:
        first = fl[:, :, 0][:, None, :].expand(B, T, M).reshape(B, R)
        dQ2r = torch.zeros(B, N1, 128, dtype=torch.float64, device=DEV).scatter_add_(1, first[:, :, None].expand(B, R, 128), dQr)
        close(dQ2, dQ2r, "dQ2")
    else:
        close(dwl, torch.einsum("br,bre->e", Load[:, :R].double(), dQr), "dwl")
    if recompute:
        gpc.record_parity(f"decoder_bwd_mode{mode}_n{N1}_glimpse_rel_err", max(worst[k] for k in ("dK", "dV", "dQ1")))


def test_adam_matches_torch_and_checkpoints():
    from elg_amd.optim import Adam
    torch.manual_seed(0)
    shapes = [(128, 128), (128,), (3, 7, 5), (1,), (513,)]
    pa = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = Adam(pa, lr=1e-3, weight_decay=1e-2)
    ob = torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-2)
    for it in range(5):
        oa.zero_grad(); ob.zero_grad()
        gs = [torch.randn(s, device=DEV) * (10.0 ** (it - 2)) for s in shapes]
        la = sum((p * g).sum() for p, g in zip(pa, gs))
        lb = sum((p * g).sum() for p, g in zip(pb, gs))
        la.backward(); lb.backward()
        oa.step(); ob.step()
        for x, y in zip(pa, pb):
            np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=0, atol=2e-6)
    # checkpoint layout = torch.optim.Adam's: load ours into torch and theirs into ours, then one more equal step
    sd = oa.state_dict()
    ob2 = torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-2)
    ob2.load_state_dict(sd)
    oa2 = Adam(pa, lr=1e-3, weight_decay=1e-2)
    oa2.load_state_dict(ob.state_dict())
    assert oa2.step_count == 5
    oa2.zero_grad(); ob2.zero_grad()
    gs = [torch.randn(s, device=DEV) for s in shapes]
    sum((p * g).sum() for p, g in zip(pa, gs)).backward()
    sum((p * g).sum() for p, g in zip(pb, gs)).backward()
    oa2.step(); ob2.step()
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=0, atol=3e-6)


def test_adam_handles_missing_gradients():
    """A parameter that took no part in the step has grad None: it is treated as a zero gradient."""
    from elg_amd.optim import Adam
    p = [torch.nn.Parameter(torch.randn(8, device=DEV)), torch.nn.Parameter(torch.randn(5, device=DEV))]
    w1 = p[1].detach().clone()
    opt = Adam(p, lr=1e-3)
    opt.zero_grad()
    (p[0] * 2).sum().backward()
    opt.step()
    assert torch.equal(p[1].detach(), w1) and opt.step_count == 1


def test_grad_bucket_uses_the_optimizer_buffer(monkeypatch):
    """Data-parallel path of bench.py / train.py: the all-reduce runs on the optimiser's packed gradient buffer and
    1/world is folded into the Adam kernel.  Two ranks with identical gradients (all_reduce emulated as x2) must
    give exactly the single-rank update."""
    import torch.distributed as dist
    from elg_amd import parallel
    from elg_amd.optim import Adam
    torch.manual_seed(3)
    shapes = [(64, 32), (32,), (5, 3)]
    pa = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = Adam(pa, lr=1e-3, weight_decay=1e-6), Adam(pb, lr=1e-3, weight_decay=1e-6)
    bucket = parallel.GradBucket(pa, oa)
    assert bucket.flat.data_ptr() == oa.grad_flat.data_ptr()
    calls = []

    def fake_all_reduce(t, op=None):
        calls.append(t.data_ptr())
        t.mul_(2.0)                                    # sum over two identical ranks
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    gs = [torch.randn(s, device=DEV) for s in shapes]
    for ps, opt in ((pa, oa), (pb, ob)):
        opt.zero_grad()
        sum((p * g).sum() for p, g in zip(ps, gs)).backward()
    bucket.allreduce(2)
    assert calls == [oa.grad_flat.data_ptr()] and oa.grad_scale == 0.5
    oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert torch.equal(x.detach(), y.detach())


@pytest.mark.parametrize("B,N,C", [(3, 101, 128), (2, 21, 128), (1, 7, 32)])
def test_add_instance_norm_matches_torch(B, N, C):
    """csrc/elg_encoder.hip against nn.InstanceNorm1d(affine) on (a + b) (reference models.py:506-527):
    outputs 2e-6 abs on O(1) values, gradients 1e-5 of their scale."""
    from elg_amd import engine as eng
    torch.manual_seed(N)
    a = torch.randn(B, N, C, device=DEV, requires_grad=True)
    b = (torch.randn(B, N, C, device=DEV) * 3 + 1).requires_grad_(True)
    norm = torch.nn.InstanceNorm1d(C, affine=True, track_running_stats=False).to(DEV)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5); norm.bias.uniform_(-1, 1)
    ref = norm((a + b).transpose(1, 2)).transpose(1, 2)
    got = eng.add_instance_norm(a, b, norm.weight, norm.bias, norm.eps)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0, atol=3e-6)
    w = torch.randn_like(ref)
    gr = torch.autograd.grad((ref * w).sum(), [a, b, norm.weight, norm.bias])
    gg = torch.autograd.grad((got * w).sum(), [a, b, norm.weight, norm.bias])
    for x, y in zip(gg, gr):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=0, atol=1e-5 * max(1.0, y.abs().max().item()))


def test_training_steps_do_not_accumulate_device_memory():
    """No reference cycles through the autograd nodes: with Python's cyclic collector switched off, the allocated device
    memory after step 12 equals the one after step 4 (an output tensor stored on a Function's ctx used to leak the decoder
    tables of every step, 16.5 MB at the bench shape, until a long run ran out of memory)."""
    import gc
    import yaml
    import os
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.CVRPModel import CVRPModel
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.optim import Adam
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "elg_amd", "CVRP", "config.yml")))
    dev = "cuda:0"
    model = CVRPModel(**cfg["model_params"])
    model.decoder.add_local_policy(dev)
    model.to(dev).train()
    env = CVRPEnv(50, dev)
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    dist = dict(cfg["distribution"], data_type="uniform")
    gc.collect()
    gc.disable()
    try:
        mem = []
        for i in range(12):
            train_step(model, env, opt, generate_vrp_data(16, 50, dist), True)
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
        assert mem[11] - mem[3] < (1 << 20), [m >> 20 for m in mem]
    finally:
        gc.enable()


def _glue_model(n=50):
    import os
    import yaml
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.CVRPModel import CVRPModel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "elg_amd", "CVRP", "config.yml")))
    torch.manual_seed(5)
    model = CVRPModel(**cfg["model_params"])
    model.decoder.add_local_policy("cuda:0")
    model.to("cuda:0").train()
    return cfg, model, CVRPEnv(n, "cuda:0")


def test_deferred_sync_rollout_matches_the_synchronous_sequence():
    """rollout_train (rollout length left on the device, host sync after the backward is queued) against
    rollout() -> check_feasible -> pomo_loss -> backward with the same draws: same actions, loss and gradients (the kernels see
    the same rows; only the order of the f32 atomics differs)."""
    import random
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import pomo_loss
    from elg_amd.CVRP.utils import check_feasible, rollout, rollout_train
    cfg, model, env = _glue_model(50)
    dist = dict(cfg["distribution"], data_type="uniform")
    batch = generate_vrp_data(8, 50, dist)

    def run(deferred):
        random.seed(3)
        torch.manual_seed(3)
        env.load_random_problems(batch)
        rs, _, _ = env.reset()
        model.zero_grad(set_to_none=True)
        model.pre_forward(rs)
        if deferred:
            ro = rollout_train(model, env, rs.node_demand[0])
            J = pomo_loss(ro.probs, ro.reward, True)
            J.backward()
            sol, rew = ro.finish(want_actions=True), ro.reward
            assert ro.probs.shape[1] >= sol.shape[2]
        else:
            sol, probs, rew = rollout(model, env, 'sample')
            check_feasible(sol[0:1], rs.node_demand[0:1])
            J = pomo_loss(probs, rew, True)
            J.backward()
        torch.cuda.synchronize()
        return sol.clone(), rew.clone(), float(J.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    sol_a, rew_a, J_a, g_a = run(False)
    sol_b, rew_b, J_b, g_b = run(True)
    assert torch.equal(sol_a, sol_b) and torch.equal(rew_a, rew_b)
    assert abs(J_a - J_b) <= 1e-6 * max(1.0, abs(J_a)), (J_a, J_b)
    assert g_a.keys() == g_b.keys() and len(g_a) > 80
    gmax = max(float(g.abs().max()) for g in g_a.values())
    for n in g_a:
        # (the biases in front of an instance norm have a mathematically zero gradient: rounding noise, measured on the global scale)
        scale = max(float(g_a[n].abs().max()), 1e-2 * gmax)
        err = float((g_a[n] - g_b[n]).abs().max()) / scale
        assert err < 2e-4, (n, err)


def test_rollout_stats_marks_the_steps_with_a_zero_probability():
    """elg_rollout_stats: longest trajectory, the any-zero flag and the per-step flags (only steps a trajectory decoded count)."""
    from elg_amd import engine as eng
    B, M, Tcap = 3, 37, 29
    g = torch.Generator().manual_seed(1)
    tlen = torch.randint(5, 20, (B, M), generator=g, dtype=torch.int32)
    probs = torch.rand(B, Tcap, M, generator=g) * 0.9 + 0.05
    probs[1, 4, 7] = 0.0                                   # inside (tlen >= 5)
    probs[2, 25, 3] = 0.0                                  # past every trajectory's end: ignored
    tlen[0, 11] = 23
    probs[0, 22, 11] = 0.0                                 # last decoded step of the longest trajectory
    res = eng.RolloutResult(actions=torch.zeros(B, M, Tcap, dtype=torch.int32, device="cuda:0"), probs=probs.cuda(),
                            reward=torch.zeros(B, M, device="cuda:0"), tlen=tlen.cuda())
    stats, zsteps, block = eng.rollout_stats_launch(res)
    assert stats.tolist() == [23, 1] and block.tolist() == [23, 1, 0, 0]
    want = [0] * Tcap
    want[4] = want[22] = 1
    assert zsteps.tolist() == want
    res.probs[1, 4, 7] = 0.5
    res.probs[0, 22, 11] = 0.5
    assert eng.rollout_stats(res) == (23, False)


def test_fused_pomo_loss_and_gradient_equal_the_kernel_plus_autograd_chain():
    """elg_pomo_loss_grad (the training step's path: scaled loss + d loss / d chosen probabilities in one launch, the +1e-6 of
    the zero-probability steps inside it) against elg_pomo_loss + torch's element-wise chain, and against the oracle's
    pomo_loss in float64."""
    from elg_amd import engine as eng
    from oracle import elg_oracle as orc
    torch.manual_seed(3)
    B, T, M = 5, 37, 23
    probs0 = torch.rand(B, T, M).clamp_min(0.02)
    probs0[1, 4, 7] = 0.0                                   # a chosen probability of exactly 0 at step 4 ...
    z = torch.zeros(T, dtype=torch.int32)
    z[4] = 1                                                # ... flagged by elg_rollout_stats
    rew = torch.randn(B, M)
    out = {}
    for fused in (True, False):
        p = probs0.clone().to(DEV).requires_grad_(True)
        if fused:
            J = eng.pomo_loss(p, rew.to(DEV), True, zero_steps=z.to(DEV))
        else:
            J = eng._PomoLoss.apply(torch.add(p, z.to(DEV)[None, :, None], alpha=1e-6), rew.to(DEV), True, False)
        J.backward()
        out[fused] = (float(J.detach()), p.grad.cpu())
    p64 = (probs0.double() + 1e-6 * z.double()[None, :, None]).requires_grad_(True)
    Jo = orc.pomo_loss(p64, rew.double(), True)
    Jo.backward()
    assert abs(out[True][0] - out[False][0]) <= 1e-6 * abs(out[False][0])
    assert abs(out[True][0] - float(Jo.detach())) <= 2e-5 * abs(float(Jo.detach()))
    np.testing.assert_allclose(out[True][1].numpy(), out[False][1].numpy(), rtol=2e-6, atol=0)
    np.testing.assert_allclose(out[True][1].numpy(), p64.grad.numpy(), rtol=2e-5, atol=0)
    # the training step's form: padded steps of probability 1 behind the device-resident step count are not read, the cotangent is the
    # cached unit scalar (no product) -- the same loss and gradient, bit for bit, as the full walk with the implicit cotangent; twice
    # (the ticket word that elects the workgroup adding the terms must come back to zero)
    T2 = T + 9
    pad = torch.ones(B, T2, M)
    pad[:, :T] = probs0
    z2 = torch.zeros(T2, dtype=torch.int32)
    z2[4] = 1
    Tdev = torch.tensor([T, 0], dtype=torch.int32, device=DEV)
    got = []
    for T_dev, unit in ((None, False), (Tdev, True), (Tdev, True)):
        p = pad.clone().to(DEV).requires_grad_(True)
        J = eng.pomo_loss(p, rew.to(DEV), True, zero_steps=z2.to(DEV), T_dev=T_dev)
        J.backward(eng.unit_grad(DEV)) if unit else J.backward()
        got.append((J.detach().cpu(), p.grad.cpu()))
    for J2, g2 in got[1:]:
        assert torch.equal(J2, got[0][0]) and torch.equal(g2, got[0][1])
    assert abs(float(got[0][0]) - out[True][0]) <= 1e-6 * abs(out[True][0])
    assert torch.equal(got[0][1][:, :T], out[True][1])
