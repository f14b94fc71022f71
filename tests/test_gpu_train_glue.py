"""Training-step glue kernels (csrc/elg_train.hip) against the plain torch formulas they replace:
 * elg_pomo_loss   -- reference CVRP/train.py:112-121 / TSP/train.py:107-118 (value 1e-5 rel, gradient 1e-5 rel)
 * elg_rows_prep   -- the cotangent rows of the decoder backward (bit-exact: same fp32 operations per element)
 * elg_adam_step   -- torch.optim.Adam(weight_decay) over several steps (1e-6 abs on O(1) weights), checkpoints."""
import numpy as np
import pytest
import torch

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


@pytest.mark.parametrize("tsp", [False, True])
def test_rows_prep_matches_torch(tsp):
    from elg_amd import _lib as L, engine as eng
    B, T, M, N1, Tcap = 2, 9, 5, 23, 12
    R, Rcap = T * M, Tcap * M
    g = torch.Generator().manual_seed(3)
    gp = torch.randn(B, T, M, generator=g).to(DEV)
    pv = torch.rand(B, T, M, generator=g).to(DEV)
    tlen = torch.randint(4, T + 1, (B, M), generator=g, dtype=torch.int32).to(DEV)
    acts = torch.randint(0, N1, (B, M, Tcap), generator=g, dtype=torch.int32).to(DEV)
    PC = torch.randn(B, Rcap, N1, generator=g).to(DEV)
    Csel = torch.randn(B, Rcap, generator=g).to(DEV)
    Slot = torch.randint(-1, N1, (B, Rcap, 48), generator=g, dtype=torch.int32).to(DEV)
    t0 = 1 if tsp else 2
    inv = 0.5
    rowDL = torch.full((B, R, N1), float("nan"), device=DEV)
    rowDU = torch.full((B, R, 48), float("nan"), device=DEV)
    Load = torch.rand(B, Rcap, generator=g).to(DEV)
    ohP = torch.full((B, R, N1 + (0 if tsp else 1)), float("nan"), device=DEV)
    ohF = torch.full((B, R, N1), float("nan"), device=DEV) if tsp else None
    ixP = torch.full((B, R), -7, device=DEV, dtype=torch.int32)
    ixF = torch.full((B, R), -7, device=DEV, dtype=torch.int32) if tsp else None
    L.check(L.lib().elg_rows_prep(eng._ptr(gp), eng._ptr(pv), eng._ptr(tlen), eng._ptr(acts), eng._ptr(PC), eng._ptr(Csel),
                                  eng._ptr(Slot), eng._ptr(Load) if not tsp else None, eng._ptr(rowDL), eng._ptr(rowDU), eng._ptr(ohP), eng._ptr(ohF), eng._ptr(ixP), eng._ptr(ixF),
                                  B, T, M, N1, Tcap, Rcap, t0, inv, eng._stream()), "rows_prep")
    # the torch chain this kernel replaces (engine._ChosenProbs.backward before the fusion)
    fl = acts[:, :, :T].long()
    tt = torch.arange(T, device=DEV)[None, :, None]
    valid = (tt >= t0) & (tt < tlen[:, None, :])
    W = (gp * pv * valid).reshape(B, R)
    sel = fl.permute(0, 2, 1).reshape(B, R)
    ref = PC[:, :R] * (-W)[:, :, None]
    ref.scatter_add_(2, sel[:, :, None], (W * Csel[:, :R])[:, :, None])
    slot = Slot[:, :R].long()
    refU = torch.gather(ref, 2, slot.clamp(min=0)) * (slot >= 0) * inv
    prev = torch.cat([torch.zeros(B, 1, M, dtype=torch.long, device=DEV), fl.permute(0, 2, 1)[:, :-1]], dim=1).reshape(B, R)
    refP = torch.zeros(B, R, N1, device=DEV).scatter_(2, prev[:, :, None], 1.0)
    np.testing.assert_allclose(rowDL.cpu().numpy(), ref.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rowDU.cpu().numpy(), refU.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(ohP[:, :, :N1], refP)
    assert torch.equal(ixP.long(), prev)
    if not tsp:
        assert torch.equal(ohP[:, :, N1], Load[:, :R])
    if tsp:
        first = fl[:, :, 0][:, None, :].expand(B, T, M).reshape(B, R)
        assert torch.equal(ohF, torch.zeros(B, R, N1, device=DEV).scatter_(2, first[:, :, None], 1.0))
        assert torch.equal(ixF.long(), first)


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


@pytest.mark.parametrize("B,R,NO,wrow,splits", [(3, 203, 102, 101, 2), (2, 37, 21, -1, 1), (1, 1000, 128, -1, 4), (2, 64, 52, 51, 3)])
def test_rows_segsum_is_onehot_transpose_times_x(B, R, NO, wrow, splits):
    """elg_rows_segsum = the one-hot GEMM of the query-gather backward (fp64 reference; exact zeros for untouched nodes)."""
    from elg_amd import _lib as L, engine as eng
    g = torch.Generator().manual_seed(R)
    X = torch.randn(B, R, 128, generator=g).to(DEV)
    nn = NO - (1 if wrow >= 0 else 0)
    idx = torch.randint(0, max(1, nn - 3), (B, R), generator=g, dtype=torch.int32).to(DEV)      # last nodes never hit
    Rcap = R + 5
    w = torch.randn(B, Rcap, generator=g).to(DEV)
    part = torch.full((splits, B, NO, 128), float("nan"), device=DEV)
    L.check(L.lib().elg_rows_segsum(eng._ptr(X), eng._ptr(idx), eng._ptr(w) if wrow >= 0 else None, eng._ptr(part), B, R, NO,
                                    wrow, Rcap, splits, eng._stream()), "segsum")
    got = part.sum(0).cpu().double()
    oh = torch.zeros(B, R, NO, dtype=torch.float64, device=DEV).scatter_(2, idx.long()[:, :, None], 1.0)
    if wrow >= 0:
        oh[:, :, wrow] = w[:, :R].double()
    ref = torch.bmm(oh.transpose(1, 2), X.double()).cpu()
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5 * ref.abs().max().item())
    if nn >= 4:
        assert (got[:, nn - 2] == 0).all()
