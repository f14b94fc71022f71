"""elg_glimpse_bwd_fused (MFMA glimpse backward) against the plain fp32 torch formulas of the same contraction
(autograd of reference CVRP/models.py:478-500): dS = a (dO V^T - <dO,O>)/4, dQ = dS K, dK = dS^T Q, dV = a^T dO.
Tolerance: 2e-5 of the result's scale (fp32 accumulation order differs; the MFMA is an exact fmaf chain)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, R, N1, Rcap, splits, seed):
    from elg_amd import _lib as L, engine as eng
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(seed)
    H, E = 8, 128
    logits = torch.randn(B, H, Rcap, N1, generator=g) * 2
    logits[..., ::7] = float("-inf")                       # masked nodes: weight exactly 0
    A = torch.softmax(logits, -1)
    A[:, :, 3] = 0                                         # an inactive row (first move / finished)
    K = torch.randn(B, N1, E, generator=g)
    V = torch.randn(B, N1, E, generator=g)
    Q = torch.randn(B, Rcap, E, generator=g)
    dO = torch.randn(B, R, E, generator=g)
    heads = lambda x: x.view(B, x.shape[1], H, 16).permute(0, 2, 1, 3)          # (B,H,X,16)
    Ad = A[:, :, :R].double()
    O = torch.matmul(Ad, heads(V).double())                                      # (B,H,R,16)
    rowO = torch.zeros(B, Rcap, E)
    rowO[:, :R] = O.permute(0, 2, 1, 3).reshape(B, R, E).float()
    dOh = heads(dO).double()
    dA = torch.matmul(dOh, heads(V).double().transpose(2, 3))
    dS = 0.25 * Ad * (dA - (dOh * O).sum(-1, keepdim=True))
    dQ_ref = torch.matmul(dS, heads(K).double()).permute(0, 2, 1, 3).reshape(B, R, E)
    dK_ref = torch.matmul(dS.transpose(2, 3), heads(Q[:, :R]).double()).permute(0, 2, 1, 3).reshape(B, N1, E)
    dV_ref = torch.matmul(Ad.transpose(2, 3), dOh).permute(0, 2, 1, 3).reshape(B, N1, E)
    t = lambda x: x.to(dev).contiguous()
    Ag, Kg, Vg, Qg, dOg, Og = t(A), t(K), t(V), t(Q), t(dO), t(rowO)
    dQ = torch.full((B, R, E), float("nan"), device=dev)
    dKp = torch.full((splits, B, N1, E), float("nan"), device=dev)
    dVp = torch.full((splits, B, N1, E), float("nan"), device=dev)
    L.check(L.lib().elg_glimpse_bwd_fused(eng._ptr(Ag), None, eng._ptr(dOg), eng._ptr(Og), eng._ptr(Qg), eng._ptr(Kg),
                                          eng._ptr(Vg), eng._ptr(dQ), eng._ptr(dKp), eng._ptr(dVp), B, R, N1,
                                          Rcap, Rcap, Rcap, splits, eng._stream()), "fused")
    torch.cuda.synchronize()
    for got, ref, what in ((dQ, dQ_ref, "dQ"), (dKp.sum(0), dK_ref, "dK"), (dVp.sum(0), dV_ref, "dV")):
        got = got.cpu().double()
        assert torch.isfinite(got).all(), what
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item(), (what, err, ref.abs().max().item())


@pytest.mark.parametrize("B,R,N1,Rcap,splits", [(2, 37, 21, 40, 1), (1, 100, 51, 100, 3), (2, 203, 101, 240, 2),
                                                (1, 64, 101, 64, 1), (1, 50, 104, 50, 1), (1, 33, 128, 33, 2),
                                                (1, 16, 113, 16, 1), (3, 5, 17, 5, 4)])
def test_fused_glimpse_backward_matches_formulas(B, R, N1, Rcap, splits):
    _run(B, R, N1, Rcap, splits, seed=R + N1)


def test_fused_rejects_large_n():
    from elg_amd import _lib as L, engine as eng
    z = torch.zeros(4, device="cuda:0")
    with pytest.raises(NotImplementedError):
        L.check(L.lib().elg_glimpse_bwd_fused(eng._ptr(z), None, eng._ptr(z), eng._ptr(z), eng._ptr(z), eng._ptr(z), eng._ptr(z),
                                              eng._ptr(z), eng._ptr(z), eng._ptr(z), 1, 4, 200, 4, 4, 4, 1, eng._stream()), "x")


@pytest.mark.parametrize("B,R,N1,Rcap,splits", [(2, 203, 101, 240, 2), (1, 37, 21, 40, 1), (2, 64, 51, 64, 1), (1, 50, 112, 50, 1),
                                                (1, 33, 128, 40, 2), (1, 19, 77, 19, 3)])
def test_fused_glimpse_backward_recomputes_the_weights(B, R, N1, Rcap, splits):
    """rowMask given: a_h = softmax(q_h K_h^T / 4 + mask) is rebuilt per tile from the saved query rows and the rows'
    64-bit mask words instead of being read -- same dQ / dK / dV as with the stored weights."""
    from elg_amd import _lib as L, engine as eng
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(R * N1)
    H, E = 8, 128
    K = torch.randn(B, N1, E, generator=g)
    V = torch.randn(B, N1, E, generator=g)
    Q = torch.randn(B, Rcap, E, generator=g)
    dO = torch.randn(B, R, E, generator=g)
    closed = torch.rand(B, Rcap, N1, generator=g) < 0.4
    closed[:, 3] = True                                     # an inert row: every node closed -> weights 0, no NaN
    closed[:, :, 0] &= closed[:, :, 1:].all(-1)             # (other rows keep at least node 0 or another one open)
    closed[:, 3] = True
    words = torch.zeros(B, Rcap, 2, dtype=torch.int64)
    for n in range(N1):
        bit = closed[:, :, n].long() << (n % 64)
        if n % 64 == 63:
            bit = torch.where(closed[:, :, n], torch.tensor(-2 ** 63), torch.tensor(0))
        words[:, :, n // 64] |= bit
    heads = lambda x: x.view(B, x.shape[1], H, 16).permute(0, 2, 1, 3)
    S = torch.matmul(heads(Q).double(), heads(K).double().transpose(2, 3)) * 0.25            # (B,H,Rcap,N1)
    S = S.masked_fill(closed[:, None], float("-inf"))
    A = torch.softmax(S, -1)
    A = torch.nan_to_num(A, nan=0.0)[:, :, :R]
    O = torch.matmul(A, heads(V).double())
    rowO = torch.zeros(B, Rcap, E)
    rowO[:, :R] = O.permute(0, 2, 1, 3).reshape(B, R, E).float()
    dOh = heads(dO).double()
    dA = torch.matmul(dOh, heads(V).double().transpose(2, 3))
    dS = 0.25 * A * (dA - (dOh * O).sum(-1, keepdim=True))
    dQ_ref = torch.matmul(dS, heads(K).double()).permute(0, 2, 1, 3).reshape(B, R, E)
    dK_ref = torch.matmul(dS.transpose(2, 3), heads(Q[:, :R]).double()).permute(0, 2, 1, 3).reshape(B, N1, E)
    dV_ref = torch.matmul(A.transpose(2, 3), dOh).permute(0, 2, 1, 3).reshape(B, N1, E)
    t = lambda x: x.to(dev).contiguous()
    Kg, Vg, Qg, dOg, Og, Mg = t(K), t(V), t(Q), t(dO), t(rowO), t(words)
    dQ = torch.full((B, R, E), float("nan"), device=dev)
    dKp = torch.full((splits, B, N1, E), float("nan"), device=dev)
    dVp = torch.full((splits, B, N1, E), float("nan"), device=dev)
    L.check(L.lib().elg_glimpse_bwd_fused(None, eng._ptr(Mg), eng._ptr(dOg), eng._ptr(Og), eng._ptr(Qg), eng._ptr(Kg),
                                          eng._ptr(Vg), eng._ptr(dQ), eng._ptr(dKp), eng._ptr(dVp), B, R, N1,
                                          0, Rcap, Rcap, splits, eng._stream()), "fused-recomp")
    torch.cuda.synchronize()
    for got, ref, what in ((dQ, dQ_ref, "dQ"), (dKp.sum(0), dK_ref, "dK"), (dVp.sum(0), dV_ref, "dV")):
        got = got.cpu().double()
        assert torch.isfinite(got).all(), what
        err = (got - ref).abs().max().item()
        assert err <= 5e-5 * ref.abs().max().item(), (what, err, ref.abs().max().item())
