"""The hand-written fp32 MFMA GEMM (csrc/elg_gemm.hip) against torch fp64 matmul: all transpose forms, ragged
sizes, bias / ReLU epilogues, split-K and the row-sum (bias gradient) output."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(a, b, ta, tb, bias=None, relu=False):
    A = a.double().t() if ta else a.double()
    B = b.double().t() if tb else b.double()
    c = A @ B
    if bias is not None:
        c = c + bias.double()
    return torch.relu(c) if relu else c


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (6464, 128, 128), (6464, 512, 128), (101, 130, 52), (128, 128, 6464), (77, 40, 36)])
def test_gemm_matches_fp64(ta, tb, M, N, K):
    from elg_amd import engine as eng
    torch.manual_seed(0)
    # asymmetric integer-ish data catches swapped / transposed fragment maps exactly
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), device=DEV).float() + 0.25
    b = torch.randint(-3, 4, (N, K) if tb else (K, N), device=DEV).float() - 0.5
    # (rows that are not 16-byte aligned -- 130, 52, 77, 36 columns -- are staged with scalar loads)
    c = eng.gemm(a, b, trans_a=ta, trans_b=tb)
    ref = _ref(a, b, ta, tb)
    assert c.shape == (M, N)
    np.testing.assert_allclose(c.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-3)
    a, b = torch.randn_like(a), torch.randn_like(b)
    c = eng.gemm(a, b, trans_a=ta, trans_b=tb)
    ref = _ref(a, b, ta, tb)
    err = (c.double() - ref).abs().max().item()
    assert err < 2e-5 * np.sqrt(K) * 4, err


def test_gemm_epilogues_and_split_k():
    from elg_amd import engine as eng
    torch.manual_seed(1)
    x, W, b = torch.randn(6464, 128, device=DEV), torch.randn(512, 128, device=DEV), torch.randn(512, device=DEV)
    y = eng.gemm(x, W, trans_b=True, bias=b, relu=True)
    np.testing.assert_allclose(y.cpu().numpy(), _ref(x, W, False, True, b, True).cpu().numpy(), rtol=1e-4, atol=1e-4)
    dy = torch.randn(6464, 512, device=DEV)
    for sk in (1, 8, 25, 64):
        dW = eng.gemm(dy, x, trans_a=True, split_k=sk)
        np.testing.assert_allclose(dW.cpu().numpy(), (dy.double().t() @ x.double()).cpu().numpy(), rtol=1e-3, atol=5e-3)


@pytest.mark.parametrize("rows,out,inp,sk", [(6464, 512, 128, 50), (6464, 128, 512, 25), (333, 70, 36, 3), (64, 64, 32, 1)])
def test_gemm_row_sums_give_the_bias_gradient(rows, out, inp, sk):
    """a_rowsum: the sums of op(A)'s rows come out of the same staged tiles (db of dW = dY^T X)."""
    from elg_amd import engine as eng
    torch.manual_seed(4)
    dy, x = torch.randn(rows, out, device=DEV), torch.randn(rows, inp, device=DEV)
    if out % 4 or inp % 4:
        dy, x = dy[:, :out - out % 4].contiguous(), x[:, :inp - inp % 4].contiguous()
    db = torch.zeros(dy.shape[1], device=DEV)
    dW = eng.gemm(dy, x, trans_a=True, split_k=sk, a_rowsum=db)
    np.testing.assert_allclose(dW.cpu().numpy(), (dy.double().t() @ x.double()).cpu().numpy(), rtol=1e-3, atol=5e-3)
    np.testing.assert_allclose(db.cpu().numpy(), dy.double().sum(0).cpu().numpy(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False)])
def test_batched_gemm_with_head_strides(ta, tb):
    """elg_gemm_f32_batched: two-level batch (instance, head) whose inner stride is a 16-column offset into 128-wide rows and
    whose rows have an odd leading dimension (N + 1 nodes) -- the shapes of the replay backward for N + 1 > 128."""
    import ctypes as C
    from elg_amd import _lib as L
    torch.manual_seed(2)
    B, H, R, N1 = 2, 8, 77, 301
    rows = torch.randn(B, H, R, N1, device=DEV)          # a_h / dS_h
    wide = torch.randn(B, R, 128, device=DEV)            # dO / Q rows, heads side by side
    tab = torch.randn(B, N1, 128, device=DEV)            # K / V tables
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    if ta:      # dK_h = rows_h^T wide_h : (N1 x R) @ (R x 16)
        out = torch.empty(B, N1, 128, device=DEV)
        L.check(L.lib().elg_gemm_f32_batched(p(rows), p(wide), p(out), N1, 16, R, N1, 128, 128, 1, 0, B, H, H * R * N1, R * N1,
                                             R * 128, 16, N1 * 128, 16, 1.0, st), "batched")
        ref = torch.einsum("bhrn,brhd->bnhd", rows.double(), wide.double().view(B, R, H, 16)).reshape(B, N1, 128)
    elif tb:    # dA_h = wide_h tab_h^T : (R x 16) @ (16 x N1)
        out = torch.empty(B, H, R, N1, device=DEV)
        L.check(L.lib().elg_gemm_f32_batched(p(wide), p(tab), p(out), R, N1, 16, 128, 128, N1, 0, 1, B, H, R * 128, 16,
                                             N1 * 128, 16, H * R * N1, R * N1, 1.0, st), "batched")
        ref = torch.einsum("brhd,bnhd->bhrn", wide.double().view(B, R, H, 16), tab.double().view(B, N1, H, 16))
    else:       # dQ_h = rows_h tab_h : (R x N1) @ (N1 x 16)
        out = torch.empty(B, R, 128, device=DEV)
        L.check(L.lib().elg_gemm_f32_batched(p(rows), p(tab), p(out), R, 16, N1, N1, 128, 128, 0, 0, B, H, H * R * N1, R * N1,
                                             N1 * 128, 16, R * 128, 16, 1.0, st), "batched")
        ref = torch.einsum("bhrn,bnhd->brhd", rows.double(), tab.double().view(B, N1, H, 16)).reshape(B, R, 128)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-5 * np.sqrt(max(R, N1)) * 4, err
