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
    if (a.shape[1] % 4) or (b.shape[1] % 4):
        with pytest.raises(ValueError, match="multiples of 4"):      # 16-byte staging loads: refused, not mis-read
            eng.gemm(a, b, trans_a=ta, trans_b=tb)
        return
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
