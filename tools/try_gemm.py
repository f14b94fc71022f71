import sys, os, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd import engine as eng
dev = "cuda:0"
def gpu_time(f, n=50):
    for _ in range(5): f()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g):
            for _ in range(n): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(6464, 128, device=dev); h = torch.randn(6464, 512, device=dev)
for (inp, out_f) in ((x, 128), (x, 384), (x, 512), (h, 128)):
    W = torch.randn(out_f, inp.shape[1], device=dev); dy = torch.randn(6464, out_f, device=dev)
    t_f = gpu_time(lambda: torch.nn.functional.linear(inp, W)); m_f = gpu_time(lambda: eng.gemm(inp, W, trans_b=True))
    t_dx = gpu_time(lambda: dy @ W); m_dx = gpu_time(lambda: eng.gemm(dy, W))
    t_dw = gpu_time(lambda: dy.t() @ inp)
    res = {}
    for sk in (8, 16, 25, 51):
        res[sk] = round(gpu_time(lambda: eng.gemm(dy, inp, trans_a=True, split_k=sk)), 1)
    print(f"in={inp.shape[1]} out={out_f}: fwd torch {t_f:.1f} us / mfma {m_f:.1f} us | dx torch {t_dx:.1f} / mfma {m_dx:.1f} | dW torch {t_dw:.1f} / mfma split-k {res}")
