import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd import _lib as L, engine as eng
dev = "cuda:0"; B, R, NO, Rcap = 64, 12100, 102, 20200
X = torch.randn(B, R, 128, device=dev); idx = torch.randint(0, 101, (B, R), device=dev, dtype=torch.int32); w = torch.randn(B, Rcap, device=dev)
for sp in (4, 8, 16, 32):
    part = torch.empty(sp, B, NO, 128, device=dev)
    def run(): L.check(L.lib().elg_rows_segsum(eng._ptr(X), eng._ptr(idx), eng._ptr(w), eng._ptr(part), B, R, NO, 101, Rcap, sp, eng._stream()), "s")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f"splits={sp}: {e0.elapsed_time(e1)/10:.3f} ms", flush=True)
oh = torch.zeros(B, R, NO, device=dev)
def g(): return torch.bmm(oh.transpose(1, 2), X)
for _ in range(3): g()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g()
e1.record(); torch.cuda.synchronize(); print(f"onehot bmm: {e0.elapsed_time(e1)/10:.3f} ms")
