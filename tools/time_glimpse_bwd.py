"""Time elg_glimpse_bwd_fused alone at the bench shape (B=64, R=121*100 rows, N1=101)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd import _lib as L, engine as eng
dev = "cuda:0"
B, M, T, N1, Tcap = 64, 100, 121, 101, 202
if len(sys.argv) > 1: B = int(sys.argv[1])
R, Rcap, E = M * T, M * Tcap, 128
A = torch.softmax(torch.randn(B, 8, Rcap, N1, device=dev), -1)
K = torch.randn(B, N1, E, device=dev); V = torch.randn(B, N1, E, device=dev)
Q = torch.randn(B, Rcap, E, device=dev); O = torch.randn(B, Rcap, E, device=dev); dO = torch.randn(B, R, E, device=dev)
dQ = torch.empty(B, R, E, device=dev)
RECOMP = len(sys.argv) > 2
MASK = (torch.rand(B, Rcap, 2, device=dev) * 2**62).long()
for splits in (1, 2, 4):
    dKp = torch.empty(splits, B, N1, E, device=dev); dVp = torch.empty(splits, B, N1, E, device=dev)
    def run():
        L.check(L.lib().elg_glimpse_bwd_fused(eng._ptr(A) if not RECOMP else None, eng._ptr(MASK) if RECOMP else None, eng._ptr(dO), eng._ptr(O), eng._ptr(Q), eng._ptr(K), eng._ptr(V),
                                              eng._ptr(dQ), eng._ptr(dKp), eng._ptr(dVp), B, R, N1, Rcap, Rcap, Rcap, splits,
                                              eng._stream()), "fused")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = B * 8 * (R / 16) * 112 * 2048
    print(f"splits={splits}: {ms:.3f} ms  ({flops/ms/1e9:.1f} TFLOP/s MFMA, A read {B*8*R*N1*4/ms/1e6:.0f} GB/s)", flush=True)
