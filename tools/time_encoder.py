"""Times elg_encoder_fwd (+ elg_encoder_bwd) at the bench shape with HIP events."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from elg_amd import _lib as L, encoder as enc_host
B, N1 = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 101
mp = dict(gu.CVRP_MODEL_PARAMS)
names = enc_host.parameter_names(L.PROBLEM_CVRP, 6)
W = gu.golden_weights("cvrp", 1, mp, True)
params = [torch.from_numpy(W[n]).cuda().requires_grad_(True) for n in names]
xy = torch.rand(B, N1, 2, device="cuda"); dem = torch.rand(B, N1, device="cuda")
def fwd():
    return enc_host.encode_and_fold(L.PROBLEM_CVRP, xy, dem, params, 6, 512)
def both():
    enc, t = fwd()
    (t["K"].sum() + t["V"].sum() + t["PK"].sum() + t["Q1"].sum() + t["pb"].sum()).backward()
for name, fn in (("fwd", lambda: fwd()), ("fwd+bwd", both)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  (B={B})")
