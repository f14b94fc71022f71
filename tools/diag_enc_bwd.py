import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import gpu_common as gc, golden_util as gu
from oracle import elg_oracle as orc
from elg_amd import encoder as enc_host
import test_gpu_encoder as T
for problem, B, N1 in [("cvrp", 2, 21), ("cvrp", 3, 101), ("tsp", 2, 128)]:
    mp, cfg, P, xy, dem, kind, names = T._setup(problem, B, N1, 7)
    g = torch.Generator().manual_seed(11)
    keys = ["enc", "K", "V", "PK", "pb", "Q1"] + (["Q2"] if problem == "tsp" else ["wl"])
    shapes = {"enc": (B, N1, 128), "K": (B, N1, 128), "V": (B, N1, 128), "PK": (B, N1, 128), "pb": (B, N1), "Q1": (B, N1, 128), "Q2": (B, N1, 128), "wl": (128,)}
    cot = {k: torch.randn(*shapes[k], generator=g) for k in keys}
    def run(dt):
        Pd = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in P.items()}
        e, t = T._oracle_tables(Pd, cfg, xy.to(dt), None if dem is None else dem.to(dt), kind)
        t = dict(t, enc=e)
        sum((t[k] * cot[k].to(dt)).sum() for k in keys).backward()
        return {n: Pd[n].grad.double().numpy() for n in names}
    g64, g32 = run(torch.float64), run(torch.float32)
    params = [P[n].detach().clone().to("cuda").contiguous().requires_grad_(True) for n in names]
    enc, t = enc_host.encode_and_fold(kind, xy.cuda(), None if dem is None else dem.cuda(), params, cfg.encoder_layer_num, mp["ff_hidden_dim"])
    t = dict(t, enc=enc)
    sum((t[k] * cot[k].cuda()).sum() for k in keys).backward()
    rows = []
    for n, p in zip(names, params):
        ref = g64[n]; sc = np.abs(ref).max()
        rows.append((np.abs(p.grad.cpu().double().numpy() - ref).max(), np.abs(g32[n] - ref).max(), sc, n))
    rows.sort(key=lambda r: -r[0] / max(r[2], 1e-9))
    print(problem, B, N1)
    for r in rows[:12]:
        print(f"  ours {r[0]:.3e}  torch32 {r[1]:.3e}  scale {r[2]:.3e}  {r[3]}")
