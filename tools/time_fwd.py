import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import golden_util as gu, gpu_common as gc
from oracle import elg_oracle as orc
from elg_amd import _lib as L, engine as eng
B, N, M = 64, 100, 100
mp = dict(gu.CVRP_MODEL_PARAMS); cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
P = gc.weights("cvrp", 5, mp, 1.0)
torch.manual_seed(0)
xy = torch.rand(B, N + 1, 2); dem = torch.cat([torch.zeros(B, 1), torch.randint(1, 10, (B, N)).float() / 50], 1)
enc = orc.encoder_forward(P, cfg, xy, dem)
prob = gc.make_problem(xy, dem, L.PROBLEM_CVRP)
pol = gc.make_policy(P, cfg, enc.to(gc.DEV), L.PROBLEM_CVRP)
starts = torch.randperm(N)[:M]
for geom in [None, (8, 4, 1), (13, 4, 1), (8, 4, 0), (8, 8, 0), (8, 16, 0)]:
    for mode in (L.MODE_SAMPLE, L.MODE_GREEDY):
        res = eng.rollout_forward(prob, pol, M, starts, mode, seed=1, geometry=geom)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(5):
            res = eng.rollout_forward(prob, pol, M, starts, mode, seed=i, geometry=geom)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        T = res.tlen.max().item(); steps = res.tlen.sum().item()
        print(f"geom={geom} mode={mode}: {dt*1e3:.2f} ms/rollout  T={T} mean_len={steps/(B*M):.1f}  {dt/T*1e6:.1f} us/step  traj-steps/s={steps/dt:.3e}")
