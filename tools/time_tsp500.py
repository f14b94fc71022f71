"""TSP-500 long-horizon decode stress (BASELINE configs[3]): B=16, pomo=500, greedy + sampled forward rollout."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.TSP.utils import rollout, check_feasible
dev = "cuda:0"
mp = dict(gu.TSP_MODEL_PARAMS)
torch.manual_seed(0)
model = TSPModel(**mp); model.decoder.add_local_policy(dev); model.to(dev).eval()
for (B, N) in ((16, 500), (64, 100), (32, 200)):
    env = TSPEnv(N, dev)
    env.load_random_problems(torch.rand(B, N, 2))
    rs, _, _ = env.reset()
    with torch.no_grad():
        model.pre_forward(rs)
        for mode in ("greedy", "sample"):
            rollout(model, env, mode); torch.cuda.synchronize(); t0 = time.time()
            a, p, r = rollout(model, env, mode); torch.cuda.synchronize(); dt = time.time() - t0
            assert check_feasible(a[0:1])
            print(f"TSP-{N} B={B} pomo={N} {mode}: {dt*1e3:.1f} ms/rollout, {dt/(B*N*N)*1e9:.1f} ns/traj-step, mean len {(-r).mean().item():.3f}", flush=True)
