"""Cooperative vs per-wavefront forward kernel at the bench shape, with phase ablations."""
import sys, os, time, torch, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
def run(tag, pol, debug=0, train=False, mode=L.MODE_SAMPLE):
    res = eng.rollout_forward(prob, pol, M, starts, mode, seed=1, debug=debug, train=train); torch.cuda.synchronize()
    t0 = time.time()
    for i in range(5): res = eng.rollout_forward(prob, pol, M, starts, mode, seed=i, debug=debug, train=train)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    steps = res.tlen.sum().item()
    print(f"{tag:44s}: {dt*1e3:7.2f} ms  Tmax={res.tlen.max().item()} ns/traj-step={dt/steps*1e9:.1f}", flush=True)
p2 = copy.copy(pol); p2.has_local = False
p3 = copy.copy(pol); p3.has_local = False; p3.has_penalty = False
run("per-wave kernel, full", pol, debug=8)
run("per-wave kernel, full, train", pol, debug=8, train=True)
run("coop, full", pol)
run("coop, full, train", pol, train=True)
run("coop, batched choice, per-trajectory advance", pol, debug=64)
run("coop, batched choice, per-trajectory advance, train", pol, debug=64, train=True)
run("coop, per-trajectory finish", pol, debug=32)
run("coop, per-trajectory finish, train", pol, debug=32, train=True)
run("coop, no MFMA phases", pol, debug=16)
run("coop, no local", p2)
run("coop, no local, no penalty", p3)
run("coop, no local/penalty, no MFMA phases", p3, debug=16)
run("coop greedy", pol, mode=L.MODE_GREEDY)
run("per-wave greedy", pol, debug=8, mode=L.MODE_GREEDY)
