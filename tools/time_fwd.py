"""Ablation timing of the forward rollout kernel at the bench shape (runtime flags only)."""
import sys, os, time, torch
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
import ctypes as C
DEBUG = 0
_orig_fill = eng._fill_common
def _fill(a, *args, **kw):
    _orig_fill(a, *args, **kw); a.debug_skip = DEBUG
eng._fill_common = _fill
def run(tag, pol, geom, mode=L.MODE_SAMPLE, debug=0):
    global DEBUG
    DEBUG = debug
    res = eng.rollout_forward(prob, pol, M, starts, mode, seed=1, geometry=geom); torch.cuda.synchronize()
    t0 = time.time()
    for i in range(5): res = eng.rollout_forward(prob, pol, M, starts, mode, seed=i, geometry=geom)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    steps = res.tlen.sum().item()
    print(f"{tag:32s} geom={geom}: {dt*1e3:7.2f} ms  mean_len={steps/(B*M):.1f}  ns/traj-step={dt/steps*1e9:.1f}")
import copy
for geom in [(8, 4, 1), (9, 4, 1), (13, 4, 1)]:
    run("full", pol, geom)
    p2 = copy.copy(pol); p2.has_local = False; run("no local (penalty only)", p2, geom)
    p3 = copy.copy(pol); p3.has_local = False; p3.has_penalty = False; run("no local, no penalty", p3, geom)
    run("  + skip glimpse", p3, geom, debug=1)
    run("  + skip pointer", p3, geom, debug=2)
    run("  + skip glimpse+pointer", p3, geom, debug=3)
    run("  greedy, skip glimpse+pointer", p3, geom, mode=L.MODE_GREEDY, debug=3)
