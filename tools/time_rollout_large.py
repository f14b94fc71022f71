"""Greedy-rollout launch time of the streaming kernel family at TSP-200 / 500 / 1000 (batch 16, pomo = N) and CVRP-1000
(batch 8, pomo 1000): min and median of 5 launches, HIP events around the launch."""
import os, sys, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
dev = "cuda:0"
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.TSP.utils import rollout as tsp_rollout
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.utils import rollout as cvrp_rollout


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return round(ts[0], 2), round(ts[len(ts) // 2], 2)


with open(os.path.join(ROOT, "elg_amd", "TSP", "config.yml")) as f:
    tcfg = yaml.load(f.read(), Loader=yaml.FullLoader)
tm = TSPModel(**tcfg["model_params"]); tm.decoder.add_local_policy(dev); tm.to(dev).eval()
torch.manual_seed(1)
for n in (200, 500, 1000):
    tenv = TSPEnv(multi_width=n, device=dev)
    tenv.load_random_problems(torch.rand(16, n, 2))
    rs, _, _ = tenv.reset()
    with torch.no_grad():
        tm.pre_forward(rs)
        print("tsp%d b16 (min, median) ms" % n, timed(lambda: tsp_rollout(tm, tenv, "greedy")), flush=True)
with open(os.path.join(ROOT, "elg_amd", "CVRP", "config.yml")) as f:
    ccfg = yaml.load(f.read(), Loader=yaml.FullLoader)
cm = CVRPModel(**ccfg["model_params"]); cm.decoder.add_local_policy(dev); cm.to(dev).eval()
for n, bsz in ((500, 16), (1000, 8)):
    cenv = CVRPEnv(n, dev)
    cenv.load_random_problems(dict(loc=torch.rand(bsz, n, 2), depot=torch.rand(bsz, 2),
                                   demand=torch.randint(1, 10, (bsz, n)).float() / 100.0))
    r, _, _ = cenv.reset()
    with torch.no_grad():
        cm.pre_forward(r)
        print("cvrp%d b%d (min, median) ms" % (n, bsz), timed(lambda: cvrp_rollout(cm, cenv, "greedy"), reps=3), flush=True)
