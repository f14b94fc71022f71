"""A/B of the large-instance rollout kernels: variant 0 (what dispatch picks: rollout_fwd_mt_kernel for 128 < N1 <= 1024) against
variant 3 (rollout_fwd_xm_kernel, rows in the L2 scratch) at TSP-500 / 1000 (batch 16, pomo N) and CVRP-1000 (batch 8, pomo 1000),
both arithmetic modes; greedy, HIP events around the launch (min of 4)."""
import os, sys, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
dev = "cuda:0"
from elg_amd import engine as eng, _lib as L
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel


def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return round(min(ts), 2), r


variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "3"])]
with open(os.path.join(ROOT, "elg_amd", "TSP", "config.yml")) as f:
    tcfg = yaml.load(f.read(), Loader=yaml.FullLoader)
torch.manual_seed(1)
tm = TSPModel(**tcfg["model_params"]); tm.decoder.add_local_policy(dev); tm.to(dev).eval()
for n in (500, 1000):
    tenv = TSPEnv(multi_width=n, device=dev)
    tenv.load_random_problems(torch.rand(16, n, 2))
    rs, _, _ = tenv.reset()
    with torch.no_grad():
        tm.pre_forward(rs)
    starts = torch.arange(n, dtype=torch.int32)
    for prec in (0, 1):
        for v in variants:
            ms, r = timed(lambda: eng.rollout_forward(tenv.problem, tm.decoder.policy, n, starts, L.MODE_GREEDY, variant=v, precision=prec))
            print(f"tsp{n} b16 precision {prec} variant {v}: {ms} ms, mean cost {float((-r.reward).mean()):.4f}", flush=True)
with open(os.path.join(ROOT, "elg_amd", "CVRP", "config.yml")) as f:
    ccfg = yaml.load(f.read(), Loader=yaml.FullLoader)
cm = CVRPModel(**ccfg["model_params"]); cm.decoder.add_local_policy(dev); cm.to(dev).eval()
for n, bsz in ((1000, 8),):
    cenv = CVRPEnv(n, dev)
    cenv.load_random_problems(dict(loc=torch.rand(bsz, n, 2), depot=torch.rand(bsz, 2),
                                   demand=torch.randint(1, 10, (bsz, n)).float() / 100.0))
    r, _, _ = cenv.reset()
    with torch.no_grad():
        cm.pre_forward(r)
    starts = torch.arange(1, n + 1, dtype=torch.int32)
    for prec in (0, 1):
        for v in variants:
            ms, r = timed(lambda: eng.rollout_forward(cenv.problem, cm.decoder.policy, n, starts, L.MODE_GREEDY, variant=v, precision=prec), reps=3)
            print(f"cvrp{n} b{bsz} precision {prec} variant {v}: {ms} ms, {int(r.tlen.max())} steps, mean cost {float((-r.reward).mean()):.4f}", flush=True)
