"""Greedy evaluation throughput (the test.py path): CVRP-100, x8 augmentation, pomo 100, batches of 100 instances."""
import sys, os, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd import evaluate as ev
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.utils import rollout
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
dev = "cuda:0"; torch.manual_seed(0)
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev)
env = CVRPEnv(100, dev)
def batches(n, bs):
    for _ in range(n // bs):
        yield {"loc": torch.rand(bs, 100, 2), "demand": torch.randint(1, 10, (bs, 100)).float() / 50, "depot": torch.rand(bs, 1, 2)}
for aug in (1, 8):
    ev.evaluate_loader(batches(200, 100), model, env, aug, rollout, lambda b: b["loc"].shape[0])      # warm-up
    torch.cuda.synchronize(); t0 = time.time()
    n = 2000
    ev.evaluate_loader(batches(n, 100), model, env, aug, rollout, lambda b: b["loc"].shape[0])
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"aug x{aug}: {n} instances in {dt:.2f} s = {n/dt:.0f} instances/s ({n*aug*100/dt/1e6:.2f} M trajectories/s)", flush=True)
