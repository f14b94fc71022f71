"""How far ahead of the GPU the host runs in train_step at the bench shape: wall time per step, and the part of it the host
spends blocked in the step's one host sync (TrainRollout.finish -> HostFetch.get).  Blocked time ~ 0 would mean the step is
host-bound (every Python microsecond between two launches shows up as GPU idle)."""
import os, sys, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP import train as tr, utils as ut
from elg_amd.CVRP.utils import seed_everything
from elg_amd.optim import Adam
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev); opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
blocked = [0.0]
orig = ut.TrainRollout.finish
def timed_finish(self):
    t = time.perf_counter(); r = orig(self); blocked[0] += time.perf_counter() - t; return r
ut.TrainRollout.finish = timed_finish
gen = [0.0]
N = 200
for i in range(N + 20):
    if i == 20:
        torch.cuda.synchronize(); blocked[0] = 0.0; gen[0] = 0.0; T0 = time.perf_counter()
    t = time.perf_counter()
    batch = generate_vrp_data(64, 100, dict(cfg["distribution"], data_type="uniform"))
    gen[0] += time.perf_counter() - t
    tr.train_step(model, env, opt, batch, True)
torch.cuda.synchronize()
wall = (time.perf_counter() - T0) / N * 1e3
print(f"wall {wall:.3f} ms/step, blocked in the host sync {blocked[0] / N * 1e3:.3f} ms/step, data generation {gen[0] / N * 1e3:.3f} ms/step, "
      f"host work {wall - blocked[0] / N * 1e3:.3f} ms/step")
