"""train_step with the reference's per-step feasibility assertion (check=True, what train.py runs) vs without."""
import sys, os, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.train import train_step
from elg_amd.CVRP.utils import seed_everything
from elg_amd.optim import Adam
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev); opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
for kind in ("uniform", "cluster", "mixed"):
    for check in (False, True):
        d = dict(cfg["distribution"], data_type=kind)
        for _ in range(5): train_step(model, env, opt, generate_vrp_data(64, 100, d), True, check=check)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): train_step(model, env, opt, generate_vrp_data(64, 100, d), True, check=check)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{kind:8s} check={check}: {dt*1e3:.2f} ms/step, {64/dt:.0f} instances/s", flush=True)
