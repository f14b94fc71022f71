"""Host side of a training step at the bench shape: how long the Python thread needs to ENQUEUE a step (no device wait inside:
the step's only sync, HostFetch.get(), is bypassed by check=False + a patched finish), against the GPU's time per step.
python tools/host_overhead.py [profile]"""
import os, sys, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd import engine as eng
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP import train as T_
from elg_amd.CVRP import utils as U_
from elg_amd.CVRP.utils import seed_everything
from elg_amd.optim import Adam
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev)
env = CVRPEnv(100, dev); opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
batches = [generate_vrp_data(int(os.environ.get("ELG_HOST_B", "64")), 100, dict(cfg["distribution"], data_type="uniform")) for _ in range(8)]
prec = 1 if os.environ.get("ELG_FWD_MODE") == "bf16" else 0
eng.FWD_PRECISION = prec
for i in range(10): T_.train_step(model, env, opt, batches[i % 8], True, check=False)
torch.cuda.synchronize()
# 1. normal steps (with the step's host sync)
t0 = time.perf_counter()
for i in range(100): T_.train_step(model, env, opt, batches[i % 8], True, check=False)
torch.cuda.synchronize(); normal = (time.perf_counter() - t0) / 100
# 2. enqueue only: the fetch's wait removed -> the host runs as far ahead as the queues allow; host time per step = loop time
#    of the FIRST few steps, before any queue fills
orig_get = eng.HostFetch.get
eng.HostFetch.get = lambda self: [120, 0, 0, 0]
torch.cuda.synchronize()
ts = []
for i in range(6):
    t0 = time.perf_counter(); T_.train_step(model, env, opt, batches[i % 8], True, check=False); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
eng.HostFetch.get = orig_get
print(f"precision {prec}: step {normal * 1e3:.3f} ms; host enqueue per step (no wait): " + " ".join(f"{t * 1e3:.2f}" for t in ts) + " ms")
if len(sys.argv) > 1:
    import cProfile, pstats
    eng.HostFetch.get = lambda self: [120, 0, 0, 0]
    pr = cProfile.Profile(); pr.enable()
    for i in range(6): T_.train_step(model, env, opt, batches[i % 8], True, check=False)
    pr.disable(); torch.cuda.synchronize()
    eng.HostFetch.get = orig_get
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
