"""Run the forward rollout and one backward a few times at the bench shape (for rocprofv3 runs)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elg_amd import engine as eng, _lib as L
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.train import train_step, pomo_loss
from elg_amd.CVRP.utils import seed_everything, rollout
import yaml
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
what = sys.argv[2] if len(sys.argv) > 2 else "train"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
seed_everything(924)
dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev)
env = CVRPEnv(100, dev)
from elg_amd.optim import Adam
opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
for i in range(reps):
    batch = generate_vrp_data(B, 100, dict(cfg["distribution"], data_type="uniform"))
    if what == "train":
        train_step(model, env, opt, batch, True, check=False)
    else:
        env.load_random_problems(batch); rs, _, _ = env.reset()
        with torch.no_grad():
            model.pre_forward(rs)
            rollout(model, env, "sample" if what == "sample" else "greedy")
torch.cuda.synchronize()
print("done")
