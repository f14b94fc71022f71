"""Short real training run on the engine: CVRP-100, batch 64, pomo 100, joint model from random init, the reference's
hyper-parameters (lr 1e-4, weight decay 1e-6, scale_norm).  Prints / stores the greedy validation cost on a fixed set of
256 uniform instances every EVAL steps -- evidence that rollout, loss, every backward kernel and the optimiser learn."""
import sys, os, time, json, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.train import train_step
from elg_amd.CVRP.utils import seed_everything, rollout
from elg_amd.optim import Adam
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
EVAL = int(sys.argv[2]) if len(sys.argv) > 2 else 500
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(cfg["seed"]); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev)
env = CVRPEnv(100, dev); venv = CVRPEnv(100, dev)
opt = Adam(model.parameters(), lr=cfg["params"]["learning_rate"], weight_decay=1e-6)
dist = dict(cfg["distribution"], data_type="uniform")
g = torch.Generator().manual_seed(1234)
val = {"loc": torch.rand(256, 100, 2, generator=g), "depot": torch.rand(256, 1, 2, generator=g),
       "demand": torch.randint(1, 10, (256, 100), generator=g).float() / 50.0}
def validate():
    model.eval()
    venv.load_random_problems(val); rs, _, _ = venv.reset()
    with torch.no_grad():
        model.pre_forward(rs)
        _, _, rew = rollout(model, venv, 'greedy')
    model.train()
    return float(-rew.max(1)[0].mean())
log = []
t0 = time.time(); c = validate(); log.append({"step": 0, "greedy_cost": c, "seconds": 0.0}); print(log[-1], flush=True)
model.train()
for i in range(1, STEPS + 1):
    J, rew = train_step(model, env, opt, generate_vrp_data(64, 100, dist), cfg["params"]["scale_norm"])
    if i % EVAL == 0:
        torch.cuda.synchronize()
        log.append({"step": i, "greedy_cost": validate(), "train_sample_cost": float(-rew.mean()), "seconds": round(time.time() - t0, 1)})
        print(log[-1], flush=True)
out = os.path.join(ROOT, "gpurun_out", "train_demo.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump({"config": "CVRP-100 batch 64 pomo 100 joint, lr 1e-4, uniform", "log": log}, open(out, "w"), indent=1)
