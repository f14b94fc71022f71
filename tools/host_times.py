"""Host-side (enqueue) time of each phase of train_step, no device syncs except the rollout's own .item()."""
import sys, os, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.train import pomo_loss
from elg_amd.CVRP.utils import seed_everything, rollout
from elg_amd.optim import Adam
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev); opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
acc = {}
def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
N = 30
for i in range(N + 5):
    if i == 5: acc.clear(); torch.cuda.synchronize(); T0 = time.perf_counter()
    t = time.perf_counter(); batch = generate_vrp_data(64, 100, dict(cfg["distribution"], data_type="uniform")); tick("gen", t)
    t = time.perf_counter(); env.load_random_problems(batch); rs, _, _ = env.reset(); tick("env.load+reset", t)
    t = time.perf_counter(); model.pre_forward(rs); tick("pre_forward", t)
    t = time.perf_counter(); sol, probs, rew = rollout(model, env, 'sample'); tick("rollout(+sync)", t)
    t = time.perf_counter(); opt.zero_grad(); J = pomo_loss(probs, rew, True); tick("loss", t)
    t = time.perf_counter(); J.backward(); tick("backward", t)
    t = time.perf_counter(); opt.step(); tick("adam", t)
torch.cuda.synchronize(); total = (time.perf_counter() - T0) / N * 1e3
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "host sum", round(sum(acc.values()) / N * 1e3, 3), "wall/step", round(total, 3))
