"""Wall-clock per phase of one training step (with syncs between phases) + no-sync total."""
import sys, os, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.train import pomo_loss, train_step
from elg_amd.CVRP.utils import seed_everything, rollout
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-6, fused=True)
dist = dict(cfg["distribution"], data_type="uniform")
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
for it in range(8):
    t0 = sync(); batch = generate_vrp_data(64, 100, dist); env.load_random_problems(batch); rs, _, _ = env.reset()
    t1 = sync(); model.pre_forward(rs)
    t2 = sync(); sol, probs, rew = rollout(model, env, 'sample')
    t3 = sync(); opt.zero_grad(); J = pomo_loss(probs, rew, True)
    t4 = sync(); J.backward()
    t5 = sync(); opt.step()
    t6 = sync()
    if it >= 3:
        for k, v in zip(("data+load+nbr", "encoder+folds", "rollout", "loss", "backward", "adam"), (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t6-t5)):
            acc[k] = acc.get(k, 0) + v / 5
print({k: round(v * 1e3, 2) for k, v in acc.items()}, "sum", round(sum(acc.values()) * 1e3, 2))
