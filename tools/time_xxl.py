"""One Vrp-Set-XXL instance (x8 augmentation, pomo 1000, greedy) stage by stage: tables, encoder, rollout (HIP events).
    python tools/time_xxl.py [Antwerp2|Leuven1|...]      (ELG_FWD_MODE=bf16 for the bf16 mode)"""
import os, sys, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from elg_amd import vrplib_io, engine as eng, _lib as L
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
name = sys.argv[1] if len(sys.argv) > 1 else "Antwerp2"
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
dev = "cuda:0"
torch.manual_seed(1)
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).eval()
sub = "X" if name.startswith("X-") else "XXL"
inst = vrplib_io.read_instance(os.path.join(gu.GOLDEN_DIR, "vrplib", sub, name + ".vrp"))
env = CVRPEnv(min(1000, len(inst["demand"]) - 1), dev)


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


for rep in range(2):
    torch.cuda.synchronize()
    e0 = ev()
    env.load_vrplib_problem(inst, aug_factor=8)
    rs, _, _ = env.reset()
    e1 = ev()
    with torch.no_grad():
        model.pre_forward(rs)
        e2 = ev()
        starts = torch.arange(1, env.multi_width + 1, dtype=torch.int32)
        res = eng.rollout_forward(env.problem, model.decoder.policy, env.multi_width, starts, L.MODE_GREEDY)
    e3 = ev()
    torch.cuda.synchronize()
    print(f"{name} N1 = {env.problem.N1}: load + neighbour tables {e0.elapsed_time(e1) :.1f} ms, encoder + tables {e1.elapsed_time(e2) :.1f} ms, "
          f"rollout {e2.elapsed_time(e3) :.1f} ms ({int(res.tlen.max())} steps, mean cost {float((-res.reward).mean()):.0f})")
