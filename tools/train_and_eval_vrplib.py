"""BASELINE configs[4] with the build's own weights: train the joint CVRP-100 model from random init for a few minutes
(uniform instances, reference hyper-parameters), then run the reference's VRPLIB evaluation (greedy, x8 augmentation,
pomo = min(N, 1000)) on all 100 X-set instances through elg_amd.CVRP.test_vrplib.VRPLib_Tester."""
import sys, os, time, json, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.train import train_step
from elg_amd.CVRP.utils import seed_everything
from elg_amd.CVRP.test_vrplib import VRPLib_Tester
from elg_amd.optim import Adam
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # > 0: VRPLIB evaluation every EVERY steps (saved as it goes)
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(cfg["seed"]); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev)
opt = Adam(model.parameters(), lr=cfg["params"]["learning_rate"], weight_decay=1e-6)
kinds = ["uniform", "cluster", "mixed"]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(os.path.join(ROOT, "gpurun_out"))
curve = []


def evaluate():
    import io, contextlib
    tester = VRPLib_Tester(dict(cfg, load_checkpoint=None, name="elg_amd_selftrained"), model=model)
    tester.vrplib_path = os.path.join(gu.GOLDEN_DIR, "vrplib", "X") + "/"
    with contextlib.redirect_stdout(io.StringIO()):
        res, summ = tester.test_on_vrplib()
    model.requires_grad_(True)
    model.train()
    return res, summ


t0 = time.time()
for i in range(STEPS):
    dist = dict(cfg["distribution"], data_type=kinds[i % 3])          # all three families of the reference's curriculum
    J, rew = train_step(model, env, opt, generate_vrp_data(64, 100, dist), cfg["params"]["scale_norm"])
    if (i + 1) % 5000 == 0:
        torch.cuda.synchronize(); print(f"step {i+1}: {time.time()-t0:.0f} s, sampled cost {float(-rew.mean()):.3f}", flush=True)
    if EVERY and (i + 1) % EVERY == 0 and (i + 1) < STEPS:
        _, summ = evaluate()
        curve.append({"step": i + 1, "seconds": round(time.time() - t0, 1), "gap_percent": summ})
        print(curve[-1], flush=True)
        json.dump({"curve": curve}, open("vrplib_selftrained_curve.json", "w"), indent=1)
torch.cuda.synchronize(); train_s = time.time() - t0
t1 = time.time()
results, summary = evaluate()
eval_s = time.time() - t1
out = {"train_steps": STEPS, "train_seconds": round(train_s, 1), "train_instances": STEPS * 64,
       "eval_seconds_100_instances": round(eval_s, 1), "summary_gap_percent": summary, "curve": curve,
       "per_instance": [{"instance": r["instance"], "n": r["record"][0]["scale"], "gap_percent": round(100 * r["record"][0]["gap"], 2)} for r in results]}
json.dump(out, open("vrplib_selftrained_long.json" if EVERY else "vrplib_selftrained.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_instance"}))
