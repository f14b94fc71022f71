"""TSP counterpart of train_and_eval_vrplib.py: train the joint TSP-100 model from random init for a few minutes, then the
reference's TSPLIB evaluation (49 instances, N <= 1002, greedy, x8 augmentation, pomo = N) -- the only benchmark the
reference publishes numbers for (fully trained: 3.06 % total)."""
import sys, os, time, json, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
from elg_amd.TSP.generate_data import generate_tsp_data
from elg_amd.TSP.train import train_step
from elg_amd.TSP.utils import seed_everything
from elg_amd.TSP.test_tsplib import TSPLib_Tester
from elg_amd.optim import Adam
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # > 0: TSPLIB evaluation every EVERY steps (saved as it goes)
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/TSP/config.yml")))
seed_everything(cfg.get("seed", 1)); dev = "cuda:0"
model = TSPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = TSPEnv(100, dev)
opt = Adam(model.parameters(), lr=cfg["params"]["learning_rate"], weight_decay=1e-6)
kinds = ["uniform", "cluster", "mixed"]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(os.path.join(ROOT, "gpurun_out"))
curve = []


def evaluate():
    tester = TSPLib_Tester(dict(cfg, load_checkpoint=None, name="elg_amd_selftrained"), model=model)
    tester.tsplib_path = os.path.join(gu.GOLDEN_DIR, "tsplib")
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        res, summ = tester.test_on_tsplib()
    model.requires_grad_(True)
    model.train()
    return res, summ


t0 = time.time()
for i in range(STEPS):
    dist = dict(cfg["distribution"], data_type=kinds[i % 3])
    J, rew = train_step(model, env, opt, generate_tsp_data(64, 100, dist), cfg["params"]["scale_norm"])
    if (i + 1) % 5000 == 0:
        torch.cuda.synchronize(); print(f"step {i+1}: {time.time()-t0:.0f} s, sampled cost {float(-rew.mean()):.3f}", flush=True)
    if EVERY and (i + 1) % EVERY == 0 and (i + 1) < STEPS:
        _, summ = evaluate()
        curve.append({"step": i + 1, "seconds": round(time.time() - t0, 1), "gap_percent": summ})
        print(curve[-1], flush=True)
        json.dump({"curve": curve}, open("tsplib_selftrained_curve.json", "w"), indent=1)
torch.cuda.synchronize(); train_s = time.time() - t0
t1 = time.time()
results, summary = evaluate()
eval_s = time.time() - t1
curve.append({"step": STEPS, "seconds": round(train_s, 1), "gap_percent": summary})
out = {"train_steps": STEPS, "train_seconds": round(train_s, 1), "train_instances": STEPS * 64,
       "eval_seconds": round(eval_s, 1), "instances": len(results), "summary_gap_percent": summary,
       "reference_fully_trained_gap_percent": {"total": 3.06, "<=200": 1.12, "200-500": 4.28, "500-1002": 8.87},
       "curve": curve,
       "per_instance": [{"instance": r["instance"], "n": r["record"][0]["scale"], "gap_percent": round(100 * r["record"][0]["gap"], 2)} for r in results]}
json.dump(out, open("tsplib_selftrained_long.json" if EVERY else "tsplib_selftrained.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_instance"}))
