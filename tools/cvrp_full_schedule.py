"""The reference's own CVRP-100 training schedule through elg_amd/CVRP/train.py (reference CVRP/config.yml:15-19: 250 000 steps x
batch 120, `mixed: True` curriculum, the local policy joins at step T = 200 000; CVRP/train.py:83-148), with `validate()` every
`log_step` on the three validation sets the reference opens (tests/golden/r05_cvrp_val100_sets.npz, made by
tools/make_golden_r05.py), reported as gaps against the reference's solver means (train.py:146), then the VRPLIB-X evaluation
(test_vrplib.py).  Writes gpurun_out/cvrp_full_schedule.json (+ the final checkpoint).
    python tools/cvrp_full_schedule.py [train_steps] [T] [log_step]"""
import io, contextlib, json, os, pickle, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, yaml
import golden_util as gu
from elg_amd.CVRP import train as T_
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.utils import seed_everything
from elg_amd.CVRP.test_vrplib import VRPLib_Tester
from elg_amd import parallel
parallel.respect_cpu_quota()            # as train.py's entry point does

cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
p = cfg["params"]
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else p["train_steps"]
TJOIN = int(sys.argv[2]) if len(sys.argv) > 2 else p["T"]
LOG = int(sys.argv[3]) if len(sys.argv) > 3 else p["log_step"]
OPTS = np.array([15.740834, 7.909336, 14.294179])        # reference train.py:146
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
# the validation pickles, in the reference's format, in a scratch data/ directory
data_dir = tempfile.mkdtemp()
z = np.load(os.path.join(gu.GOLDEN_DIR, "r05_cvrp_val100_sets.npz"))
for kind in ("uniform", "cluster", "mixed"):
    rows = [(z[f"{kind}_depot"][i].astype(np.float64).tolist(), z[f"{kind}_loc"][i].astype(np.float64).tolist(),
             z[f"{kind}_demand"][i].astype(np.float64).tolist(), float(z[f"{kind}_capacity"][i])) for i in range(1000)]
    pickle.dump(rows, open(os.path.join(data_dir, f"vrp_{kind}100_1000_seed1234.pkl"), "wb"))

dev = "cuda:0"
seed_everything(cfg["seed"])
model = CVRPModel(**cfg["model_params"]).to(dev)
curve, t0 = [], time.time()


class FileLog:                                             # train() logs validate()'s list here
    def log(self, info):
        torch.cuda.synchronize()
        gaps = (np.array(info) - OPTS) / OPTS
        curve.append({"step": len(curve) * LOG + LOG, "seconds": round(time.time() - t0, 1), "val_cost": [round(float(v), 4) for v in info],
                      "gap_percent_vs_reference_solver_means": [round(100 * float(g), 3) for g in gaps]})
        print(curve[-1], flush=True)
        json.dump({"curve": curve}, open(os.path.join(out_dir, "cvrp_full_schedule_curve.json"), "w"), indent=1)


_validate = T_.validate
T_.validate = lambda model, mw, device, mixed=True, data_dir_=data_dir: _validate(model, mw, device, mixed, data_dir_)
ck = tempfile.mkdtemp()
T_.train(model=model, training=cfg["training"], T=TJOIN, start_steps=0, train_steps=STEPS - 1, mixed=p["mixed"],
         train_batch_size=p["train_batch_size"], problem_size=p["problem_size"], distribution=cfg["distribution"],
         multiple_width=p["multiple_width"], lr=p["learning_rate"], device=dev, logger=None, scale_norm=p["scale_norm"],
         fileLogger=FileLog(), dir_path=ck, log_step=LOG)
torch.cuda.synchronize(); train_s = time.time() - t0
torch.save({"step": STEPS, "model_state_dict": model.state_dict()}, os.path.join(out_dir, "cvrp_full_schedule_final.pt"))
t1 = time.time()
tester = VRPLib_Tester(dict(cfg, load_checkpoint=None, name="elg_amd_full_schedule"), model=model)
tester.vrplib_path = os.path.join(gu.GOLDEN_DIR, "vrplib", "X") + "/"
with contextlib.redirect_stdout(io.StringIO()):
    res, summ = tester.test_on_vrplib()
out = {"schedule": {"train_steps": STEPS, "train_batch_size": p["train_batch_size"], "T_joint": TJOIN, "mixed": p["mixed"], "log_step": LOG,
                    "learning_rate": p["learning_rate"], "seed": cfg["seed"]},
       "train_seconds": round(train_s, 1), "train_instances": STEPS * p["train_batch_size"],
       "instances_per_second_incl_validation": round(STEPS * p["train_batch_size"] / train_s, 1),
       "reference_solver_means": OPTS.tolist(), "curve": curve, "final_validation": curve[-1] if curve else None,
       "vrplib_X_eval_seconds": round(time.time() - t1, 1), "vrplib_X_summary_gap_percent": summ,
       "per_instance": [{"instance": r["instance"], "n": r["record"][0]["scale"], "gap_percent": round(100 * r["record"][0]["gap"], 2)} for r in res]}
json.dump(out, open(os.path.join(out_dir, "cvrp_full_schedule.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("per_instance", "curve")}))
