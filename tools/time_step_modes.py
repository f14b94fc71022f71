"""Training-step time at the bench shape for a few settings of the side-stream split (ELG_LOCAL_BWD_GRID is read at import:
run once per setting), f32 and bf16 modes.  python tools/time_step_modes.py [steps]"""
import os, sys, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd import engine as eng
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.train import train_step
from elg_amd.CVRP.utils import seed_everything
from elg_amd.optim import Adam
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev)
env = CVRPEnv(100, dev); opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
batches = [generate_vrp_data(64, 100, dict(cfg["distribution"], data_type="uniform")) for _ in range(8)]
for prec in (0, 1):
    eng.FWD_PRECISION = prec
    for i in range(10): train_step(model, env, opt, batches[i % 8], True, check=False)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(steps): train_step(model, env, opt, batches[i % 8], True, check=False)
    torch.cuda.synchronize()
    print(f"grid {os.environ.get('ELG_LOCAL_BWD_GRID', 'default')} side {os.environ.get('ELG_SIDE_LOCAL_BWD', '1')} precision {prec}: {(time.time() - t0) / steps * 1e3:.3f} ms/step")
