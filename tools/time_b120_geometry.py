"""Training-step time at the reference's default batch 120 for a few launch geometries of the cooperative rollout kernel
(tiles = workgroups per instance).  python tools/time_b120_geometry.py [B]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev)
env = CVRPEnv(100, dev); opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
batches = [generate_vrp_data(B, 100, dict(cfg["distribution"], data_type="uniform")) for _ in range(8)]
orig = eng.launch_geometry
for tiles in (0, 1, 2, 3, 4, 5, 7):
    eng.launch_geometry = (lambda t: (lambda B_, M, N1, n_cu=None: orig(B_, M, N1, n_cu) if t == 0 else (8, t, 1)))(tiles)
    for i in range(5): train_step(model, env, opt, batches[i % 8], True, check=False)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(40): train_step(model, env, opt, batches[i % 8], True, check=False)
    torch.cuda.synchronize()
    print(f"B {B} tiles {'default ' + str(orig(B, 100, 101)[1]) if tiles == 0 else tiles}: {(time.time() - t0) / 40 * 1e3:.3f} ms/step")
