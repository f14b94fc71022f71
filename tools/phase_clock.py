"""Per-phase clock totals of the cooperative rollout kernel (temporary instrumentation: dump_T = -7)."""
import os, sys, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from elg_amd import engine as eng, _lib as L
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.utils import seed_everything
import ctypes as C
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev)
batch = generate_vrp_data(64, 100, dict(cfg["distribution"], data_type="uniform"))
env.load_random_problems(batch); rs, _, _ = env.reset()
train = (len(sys.argv) > 1 and sys.argv[1] == "train")
with torch.no_grad():
    model.pre_forward(rs)
pol = model.decoder.policy
starts = torch.tensor(model.draw_starts(100, 100), dtype=torch.int32)
orig = L.lib().elg_rollout_fwd
buf = torch.zeros(64, dtype=torch.int64, device=dev)
class Hook:
    def __call__(self, aref, stream):
        a = aref._obj
        a.full_probs, a.dump_T = C.c_void_p(buf.data_ptr()), -7
        return orig(aref, stream)
L.lib().elg_rollout_fwd = Hook()
for it in range(3):
    buf.zero_()
    res = eng.rollout_forward(env.problem, pol, 100, starts, L.MODE_SAMPLE, seed=it, train=train)
    torch.cuda.synchronize()
names = ["glimpse", "barrier1", "pointer|local", "barrier2", "finish4", "advance4", "kv-load", "barrier3"]
v = buf.cpu().view(8, 8).double()
tot = v.sum(1)
print("mean T", float(res.tlen.float().mean()), "max T", int(res.tlen.max()), "train", train)
nwg = 256
print("per-workgroup totals (us, assuming a 100 MHz s_memtime):", "  ".join(f"{n} {float(v[5, i]) * 0.01 / nwg:.0f}" for i, n in enumerate(names)), " sum", f"{float(tot[5]) * 0.01 / nwg:.0f}")
for w in (0, 5, 6, 7):
    print(f"wave {w}: " + "  ".join(f"{n} {100 * float(v[w, i] / tot[w]):.1f}%" for i, n in enumerate(names)))
