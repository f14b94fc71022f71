"""In-kernel phase clock of rollout_fwd_mt_kernel (TSP-500, B = 16, pomo 500, greedy) on the DIAGNOSTIC build (-DELG_STAMPS, made by
`python tools/stamp_coop.py build`); the stamp sums leave through the `uniforms` pointer, which a greedy rollout does not read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ELG_HIP_LIB"] = os.path.join(ROOT, "tools", "_diag", "libelg_hip_stamps.so")
sys.path.insert(0, ROOT)
import ctypes as C, torch, yaml
from elg_amd import _lib as L, engine as eng
from elg_amd.TSP.TSPEnv import TSPEnv
from elg_amd.TSP.TSPModel import TSPModel
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
kind = sys.argv[2] if len(sys.argv) > 2 else "tsp"
torch.manual_seed(1)
if kind == "tsp":
    cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/TSP/config.yml")))
    tm = TSPModel(**cfg["model_params"]); tm.decoder.add_local_policy(dev); tm.to(dev).eval()
    env = TSPEnv(multi_width=N, device=dev)
    env.load_random_problems(torch.rand(16, N, 2))
else:
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.CVRPModel import CVRPModel
    cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
    tm = CVRPModel(**cfg["model_params"]); tm.decoder.add_local_policy(dev); tm.to(dev).eval()
    env = CVRPEnv(N, dev)
    env.load_random_problems(dict(loc=torch.rand(8, N, 2), depot=torch.rand(8, 2), demand=torch.randint(1, 10, (8, N)).float() / 100.0))
rs, _, _ = env.reset()
dbg = torch.zeros(4096 * 8 * 16, device=dev)
orig = L.lib().elg_rollout_fwd
def hooked(a, stream):
    a._obj.uniforms = C.c_void_p(dbg.data_ptr())
    return orig(a, stream)
L.lib().elg_rollout_fwd = hooked
with torch.no_grad():
    tm.pre_forward(rs)
    starts = torch.arange(N, dtype=torch.int32) if kind == "tsp" else torch.arange(1, N + 1, dtype=torch.int32)
    for _ in range(2):
        res = eng.rollout_forward(env.problem, tm.decoder.policy, N, starts, L.MODE_GREEDY)
torch.cuda.synchronize()
nb = int((dbg.view(-1, 8, 16).abs().sum((1, 2)) > 0).sum())
acc = dbg.view(-1, 8, 16)[:nb].cpu()
names = ["owners: prepare (query, stores)", "barrier 1", "glimpse", "barrier 2", "pointer || local policy", "barrier 3",
         "choose: slot-word restore", "owners: advance", "loop barrier", "prepare: mask build", "prepare: k-NN walk + slot features",
         "choose: slot terms", "choose: the pass", "choose: lane merge", "choose: choice + probability"]
a = acc[:, :, :15].mean((0, 1)); tot = float(a.sum())
T = float(res.tlen.max())
print(f"{nb} workgroups, {tot / T:.0f} cycles per step (max T {T:.0f});", "  ".join(f"{n} {100 * float(v) / tot:.1f}%" for n, v in zip(names, a)))
for w in (0, 7):
    a = acc[:, w, :15].mean(0); tot = float(a.sum())
    print(f"wave {w}:", "  ".join(f"{100 * float(v) / tot:.1f}%" for v in a))
