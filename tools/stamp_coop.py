"""In-kernel phase clock of rollout_fwd_coop_kernel: builds a DIAGNOSTIC copy of the library with -DELG_STAMPS (s_memtime at the
phase boundaries, segment sums through elg_rollout_args.scratch), runs one training rollout at the bench shape and prints the
share of a step every segment takes.  The shipped library executes no stamp; never quote this build's run time.
    python tools/stamp_coop.py build      (in the build container: writes tools/_diag/libelg_hip_stamps.so -- a diagnostic build, never in the package)
    python tools/stamp_coop.py            (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "_diag", "libelg_hip_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from elg_amd import build as b
    objs = []
    for src in b.SOURCES:
        obj = os.path.join("/tmp", "stamps_" + src.replace(".hip", ".o"))
        subprocess.check_call([b._hipcc(), *b.FLAGS, "-DELG_STAMPS", "-c", os.path.join(b.CSRC, src), "-o", obj])
        objs.append(obj)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    print(LIB)
    sys.exit(0)
os.environ["ELG_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import ctypes as C, torch, yaml
from elg_amd import _lib as L, engine as eng
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.utils import seed_everything
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev)
env.load_random_problems(generate_vrp_data(64, 100, dict(cfg["distribution"], data_type="uniform")))
rs, _, _ = env.reset()
with torch.no_grad():
    model.pre_forward(rs)
pol = model.decoder.policy
starts = torch.tensor(model.draw_starts(100, 100), dtype=torch.int32)
dbg = torch.zeros(256 * 8 * 16, device=dev)
orig = L.lib().elg_rollout_fwd
def hooked(a, stream):
    a._obj.scratch = C.c_void_p(dbg.data_ptr())
    return orig(a, stream)
L.lib().elg_rollout_fwd = hooked
for _ in range(2):
    res = eng.rollout_forward(env.problem, pol, 100, starts, L.MODE_SAMPLE, seed=1234, train=True)
torch.cuda.synchronize()
acc = dbg.view(256, 8, 16).cpu()
names = ["glimpse+head", "barrier 1", "pointer || tail", "barrier 2", "finish: slot scatter", "finish: clip/softmax", "finish: choose",
         "finish: prob + row stores", "advance: loads, transition, mask", "advance: query row", "advance: k-NN walk",
         "advance: slot features", "K/V operand reload", "barrier 3"]
T = float(res.tlen.max())
for w in (0, 3, 6, 7):
    a = acc[:, w, :14].mean(0)
    tot = a.sum()
    print(f"wave {w}: {tot / T:.0f} cycles per step (max T {T:.0f});", "  ".join(f"{n} {100 * v / tot:.1f}%" for n, v in zip(names, a)))
a = acc[:, :, :14].mean((0, 1))
print("all waves, cycles per step:", {n: round(float(v) / T) for n, v in zip(names, a)})
