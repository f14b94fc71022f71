"""In-kernel phase clock of the fused encoder kernels (csrc/elg_enc_fused.hip): builds a DIAGNOSTIC copy of the library with
-DELG_STAMPS (s_memtime at the phase boundaries), runs the encoder forward + backward at the bench shape and prints, per kernel
and wave, the mean cycles between consecutive stamps.  The shipped library executes no stamp; never quote this build's run time.
    python tools/stamp_enc.py build      (in the build container: writes tools/_diag/libelg_hip_stamps.so)
    python tools/stamp_enc.py            (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "_diag", "libelg_hip_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from elg_amd import build as b
    objs = []
    for src in b.SOURCES:
        obj = os.path.join("/tmp", "stamps_" + src.replace(".hip", ".o"))
        if src == "elg_enc_fused.hip" or not os.path.exists(obj):
            subprocess.check_call([b._hipcc(), *b.FLAGS, "-DELG_STAMPS", "-c", os.path.join(b.CSRC, src), "-o", obj])
        objs.append(obj)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    print(LIB)
    sys.exit(0)
os.environ["ELG_HIP_LIB"] = LIB
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C, torch
import golden_util as gu
from elg_amd import _lib as L, encoder as enc_host
B, N1 = 64, 101
mp = dict(gu.CVRP_MODEL_PARAMS)
names = enc_host.parameter_names(L.PROBLEM_CVRP, 6)
W = gu.golden_weights("cvrp", 1, mp, True)
params = [torch.from_numpy(W[n]).cuda().requires_grad_(True) for n in names]
xy = torch.rand(B, N1, 2, device="cuda"); dem = torch.rand(B, N1, device="cuda")
NK, NWG, NW, NS = 6, 2048, 8, 16
buf = torch.zeros(NK * NWG * NW * NS, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.elg_enc_debug_stamps.argtypes = [C.c_void_p]
def both():
    enc, t = enc_host.encode_and_fold(L.PROBLEM_CVRP, xy, dem, params, 6, 512)
    (t["K"].sum() + t["V"].sum() + t["PK"].sum() + t["Q1"].sum() + t["pb"].sum()).backward()
for _ in range(3): both()
torch.cuda.synchronize()
assert lib.elg_enc_debug_stamps(C.c_void_p(buf.data_ptr())) == 0
both(); torch.cuda.synchronize()
st = buf.view(NK, NWG, NW, NS).cpu().double()
for kid, name, nwg in ((0, "f1", 256), (1, "f2", 256), (3, "b1", 256), (4, "b2", 256)):
    s = st[kid, :nwg]
    if s.abs().sum() == 0: continue
    for w in (0, 3, 6):
        v = s[:, w]
        n = int((v[0] > 0).sum())
        d = (v[:, 1:n] - v[:, :n - 1]).mean(0)
        print(f"{name} wave {w}: total {float((v[:, n - 1] - v[:, 0]).mean()):.0f} cyc;", " ".join(f"{i}->{i + 1}:{float(x):.0f}" for i, x in enumerate(d)))
    # spread of the workgroups' start / end (dispatch skew), in cycles of the first workgroup's clock
    t0 = s[:, 0, 0]
    print(f"{name}: start skew over workgroups {float(t0.max() - t0.min()):.0f} cyc")
