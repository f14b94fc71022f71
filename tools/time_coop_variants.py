"""Times the fused rollout at the bench shape (CVRP-100, B = 64, pomo 100, sampled; with and without saved rows) with HIP events.
A/B of two builds of the library: ELG_HIP_LIB=/path/to/other/libelg_hip.so python tools/time_coop_variants.py (same seeds ->
same tours).   python tools/time_coop_variants.py [B] [reps]"""
import os, sys, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elg_amd import _lib as L, engine as eng
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.generate_data import generate_vrp_data
from elg_amd.CVRP.utils import seed_everything
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = yaml.safe_load(open(os.path.join(ROOT, "elg_amd/CVRP/config.yml")))
seed_everything(924); dev = "cuda:0"
model = CVRPModel(**cfg["model_params"]); model.decoder.add_local_policy(dev); model.to(dev).train()
env = CVRPEnv(100, dev)
env.load_random_problems(generate_vrp_data(B, 100, dict(cfg["distribution"], data_type="uniform")))
rs, _, _ = env.reset()
with torch.no_grad():
    model.pre_forward(rs)
pol = model.decoder.policy
starts = torch.tensor(model.draw_starts(100, 100), dtype=torch.int32)
out = {}
for train in (True, False):
    for variant, geo in ((0, None),):
        def run():
            return eng.rollout_forward(env.problem, pol, 100, starts, L.MODE_SAMPLE, seed=1234, train=train, variant=variant, geometry=geo)
        for _ in range(3): res = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): res = run()
        e1.record(); torch.cuda.synchronize()
        if variant == 0: ref = res
        else: print('   same tours as variant 0:', bool(torch.equal(ref.actions, res.actions)), 'probs max diff', float((ref.probs - res.probs).abs().max()))
        print(f"train={train} variant={variant} geometry={geo}: {e0.elapsed_time(e1) / reps:.3f} ms  mean T {res.tlen.float().mean().item():.1f}  "
              f"mean cost {(-res.reward).mean().item():.4f}")
