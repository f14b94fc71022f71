import sys, os, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from oracle import elg_oracle as orc
from elg_amd import vrplib_io, engine as eng, _lib as L
from elg_amd.CVRP.CVRPEnv import CVRPEnv
from elg_amd.CVRP.CVRPModel import CVRPModel
from elg_amd.CVRP.utils import rollout
dev = "cuda:0"
mp = dict(gu.CVRP_MODEL_PARAMS)
model = CVRPModel(**mp); model.decoder.add_local_policy(dev)
w = gu.golden_weights("cvrp", 17, mp, True, 1.0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); model.to(dev).eval()
for name in sys.argv[1:]:
    inst = vrplib_io.read_instance(os.path.join(gu.GOLDEN_DIR, "vrplib", "X", name + ".vrp"))
    n = inst["node_coord"].shape[0] - 1
    env = CVRPEnv(min(n, 1000), dev)
    env.load_vrplib_problem(inst, aug_factor=8)
    rs, _, _ = env.reset()
    with torch.no_grad():
        model.pre_forward(rs)
        torch.cuda.synchronize(); t0 = time.time()
        acts, _, rew = rollout(model, env, 'greedy')
        torch.cuda.synchronize(); dt = time.time() - t0
    best = float(-rew.max())
    sol = vrplib_io.read_solution(os.path.join(gu.GOLDEN_DIR, "vrplib", "X", name + ".sol"))
    print(f"{name}: N={n} T={acts.shape[2]} rollout {dt:.2f}s best {best:.0f} optimal {sol['cost']:.0f} gap {(best-sol['cost'])/sol['cost']:.3f}", flush=True)
    for b in range(2):
        orc.check_feasible(acts[b, :8].cpu().numpy(), env.depot_node_demand[b, 1:].cpu().numpy())
    # oracle agreement on the first decode steps of 2 augmentations x 4 trajectories
    P = {k: torch.from_numpy(v) for k, v in w.items()}
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    xy = env.depot_node_xy[:2].cpu(); dem = env.depot_node_demand[:2].cpu()
    a = acts[:2, :4].cpu()
    out = orc.rollout_cvrp(P, cfg, xy, dem, 4, starts=a[0, :, 1], forced=a, keep_probs=True, max_steps=8,
                           enc=model.encoded_nodes[:2].cpu())
    agree = np.mean([(p.argmax(-1) == a[:, :, i + 2]).float().mean().item() for i, p in enumerate(out["full_probs"])])
    print("   oracle argmax agreement (first 6 decode steps):", agree, flush=True)
