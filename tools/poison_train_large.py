"""Uninitialised-read hunt: the large-instance training step of tests/test_gpu_train_large.py with the caching allocator's
free blocks filled with a poison value first (every torch.empty the step makes then starts as poison).  Prints, per
parameter, the gradient error against the oracle as a fraction of the test's limit.
    python tools/poison_train_large.py cvrp 150 8 2 replay nan [seed]"""
import random
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import golden_util as gu  # noqa: E402
import gpu_common as gc  # noqa: E402
from oracle import elg_oracle as orc  # noqa: E402
from elg_amd import engine as eng  # noqa: E402


def poison(value, gb=6):
    blocks = [torch.full((256 * 2 ** 20,), value, device="cuda:0") for _ in range(gb)]
    small = [torch.full((n,), value, device="cuda:0") for n in (2 ** 10, 2 ** 14, 2 ** 18, 2 ** 20, 2 ** 22) for _ in range(8)]
    torch.cuda.synchronize()
    del blocks, small


def main():
    problem, N, M, B, path, pv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    value = float(pv)
    seed = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    eng.LARGE_ROWS_BUDGET = 0.45 if path == "rows" else 0.0
    if problem == "cvrp":
        from elg_amd.CVRP.CVRPEnv import CVRPEnv as Env
        from elg_amd.CVRP.train import pomo_loss
        from elg_amd.CVRP.utils import rollout
        mp = dict(gu.CVRP_MODEL_PARAMS)
        depot, loc_xy, demand = gu.golden_cvrp_problem(31 + N, B, N, 80.0)
        xy = torch.from_numpy(np.concatenate([depot, loc_xy], 1))
        dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1))
        batch = dict(depot=torch.from_numpy(depot), loc=torch.from_numpy(loc_xy), demand=torch.from_numpy(demand))
    else:
        from elg_amd.TSP.TSPEnv import TSPEnv as Env
        from elg_amd.TSP.train import pomo_loss
        from elg_amd.TSP.utils import rollout
        mp = dict(gu.TSP_MODEL_PARAMS)
        xy, dem = torch.from_numpy(gu.golden_tsp_problem(31 + N, B, N)), None
        batch = xy.clone()
    cfg = orc.ModelCfg.from_model_params(mp, problem)
    ref = None
    for rep in range(4):
        model = gc.load_model(problem, 17, mp, gain=1.0).train()
        env = Env(multi_width=M, device="cuda:0")
        if rep:
            poison(value)
        env.load_random_problems(batch)
        rs, _, _ = env.reset()
        model.pre_forward(rs)
        torch.manual_seed(5)
        random.seed(seed)
        acts, probs, rew = rollout(model, env, 'sample')
        torch.manual_seed(6)
        rew_n = rew + 0.3 * torch.randn(B, M, device=rew.device)
        J = pomo_loss(probs, rew_n, True)
        J.backward()
        got = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
        if ref is None:
            P = {k: v.clone().requires_grad_(True) for k, v in gc.weights(problem, 17, mp, 1.0).items()}
            a = acts.cpu()
            out = (orc.rollout_cvrp(P, cfg, xy, dem, M, starts=a[0, :, 1], forced=a) if problem == "cvrp"
                   else orc.rollout_tsp(P, cfg, xy, M, starts=a[0, :, 0], forced=a))
            Jo = orc.pomo_loss(out["probs"], rew_n.cpu(), True, guard_zero=(problem == "tsp"))
            Jo.backward()
            ref = {k: v.grad for k, v in P.items()}
            ref_acts = a
        assert torch.equal(acts.cpu(), ref_acts), "tours differ between repetitions"
        rms = max(float(v.norm()) / np.sqrt(v.numel()) for v in ref.values())
        bad = []
        for k, r in ref.items():
            err = float((got[k] - r).abs().max())
            lim = 2e-3 * float(r.abs().max()) + 2e-3 * rms
            frac = err / lim if np.isfinite(err) else float("inf")
            if not frac < 0.6:
                bad.append((k, round(frac, 3)))
                d = (got[k] - r).abs()
                if d.dim() == 2 and rep == 0:       # one hidden unit (a ReLU at its kink) or spread over the matrix?
                    rows = d.max(dim=1)[0]
                    top = torch.topk(rows, 3)
                    print("   ", k, "row-wise max error: top", [(int(i), float(v)) for v, i in zip(top[0], top[1])],
                          "median row", float(rows.median()), flush=True)
        print(f"seed {seed} rep {rep} poison {'-' if rep == 0 else pv}: {len(bad)} parameters over 0.6 of the limit", bad[:12], flush=True)


if __name__ == "__main__":
    main()
