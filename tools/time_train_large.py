"""Time one training step (encoder -> sampled rollout -> loss -> backward -> Adam) above 128 nodes, with the decoder backward
over the rows the streaming kernel saved vs through the replay kernel.   python tools/time_train_large.py cvrp 200 32 [M]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from elg_amd import engine as eng  # noqa: E402


def main():
    kind, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    only = sys.argv[4] if len(sys.argv) > 4 and not sys.argv[4].isdigit() else None       # "rows" / "replay": that path only
    M = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else N
    dev = "cuda:0"
    if kind == "cvrp":
        from elg_amd.CVRP.CVRPEnv import CVRPEnv as Env
        from elg_amd.CVRP.CVRPModel import CVRPModel as Model
        from elg_amd.CVRP.train import train_step
        from elg_amd.CVRP.generate_data import generate_vrp_data
        mp = dict(ensemble=True, distance_penalty=True, positional=True, xi=-1, local_size=[40], ensemble_size=1, demand=True,
                  euclidean=False, embedding_dim=128, encoder_layer_num=6, head_num=8, qkv_dim=16, logit_clipping=50,
                  ff_hidden_dim=512, local_att_hidden_dim=32, local_att_head_num=4, local_att_qkv_dim=8)   # config.yml
        batch = lambda: generate_vrp_data(B, N, dict(data_type="uniform"))
    else:
        from elg_amd.TSP.TSPEnv import TSPEnv as Env
        from elg_amd.TSP.TSPModel import TSPModel as Model
        from elg_amd.TSP.train import train_step
        mp = dict(ensemble=True, distance_penalty=True, positional=True, ensemble_size=1, xi=-1, local_size=[30],
                  euclidean=False, embedding_dim=128, encoder_layer_num=6, head_num=8, qkv_dim=16, logit_clipping=50,
                  ff_hidden_dim=512, local_att_hidden_dim=32, local_att_head_num=4, local_att_qkv_dim=8)   # config.yml
        batch = lambda: torch.rand(B, N, 2)
    from elg_amd.optim import Adam
    out = {}
    for path in ((only,) if only else ("rows", "replay")):
        eng.TrainRows._cache.clear()
        torch.cuda.empty_cache()
        eng.LARGE_ROWS_BUDGET = 0.45 if path == "rows" else 0.0
        torch.manual_seed(1)
        np.random.seed(1)
        model = Model(**mp).to(dev).train()
        if hasattr(model.decoder, "add_local_policy"):
            model.decoder.add_local_policy(dev)
        env = Env(multi_width=M, device=dev)
        opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
        ts = []
        for i in range(6):
            b = batch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            J, _ = train_step(model, env, opt, b, True, None, 1, False) if kind == "cvrp" else train_step(model, env, opt, b)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        saved = any(k[2] > 128 for k in eng.TrainRows._cache)
        out[path] = dict(ms=[round(t, 2) for t in ts], saved_rows=saved, loss=float(J),
                         peak_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
        torch.cuda.reset_peak_memory_stats()
    if not only:
        out["speedup"] = round(min(out["replay"]["ms"][2:]) / min(out["rows"]["ms"][2:]), 3)
    print(json.dumps(dict(kind=kind, N=N, B=B, M=M, **out)))


if __name__ == "__main__":
    main()
