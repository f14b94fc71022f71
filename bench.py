#!/usr/bin/env python3
"""Headline benchmark: CVRP-100 REINFORCE/POMO training throughput (instances/s), BASELINE.json configs[1]
(batch 64 per GPU, pomo 100, joint model with the local policy attached), synthetic uniform instances,
random-init weights.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = generate batch -> H2D -> neighbour tables -> encoder -> sampled POMO rollout (one persistent HIP
launch) -> loss kernel -> backward (row prep, MFMA glimpse / local-policy backward kernels, three batched GEMMs,
encoder autograd) -> [RCCL all-reduce of the packed gradient] -> one-launch Adam.  Rank 0 prints ONE JSON line."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_NODES, POMO, LOCAL_BATCH = 100, 100, 64
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)


def model_params():
    import yaml
    with open(os.path.join(ROOT, "elg_amd", "CVRP", "config.yml")) as f:
        return yaml.load(f.read(), Loader=yaml.FullLoader)


def algorithmic_bytes_per_decode_step(B, M, N1, d=128, s=4):
    """per decode step, every instance's K/V/pointer-key tables + coordinates/demand read once,
    plus per-trajectory state in/out (cur, load, visited bitmask; selected, prob, load, bitmask, finished)."""
    words = (N1 + 31) // 32
    r = 4 + 4 + 4 * words
    w = 4 + 4 + 4 + 4 * words + 1
    return B * (3 * N1 * d * s + 12 * N1) + B * M * (r + w)


def cpu_baseline(cfg, seconds_hint=20.0):
    """The oracle (CPU restatement of the reference, torch eager, autograd tape) timed on a bounded sample of the
    same workload: one full training step (encoder, sampled rollout, loss, backward, Adam) at CVRP-100, pomo 100,
    batch 8.  Test infrastructure used as the reported baseline only -- never on the product path."""
    from oracle import elg_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as gu
    Bc = 8
    torch.manual_seed(0)
    mp = cfg["model_params"]
    ocfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gu.golden_weights("cvrp", 1, mp, True).items()}
    xy = torch.rand(Bc, N_NODES + 1, 2)
    dem = torch.cat([torch.zeros(Bc, 1), torch.randint(1, 10, (Bc, N_NODES)).float() / 50.0], 1)
    opt = torch.optim.Adam(list(P.values()), lr=1e-4, weight_decay=1e-6)
    threads = torch.get_num_threads()
    t0 = time.time()
    uni = torch.rand(Bc, POMO, 2 * (N_NODES + 1))
    out = orc.rollout_cvrp(P, ocfg, xy, dem, POMO, starts=torch.randperm(N_NODES)[:POMO], mode="sample", uniforms=uni)
    J = orc.pomo_loss(out["probs"], out["reward"])
    opt.zero_grad()
    J.backward()
    opt.step()
    dt = time.time() - t0
    return {"value": round(Bc / dt, 4), "unit": "instances/s", "cores": threads, "kind": "port",
            "sample": f"1 full training step, CVRP-100 batch={Bc} pomo={POMO} fp32, oracle/elg_oracle.py on "
                      f"{threads} torch threads ({dt:.1f} s, T={out['actions'].shape[2]})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from elg_amd import parallel
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"

    from elg_amd import engine as eng
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.CVRPModel import CVRPModel
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.CVRP.utils import seed_everything

    cfg = model_params()
    seed_everything(cfg["seed"] + rank)
    model = CVRPModel(**cfg["model_params"])
    model.decoder.add_local_policy(dev)            # steady-state regime of joint training (after step T)
    model.to(dev)
    parallel.broadcast_parameters(model)
    env = CVRPEnv(multi_width=POMO, device=dev)
    from elg_amd.optim import Adam
    opt = Adam(model.parameters(), lr=cfg["params"]["learning_rate"], weight_decay=1e-6)
    bucket = parallel.GradBucket(model.parameters(), opt) if world > 1 else None
    dist_cfg = dict(cfg["distribution"], data_type="uniform")

    # time the persistent rollout kernel with HIP events on the launch stream (torch's current stream)
    fwd_events, fwd_steps = [], []
    orig_fwd = eng.rollout_forward

    def timed_fwd(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        res = orig_fwd(*a, **k)
        e1.record()
        fwd_events.append((e0, e1))
        fwd_steps.append(res.tlen)
        return res
    eng.rollout_forward = timed_fwd

    def one_step():
        batch = generate_vrp_data(LOCAL_BATCH, N_NODES, dist_cfg)
        model.train()
        return train_step(model, env, opt, batch, cfg["params"]["scale_norm"], bucket, world, check=True)

    for _ in range(args.warmup):
        one_step()
    fwd_events.clear()
    fwd_steps.clear()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        kern_ms = sum(a.elapsed_time(b) for a, b in fwd_events) / len(fwd_events)
        mean_T = float(torch.stack([t.float().mean() for t in fwd_steps]).mean().item())
        max_T = float(torch.stack([t.float().max() for t in fwd_steps]).mean().item())
        bytes_launch = algorithmic_bytes_per_decode_step(LOCAL_BATCH, POMO, N_NODES + 1) * mean_T
        traffic = None                       # measured HBM bytes per launch (PMC passes, see profiles/)
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "CVRP-100 train instances/sec", "value": round(LOCAL_BATCH * world * args.steps / dt, 2),
            "unit": "instances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic uniform CVRP-100, random-init weights",
            "config": {"workload": "CVRP-100 batch=64/GPU pomo=100 joint (local policy on), BASELINE configs[1]",
                       "global_batch": LOCAL_BATCH * world, "pomo": POMO, "problem_size": N_NODES,
                       "parallelism": f"dp{world}"},
            "roofline": {"kernel": "rollout_fwd_coop_kernel (persistent decode: all steps of all trajectories)",
                         "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_launch),
                         "launch_ms": round(kern_ms, 4), "decode_steps_mean": round(mean_T, 2),
                         "decode_steps_max": round(max_T, 2),
                         "algorithmic_MB_per_decode_step": round(algorithmic_bytes_per_decode_step(
                             LOCAL_BATCH, POMO, N_NODES + 1) / 1e6, 3)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    parallel.barrier()


if __name__ == "__main__":
    main()
