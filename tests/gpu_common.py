"""Helpers shared by the `-m gpu` parity tests: build engine inputs from golden weights/problems."""
from typing import Dict

import numpy as np
import torch

import golden_util as gu
from oracle import elg_oracle as orc
from elg_amd import _lib as L
from elg_amd import engine as eng

DEV = "cuda:0"


def weights(problem, seed, mp, gain=1.0, dtype=torch.float32):
    return {k: torch.from_numpy(v).to(dtype) for k, v in gu.golden_weights(problem, seed, mp, True, gain).items()}


def sub(P, prefix):
    return {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}


import math  # noqa: E402


def fold_decoder_tables(dec: Dict[str, torch.Tensor], enc: torch.Tensor, problem: int) -> Dict[str, torch.Tensor]:
    """Per-instance tables of the pointer decoder (reference models.py:300-352, TSP/models.py:231-270):
    K = Wk enc, V = Wv enc, PK = enc Wc / sqrt(E) (pointer keys with multi_head_combine folded in),
    pb = enc . bc / sqrt(E), Q1/Q2 = the per-node query contributions."""
    Wc, bc = dec["multi_head_combine.weight"], dec["multi_head_combine.bias"]
    t = {
        "K": (enc @ dec["Wk.weight"].T).contiguous(),
        "V": (enc @ dec["Wv.weight"].T).contiguous(),
        "PK": ((enc @ Wc) / math.sqrt(128)).contiguous(),
        "pb": ((enc @ bc) / math.sqrt(128)).contiguous(),
    }
    if problem == L.PROBLEM_CVRP:
        Wq = dec["Wq_last.weight"]
        t["Q1"] = (enc @ Wq[:, :128].T).contiguous()
        t["wl"] = Wq[:, 128].contiguous()
        t["Q2"] = None
    else:
        t["Q1"] = (enc @ dec["Wq_last.weight"].T).contiguous()
        t["Q2"] = (enc @ dec["Wq_first.weight"].T).contiguous()
        t["wl"] = None
    return t


def make_policy(P, cfg, enc_gpu, kind, has_local=True):
    """Fold oracle-format weights (CPU dict) into engine tables on the GPU."""
    Pg = {k: v.to(DEV) for k, v in P.items()}
    tables = fold_decoder_tables(sub(Pg, "decoder."), enc_gpu, kind)
    lp_prefix = "decoder.local_policies.0." if kind == L.PROBLEM_CVRP else "decoder.local_policy_0."
    nfeat = 3 if kind == L.PROBLEM_CVRP else 2
    nslots = cfg.local_size + (1 if kind == L.PROBLEM_CVRP else 0)
    loc = eng.fold_local_tables(sub(Pg, lp_prefix), nfeat, nslots) if has_local else None
    return eng.Policy(tables, loc, cfg.local_size, cfg.xi, cfg.logit_clipping, 1.0 / cfg.ensemble_size,
                      has_local and cfg.ensemble, cfg.distance_penalty)


def make_problem(xy, demand, kind):
    xyg = xy.to(DEV).float().contiguous()
    dg = None if demand is None else demand.to(DEV).float().contiguous()
    return eng.Problem(kind, xyg, dg, eng.nbr_tables(xyg))


def cvrp_fixture(tag):
    fx = gu.load_golden(f"cvrp_rollout_{tag}.npz")
    B, N, M, wseed, pseed, local_size, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.CVRP_MODEL_PARAMS)
    mp["local_size"] = [local_size]
    cfg = orc.ModelCfg.from_model_params(mp, "cvrp")
    P = weights("cvrp", wseed, mp, float(fx["gain"]))
    depot, loc, demand = gu.golden_cvrp_problem(pseed, B, N, float(fx["capacity"]))
    xy = torch.from_numpy(np.concatenate([depot, loc], 1))
    dem = torch.from_numpy(np.concatenate([np.zeros((B, 1), np.float32), demand], 1))
    return fx, cfg, P, xy, dem, B, N, M


def tsp_fixture(tag):
    fx = gu.load_golden(f"tsp_rollout_{tag}.npz")
    B, N, M, wseed, pseed, local_size, rseed = [int(x) for x in fx["meta"]]
    mp = dict(gu.TSP_MODEL_PARAMS)
    mp["local_size"] = [local_size]
    cfg = orc.ModelCfg.from_model_params(mp, "tsp")
    P = weights("tsp", wseed, mp, float(fx["gain"]))
    xy = torch.from_numpy(gu.golden_tsp_problem(pseed, B, N))
    return fx, cfg, P, xy, B, N, M


def rel_err_probs(got, ref, floor=1e-6):
    big = ref > floor
    return float((np.abs(got[big] - ref[big]) / ref[big]).max()) if big.any() else 0.0


def assert_same_mask(got, ref, what=""):
    """Masked nodes (reference probability exactly 0) must get exactly 0; unmasked nodes must not,
    except where the reference value itself is below fp32 normal range (the device exp flushes
    denormals: exp(-100) = 3.7e-44 is a denormal)."""
    assert not (got[ref == 0] != 0).any(), f"{what}: masked node received probability"
    dead = (got == 0) & (ref > 1e-35)
    assert not dead.any(), f"{what}: unmasked node lost its probability ({ref[dead][:5]})"
