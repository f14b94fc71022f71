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


def position_encoding(Lslots: int, emb: int, device) -> torch.Tensor:
    """Sinusoid table of the local policy, sin block then cos block (reference models.py:28-49)."""
    nts = emb // 2
    inc = math.log(10000.0) / max(nts - 1, 1)
    inv = torch.exp(torch.arange(nts, dtype=torch.float32, device=device) * -inc)
    pos = torch.arange(Lslots, dtype=torch.float32, device=device)
    st = pos[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def fold_local_tables(lp: Dict[str, torch.Tensor], nfeat: int, n_slots: int, pe_scale: float = 1.0) -> torch.Tensor:
    """Fold local_policy_att's projections into slot tables (layout: include/elg_hip.h ELG_LOC_*).

    reference models.py:133-166:  e_j = We f_j + be + PE[j];  q = Wq c;  k_j = Wk e_j;  v_j = Wv e_j;
    u_j = (Wc softmax(q k / sqrt 8) v + bc) . e_j / sqrt 32.  Everything that does not depend on the
    features f_j is precomputed per slot j here."""
    We, be = lp["init_emb.weight"], lp["init_emb.bias"]
    dev = We.device
    pe = position_encoding(n_slots, 32, dev).to(We.dtype) * pe_scale
    base = be[None, :] + pe                                        # (L,32)  be + PE[j]
    q = (lp["Wq.weight"] @ lp["cur_token_emb"]).view(4, 8)
    WkWe = (lp["Wk.weight"] @ We).view(4, 8, nfeat)
    la = torch.einsum("hd,hdf->hf", q, WkWe) / math.sqrt(8)      # (4,F)
    kb = (base @ lp["Wk.weight"].T).view(n_slots, 4, 8)
    lt = torch.einsum("hd,jhd->jh", q, kb) / math.sqrt(8)        # (L,4)
    lAv = lp["Wv.weight"] @ We                                     # (32,F)
    lcv = base @ lp["Wv.weight"].T                                 # (L,32)
    lWc = lp["multi_head_combine.weight"]
    lbc = lp["multi_head_combine.bias"]
    lWe = We / math.sqrt(32)
    lpe = base / math.sqrt(32)

    def pad_cols(x, cols):
        return torch.nn.functional.pad(x, (0, cols - x.shape[1]))

    def pad_rows(x, rows):
        return torch.nn.functional.pad(x, (0, 0, 0, rows - x.shape[0]))
    pieces = [
        torch.nn.functional.pad(pad_cols(la, 3).reshape(-1), (0, 4)),          # LA   16
        pad_rows(lt, 64).reshape(-1),                                   # LT   256
        pad_cols(lAv, 3).reshape(-1),                                           # LAV  96
        pad_rows(lcv, 64).reshape(-1),                                  # LCV  2048
        lWc.reshape(-1),                                                        # LWC  1024
        lbc.reshape(-1),                                                        # LBC  32
        pad_cols(lWe, 3).reshape(-1),                                           # LWE  96
        pad_rows(lpe, 64).reshape(-1),                                  # LPE  2048
    ]
    out = torch.cat(pieces).contiguous()
    assert out.numel() == 5616
    return out


def make_policy(P, cfg, enc_gpu, kind, has_local=True):
    """Fold oracle-format weights (CPU dict) into engine tables on the GPU."""
    Pg = {k: v.to(DEV) for k, v in P.items()}
    tables = fold_decoder_tables(sub(Pg, "decoder."), enc_gpu, kind)
    lp_prefix = "decoder.local_policies.0." if kind == L.PROBLEM_CVRP else "decoder.local_policy_0."
    nfeat = 3 if kind == L.PROBLEM_CVRP else 2
    nslots = cfg.local_size + (1 if kind == L.PROBLEM_CVRP else 0)
    loc = fold_local_tables(sub(Pg, lp_prefix), nfeat, nslots) if has_local else None
    return eng.Policy(tables, loc, cfg.local_size, cfg.xi, cfg.logit_clipping, 1.0 / cfg.ensemble_size,
                      has_local and cfg.ensemble, cfg.distance_penalty, bool(getattr(cfg, 'euclidean', False)))


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


def record_parity(name: str, value: float):
    """Keep the worst observed error of a parity test: merged into gpurun_out/parity_r06.json (copied to profiles/)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_r06.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[name] = max(float(value), float(cur.get(name, 0.0)))
        with open(path, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
    except OSError:
        pass


def load_model(problem, wseed, mp, gain=1.0, local=True):
    """elg_amd model on the GPU carrying the golden weights (the state_dict names are the reference's)."""
    if problem == "cvrp":
        from elg_amd.CVRP.CVRPModel import CVRPModel as Model
    else:
        from elg_amd.TSP.TSPModel import TSPModel as Model
    model = Model(**mp)
    if local:
        model.decoder.add_local_policy(DEV)
    w = gu.golden_weights(problem, wseed, mp, local, gain)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return model.to(DEV).eval()
