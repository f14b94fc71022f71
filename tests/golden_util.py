"""Deterministic weights / problems shared by tools/make_golden.py (which feeds them to the real
reference) and by the tests (which feed the same values to the oracle and to the HIP engine).

Weights are drawn from numpy's frozen legacy ``RandomState`` stream, so a fixture only needs to
store a seed: no 5 MB state-dicts in git.  Shapes/names follow the reference's ``state_dict``
layout."""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CVRP_MODEL_PARAMS = dict(
    ensemble=True, distance_penalty=True, positional=True, xi=-1, local_size=[40], ensemble_size=1,
    demand=True, euclidean=False, embedding_dim=128, encoder_layer_num=6, head_num=8, qkv_dim=16,
    logit_clipping=50, ff_hidden_dim=512, local_att_hidden_dim=32, local_att_head_num=4,
    local_att_qkv_dim=8)

TSP_MODEL_PARAMS = dict(
    ensemble=True, distance_penalty=True, positional=True, ensemble_size=1, xi=-1, local_size=[30],
    euclidean=False, embedding_dim=128, encoder_layer_num=6, head_num=8, qkv_dim=16,
    logit_clipping=50, ff_hidden_dim=512, local_att_hidden_dim=32, local_att_head_num=4,
    local_att_qkv_dim=8)


def model_param_shapes(problem: str, mp: dict, local: bool = True) -> "OrderedDict[str, tuple]":
    E, H, dk, F = mp["embedding_dim"], mp["head_num"], mp["qkv_dim"], mp["ff_hidden_dim"]
    L = mp["encoder_layer_num"]
    le, lh, ldk = mp["local_att_hidden_dim"], mp["local_att_head_num"], mp["local_att_qkv_dim"]
    s: "OrderedDict[str, tuple]" = OrderedDict()
    if problem == "cvrp":
        s["encoder.embedding_depot.weight"] = (E, 2)
        s["encoder.embedding_depot.bias"] = (E,)
        s["encoder.embedding_node.weight"] = (E, 3)
        s["encoder.embedding_node.bias"] = (E,)
        n1, n2, ff = "add_n_normalization_1", "add_n_normalization_2", "feed_forward"
    else:
        s["encoder.embedding.weight"] = (E, 2)
        s["encoder.embedding.bias"] = (E,)
        n1, n2, ff = "addAndNormalization1", "addAndNormalization2", "feedForward"
    for l in range(L):
        p = f"encoder.layers.{l}."
        for w in ("Wq", "Wk", "Wv"):
            s[p + w + ".weight"] = (H * dk, E)
        s[p + "multi_head_combine.weight"] = (E, H * dk)
        s[p + "multi_head_combine.bias"] = (E,)
        s[p + n1 + ".norm.weight"] = (E,)
        s[p + n1 + ".norm.bias"] = (E,)
        s[p + ff + ".W1.weight"] = (F, E)
        s[p + ff + ".W1.bias"] = (F,)
        s[p + ff + ".W2.weight"] = (E, F)
        s[p + ff + ".W2.bias"] = (E,)
        s[p + n2 + ".norm.weight"] = (E,)
        s[p + n2 + ".norm.bias"] = (E,)
    if problem == "cvrp":
        s["decoder.Wq_last.weight"] = (H * dk, E + 1)
    else:
        s["decoder.Wq_first.weight"] = (H * dk, E)
        s["decoder.Wq_last.weight"] = (H * dk, E)
    s["decoder.Wk.weight"] = (H * dk, E)
    s["decoder.Wv.weight"] = (H * dk, E)
    s["decoder.multi_head_combine.weight"] = (E, H * dk)
    s["decoder.multi_head_combine.bias"] = (E,)
    if local:
        # one local policy per ensemble member for CVRP (reference models.py:296-298); member 0 first, so that the
        # weight stream of the ensemble_size = 1 fixtures is unchanged
        members = int(mp.get("ensemble_size", 1)) if problem == "cvrp" else 1
        for i in range(members):
            lp = f"decoder.local_policies.{i}." if problem == "cvrp" else "decoder.local_policy_0."
            nfeat = 3 if problem == "cvrp" else 2
            s[lp + "cur_token_emb"] = (le,)
            s[lp + "init_emb.weight"] = (le, nfeat)
            s[lp + "init_emb.bias"] = (le,)
            s[lp + "Wq.weight"] = (lh * ldk, le)
            s[lp + "Wk.weight"] = (lh * ldk, le)
            s[lp + "Wv.weight"] = (lh * ldk, le)
            s[lp + "multi_head_combine.weight"] = (le, lh * ldk)
            s[lp + "multi_head_combine.bias"] = (le,)
    return s


def golden_weights(problem: str, seed: int, mp: dict, local: bool = True, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """fp32 weights, torch-like scale: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norm scale U(0.5,1.5),
    norm shift U(-0.1,0.1); cur_token_emb U(-1,1).  ``gain`` multiplies decoder + local matrices
    (sharper policies, exercises tanh saturation)."""
    rs = np.random.RandomState(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shp in model_param_shapes(problem, mp, local).items():
        if name.endswith("norm.weight"):
            w = rs.uniform(0.5, 1.5, size=shp)
        elif name.endswith("norm.bias"):
            w = rs.uniform(-0.1, 0.1, size=shp)
        elif name.endswith("cur_token_emb"):
            w = rs.uniform(-1.0, 1.0, size=shp)
        else:
            fan_in = shp[1] if len(shp) == 2 else shp[0]
            b = 1.0 / np.sqrt(fan_in)
            w = rs.uniform(-b, b, size=shp)
            if name.startswith("decoder.") and len(shp) == 2:
                w = w * gain
        out[name] = w.astype(np.float32)
    return out


def golden_cvrp_problem(seed: int, B: int, N: int, capacity: float):
    """depot (B,1,2), loc (B,N,2), demand (B,N) fp32 -- same shape/semantics as
    generate_vrp_data's uniform branch (generate_data.py:10-14,84-89)."""
    rs = np.random.RandomState(seed)
    depot = rs.uniform(size=(B, 1, 2)).astype(np.float32)
    loc = rs.uniform(size=(B, N, 2)).astype(np.float32)
    demand = (rs.randint(1, 10, size=(B, N)).astype(np.float32) / np.float32(capacity)).astype(np.float32)
    return depot, loc, demand


def golden_tsp_problem(seed: int, B: int, N: int):
    rs = np.random.RandomState(seed)
    return rs.uniform(size=(B, N, 2)).astype(np.float32)


def load_golden(name: str):
    path = os.path.join(GOLDEN_DIR, name)
    return np.load(path, allow_pickle=False)
