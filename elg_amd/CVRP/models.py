"""Parameter containers + encoder of the CVRP policy, with the reference's module / state_dict
names (gaocrr/ELG CVRP/models.py; key layout in SURVEY.md A.5) so checkpoints are interchangeable.

Only the encoder runs as PyTorch ops here (dense batched GEMMs, once per batch); the decoder and
the local policy never execute in Python: `CVRP_Decoder.set_kv` folds their weights into the tables
the HIP rollout kernels consume (elg_amd/engine.py)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from elg_amd import engine as eng
from elg_amd import _lib as L


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose weight gradient uses the split-K MFMA GEMM."""

    def forward(self, x):
        return eng.linear(x, self.weight, self.bias)


class local_policy_att(nn.Module):
    """Weights of the k-NN local attention policy (reference models.py:7-36).  Forward lives in
    csrc/elg_rollout.h::local_policy; this module only owns the parameters."""

    def __init__(self, model_params, idx=0):
        super().__init__()
        self.model_params = model_params
        self.emb_dim = model_params['local_att_hidden_dim']
        self.head_num = model_params['local_att_head_num']
        self.qkv_dim = model_params['local_att_qkv_dim']
        self.local_size = model_params['local_size'][idx]
        n_feat = 3 if model_params.get('demand', False) else 2
        self.init_emb = nn.Linear(n_feat, self.emb_dim)
        self.cur_token_emb = nn.Parameter(torch.empty(self.emb_dim).uniform_(-1, 1))
        hd = self.head_num * self.qkv_dim
        self.Wq = nn.Linear(self.emb_dim, hd, bias=False)
        self.Wk = nn.Linear(self.emb_dim, hd, bias=False)
        self.Wv = nn.Linear(self.emb_dim, hd, bias=False)
        self.multi_head_combine = nn.Linear(hd, self.emb_dim)
        if (self.emb_dim, self.head_num, self.qkv_dim) != (eng.LE, eng.LH, eng.LDK):
            raise NotImplementedError("HIP kernels are built for local_att 32/4/8")
        if model_params.get('euclidean', False):
            raise NotImplementedError("euclidean local features are not built (SURVEY 8f rank 4)")

    def folded_tables(self, n_slots: int) -> torch.Tensor:
        lp = {k: v for k, v in self.named_parameters()}
        return eng.fold_local_tables(lp, self.init_emb.in_features, n_slots)


class AddAndInstanceNormalization(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.norm = nn.InstanceNorm1d(model_params['embedding_dim'], affine=True, track_running_stats=False)

    def forward(self, a, b):
        # per (instance, channel) statistics over the node axis (reference models.py:506-527); one fused HIP kernel
        # forward and one backward (csrc/elg_encoder.hip) instead of add + transposes + MIOpen batch-norm
        return eng.add_instance_norm(a, b, self.norm.weight, self.norm.bias, self.norm.eps)


class FeedForward(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.W1 = Linear(model_params['embedding_dim'], model_params['ff_hidden_dim'])
        self.W2 = Linear(model_params['ff_hidden_dim'], model_params['embedding_dim'])

    def forward(self, x):
        return self.W2(F.relu(self.W1(x)))


class EncoderLayer(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        self.Wq = Linear(e, h * d, bias=False)
        self.Wk = Linear(e, h * d, bias=False)
        self.Wv = Linear(e, h * d, bias=False)
        self.multi_head_combine = Linear(h * d, e)
        self.add_n_normalization_1 = AddAndInstanceNormalization(**model_params)
        self.feed_forward = FeedForward(**model_params)
        self.add_n_normalization_2 = AddAndInstanceNormalization(**model_params)

    def forward(self, x):
        B, n, _ = x.shape
        h = self.model_params['head_num']

        def heads(t):
            return t.view(B, n, h, -1).transpose(1, 2)
        q, k, v = eng.qkv_linear(x, self.Wq.weight, self.Wk.weight, self.Wv.weight)      # one GEMM, shared input
        att = eng.self_attention(q, k, v)                          # SDPA forward, MFMA attention backward
        o1 = self.add_n_normalization_1(x, self.multi_head_combine(att))
        return self.add_n_normalization_2(o1, self.feed_forward(o1))


class CVRP_Encoder(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e = model_params['embedding_dim']
        self.embedding_depot = nn.Linear(2, e)
        self.embedding_node = nn.Linear(3, e)
        self.layers = nn.ModuleList([EncoderLayer(**model_params) for _ in range(model_params['encoder_layer_num'])])

    def forward(self, depot_xy, node_xy_demand, dist=None):
        out = torch.cat((self.embedding_depot(depot_xy), self.embedding_node(node_xy_demand)), dim=1)
        for layer in self.layers:
            out = layer(out)
        return out


class CVRP_Decoder(nn.Module):
    """Owns Wq_last / Wk / Wv / multi_head_combine (+ local_policies) and folds them per batch."""

    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        if (e, h, d) != (eng.E, eng.H, eng.DK):
            raise NotImplementedError("HIP kernels are built for embedding 128, 8 heads x 16")
        self.Wq_last = nn.Linear(e + 1, h * d, bias=False)
        self.Wk = nn.Linear(e, h * d, bias=False)
        self.Wv = nn.Linear(e, h * d, bias=False)
        self.multi_head_combine = nn.Linear(h * d, e)
        self.local = False
        self.policy = None           # engine.Policy of the current batch

    def add_local_policy(self, device):
        n = self.model_params['ensemble_size']
        if n != 1:
            raise NotImplementedError("ensemble_size > 1 is not built (SURVEY 8f rank 4)")
        self.local_policies = nn.ModuleList([local_policy_att(self.model_params, idx=i).to(device) for i in range(n)])
        self.local = True

    def fold(self, encoded_nodes):
        """(tables, loc): decoder / local-policy weights folded for the HIP kernels (engine.fold_*)."""
        mp = self.model_params
        dec = {"Wq_last.weight": self.Wq_last.weight, "Wk.weight": self.Wk.weight, "Wv.weight": self.Wv.weight,
               "multi_head_combine.weight": self.multi_head_combine.weight,
               "multi_head_combine.bias": self.multi_head_combine.bias}
        tables = eng.fold_decoder_tables(dec, encoded_nodes, L.PROBLEM_CVRP)
        has_local = bool(mp['ensemble'] and self.local)
        loc = self.local_policies[0].folded_tables(int(mp['local_size'][0]) + 1) if has_local else None
        return tables, loc

    def set_tables(self, encoded_nodes, tables, loc):
        mp = self.model_params
        has_local = bool(mp['ensemble'] and self.local)
        self.policy = eng.Policy(tables, loc, int(mp['local_size'][0]), float(mp['xi']), float(mp['logit_clipping']),
                                 1.0 / float(mp['ensemble_size']), has_local, bool(mp['distance_penalty']))
        # attributes the reference exposes after set_kv
        self.k, self.v = tables["K"], tables["V"]
        self.single_head_key = encoded_nodes.transpose(1, 2)

    def set_kv(self, encoded_nodes):
        """reference models.py:300-308, plus the folds described in engine.fold_decoder_tables."""
        tables, loc = self.fold(encoded_nodes)
        self.set_tables(encoded_nodes, tables, loc)
