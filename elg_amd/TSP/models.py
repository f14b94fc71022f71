"""Parameter containers + encoder of the TSP policy with the reference's module / state_dict names
(gaocrr/ELG TSP/models.py).  Decoder and local policy run inside the HIP kernels."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from elg_amd import engine as eng
from elg_amd import _lib as L
from elg_amd.CVRP.models import AddAndInstanceNormalization, FeedForward, Linear, local_policy_att  # same definitions


class EncoderLayer(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        self.Wq = Linear(e, h * d, bias=False)
        self.Wk = Linear(e, h * d, bias=False)
        self.Wv = Linear(e, h * d, bias=False)
        self.multi_head_combine = Linear(h * d, e)
        self.addAndNormalization1 = AddAndInstanceNormalization(**model_params)
        self.feedForward = FeedForward(**model_params)
        self.addAndNormalization2 = AddAndInstanceNormalization(**model_params)

    def forward(self, x):
        B, n, _ = x.shape
        h = self.model_params['head_num']

        def heads(t):
            return t.view(B, n, h, -1).transpose(1, 2)
        q, k, v = eng.qkv_linear(x, self.Wq.weight, self.Wk.weight, self.Wv.weight)      # one GEMM, shared input
        att = eng.self_attention(q, k, v)                          # SDPA forward, MFMA attention backward
        o1 = self.addAndNormalization1(x, self.multi_head_combine(att))
        return self.addAndNormalization2(o1, self.feedForward(o1))


class TSP_Encoder(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.embedding = nn.Linear(2, model_params['embedding_dim'])
        self.layers = nn.ModuleList([EncoderLayer(**model_params) for _ in range(model_params['encoder_layer_num'])])

    def forward(self, data):
        out = self.embedding(data)
        for layer in self.layers:
            out = layer(out)
        return out


class TSP_Decoder(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        if (e, h, d) != (eng.E, eng.H, eng.DK):
            raise NotImplementedError("HIP kernels are built for embedding 128, 8 heads x 16")
        self.Wq_first = nn.Linear(e, h * d, bias=False)
        self.Wq_last = nn.Linear(e, h * d, bias=False)
        self.Wk = nn.Linear(e, h * d, bias=False)
        self.Wv = nn.Linear(e, h * d, bias=False)
        self.multi_head_combine = nn.Linear(h * d, e)
        self.local = False
        self.policy = None

    def add_local_policy(self, device, idx=0):
        mp = dict(self.model_params)
        mp['demand'] = False
        self.local_policy_0 = local_policy_att(mp).to(device)
        self.local = True

    def fold(self, encoded_nodes):
        """(tables, loc): decoder / local-policy weights folded for the HIP kernels (engine.fold_*)."""
        mp = self.model_params
        dec = {"Wq_first.weight": self.Wq_first.weight, "Wq_last.weight": self.Wq_last.weight,
               "Wk.weight": self.Wk.weight, "Wv.weight": self.Wv.weight,
               "multi_head_combine.weight": self.multi_head_combine.weight,
               "multi_head_combine.bias": self.multi_head_combine.bias}
        tables = eng.fold_decoder_tables(dec, encoded_nodes, L.PROBLEM_TSP)
        has_local = bool(mp['ensemble'] and self.local)
        loc = self.local_policy_0.folded_tables(int(mp['local_size'][0])) if has_local else None
        return tables, loc

    def set_tables(self, encoded_nodes, tables, loc):
        mp = self.model_params
        has_local = bool(mp['ensemble'] and self.local)
        self.policy = eng.Policy(tables, loc, int(mp['local_size'][0]), float(mp['xi']), float(mp['logit_clipping']), 1.0,
                                 has_local, bool(mp['distance_penalty']))
        self.k, self.v = tables["K"], tables["V"]
        self.single_head_key = encoded_nodes.transpose(1, 2)

    def set_kv(self, encoded_nodes):
        """reference TSP/models.py:231-241 (+ set_q1 :236-241 folded into the Q2 table)."""
        tables, loc = self.fold(encoded_nodes)
        self.set_tables(encoded_nodes, tables, loc)
