"""Parameter containers + encoder of the CVRP policy, with the reference's module / state_dict
names (gaocrr/ELG CVRP/models.py; key layout in SURVEY.md A.5) so checkpoints are interchangeable.

Nothing here computes in PyTorch: the modules own the parameters (same names, shapes and
seeded-init order as the reference), the encoder + `set_kv` run in csrc/elg_enc.hip
(elg_encoder_fwd / elg_encoder_bwd via elg_amd/encoder.py) and the decoder / local policy inside the
rollout kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from elg_amd import engine as eng
from elg_amd import encoder as enc_host
from elg_amd import _lib as L


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
            raise ValueError(f"local_att_hidden_dim / local_att_head_num / local_att_qkv_dim = {(self.emb_dim, self.head_num, self.qkv_dim)}: "
                             f"libelg_hip.so is built for {(eng.LE, eng.LH, eng.LDK)} only (the reference's config.yml:47-49)")
        if not 1 <= int(self.local_size) <= L.MAX_LOCAL_SIZE:
            raise ValueError(f"local_size {self.local_size}: the kernels hold the k nearest neighbours (+ the depot) in "
                             f"{L.MAX_LOCAL_SIZE + 1} slots, one per lane -- supported 1 .. {L.MAX_LOCAL_SIZE} (the reference's default: 40 / 30; "
                             f"above {L.ROWS_LOCAL_SIZE} the one-wavefront kernels and the replay backward run)")

    def folded_tables(self, n_slots: int) -> torch.Tensor:
        lp = {k: v for k, v in self.named_parameters()}
        return eng.fold_local_tables(lp, self.init_emb.in_features, n_slots, bool(self.model_params.get('positional', True)))


class AddAndInstanceNormalization(nn.Module):
    """Parameters of `InstanceNorm1d(embedding_dim, affine=True)` over the node axis (reference models.py:506-527); the
    normalisation itself is the epilogue of the combine / feed-forward GEMMs in csrc/elg_enc.hip."""

    def __init__(self, **model_params):
        super().__init__()
        self.norm = nn.InstanceNorm1d(model_params['embedding_dim'], affine=True, track_running_stats=False)


class FeedForward(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.W1 = nn.Linear(model_params['embedding_dim'], model_params['ff_hidden_dim'])
        self.W2 = nn.Linear(model_params['ff_hidden_dim'], model_params['embedding_dim'])


class EncoderLayer(nn.Module):
    """Parameter container with the reference's names (models.py:232-247); computed by elg_encoder_fwd / _bwd."""

    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        self.Wq = nn.Linear(e, h * d, bias=False)
        self.Wk = nn.Linear(e, h * d, bias=False)
        self.Wv = nn.Linear(e, h * d, bias=False)
        self.multi_head_combine = nn.Linear(h * d, e)
        self.add_n_normalization_1 = AddAndInstanceNormalization(**model_params)
        self.feed_forward = FeedForward(**model_params)
        self.add_n_normalization_2 = AddAndInstanceNormalization(**model_params)


class CVRP_Encoder(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e = model_params['embedding_dim']
        self.embedding_depot = nn.Linear(2, e)
        self.embedding_node = nn.Linear(3, e)
        self.layers = nn.ModuleList([EncoderLayer(**model_params) for _ in range(model_params['encoder_layer_num'])])

    def forward(self, depot_xy, node_xy_demand, dist=None):
        """reference models.py:211-229, inference only (training goes through CVRPModel.pre_forward, which also
        produces the decoder tables and carries the backward)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("CVRP_Encoder.forward is inference-only; train through CVRPModel.pre_forward")
        xy = torch.cat((depot_xy, node_xy_demand[:, :, :2]), dim=1)
        dem = torch.cat((torch.zeros_like(depot_xy[:, :, 0]), node_xy_demand[:, :, 2]), dim=1)
        names = enc_host.parameter_names(L.PROBLEM_CVRP, len(self.layers))
        sd = dict(self.named_parameters(prefix="encoder"))
        params = [sd[n] for n in names if n.startswith("encoder.")]
        return enc_host.encode_only(L.PROBLEM_CVRP, xy, dem, params, len(self.layers), self.model_params['ff_hidden_dim'])


class CVRP_Decoder(nn.Module):
    """Owns Wq_last / Wk / Wv / multi_head_combine (+ local_policies) and folds them per batch."""

    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        if (e, h, d) != (eng.E, eng.H, eng.DK):
            raise ValueError(f"embedding_dim / head_num / qkv_dim = {(e, h, d)}: libelg_hip.so is built for "
                             f"{(eng.E, eng.H, eng.DK)} only (the reference's config.yml:41-44)")
        self.Wq_last = nn.Linear(e + 1, h * d, bias=False)
        self.Wk = nn.Linear(e, h * d, bias=False)
        self.Wv = nn.Linear(e, h * d, bias=False)
        self.multi_head_combine = nn.Linear(h * d, e)
        self.local = False
        self.policy = None           # engine.Policy of the current batch

    def add_local_policy(self, device):
        n = int(self.model_params['ensemble_size'])
        if not 1 <= n <= L.MAX_ENS:
            raise ValueError(f"ensemble_size {n}: supported 1 .. {L.MAX_ENS} local policies (the reference's default: 1)")
        if len(self.model_params['local_size']) < n:
            raise IndexError("model_params['local_size'] needs one entry per ensemble member (reference models.py:14)")
        self.local_policies = nn.ModuleList([local_policy_att(self.model_params, idx=i).to(device) for i in range(n)])
        self.local = True

    def fold_local(self):
        """Folded local-policy tables (engine.fold_local_tables), or None without the ensemble head."""
        mp = self.model_params
        has_local = bool(mp['ensemble'] and self.local)
        if not has_local:
            return None
        locs = [lp.folded_tables(int(lp.local_size) + 1) for lp in self.local_policies]     # reference models.py:296-298
        return locs[0] if len(locs) == 1 else torch.cat(locs)

    def set_tables(self, encoded_nodes, tables, loc):
        mp = self.model_params
        has_local = bool(mp['ensemble'] and self.local)
        self.policy = eng.Policy(tables, loc, int(mp['local_size'][0]), float(mp['xi']), float(mp['logit_clipping']),
                                 1.0 / float(mp['ensemble_size']), has_local, bool(mp['distance_penalty']),
                                 bool(mp.get('euclidean', False)),
                                 tuple(int(lp.local_size) for lp in self.local_policies) if (has_local and len(self.local_policies) > 1) else ())
        # attributes the reference exposes after set_kv
        self.k, self.v = tables["K"], tables["V"]
        self.single_head_key = encoded_nodes.transpose(1, 2)

    def set_kv(self, encoded_nodes):
        """reference decoder.set_kv on given encodings (inference only; CVRPModel.pre_forward produces the same tables
        together with the encoder and carries the backward)."""
        if torch.is_grad_enabled() and encoded_nodes.requires_grad:
            raise RuntimeError("set_kv is inference-only; train through CVRPModel.pre_forward")
        sd = dict(self.named_parameters(prefix="decoder"))
        names = [n for n in enc_host.parameter_names(L.PROBLEM_CVRP, 0) if n.startswith("decoder.")]
        tables = enc_host.fold_only(L.PROBLEM_CVRP, encoded_nodes, [sd[n] for n in names])
        self.set_tables(encoded_nodes, tables, self.fold_local())
