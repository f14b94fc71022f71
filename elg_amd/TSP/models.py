"""Parameter containers + encoder of the TSP policy with the reference's module / state_dict names
(gaocrr/ELG TSP/models.py).  Decoder and local policy run inside the HIP kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from elg_amd import encoder as enc_host
from elg_amd import engine as eng
from elg_amd import _lib as L
from elg_amd.CVRP.models import AddAndInstanceNormalization, FeedForward, local_policy_att  # same definitions


class EncoderLayer(nn.Module):
    """Parameter container with the reference's names (TSP/models.py:157-172); computed by elg_encoder_fwd / _bwd."""

    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        self.Wq = nn.Linear(e, h * d, bias=False)
        self.Wk = nn.Linear(e, h * d, bias=False)
        self.Wv = nn.Linear(e, h * d, bias=False)
        self.multi_head_combine = nn.Linear(h * d, e)
        self.addAndNormalization1 = AddAndInstanceNormalization(**model_params)
        self.feedForward = FeedForward(**model_params)
        self.addAndNormalization2 = AddAndInstanceNormalization(**model_params)


class TSP_Encoder(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        self.embedding = nn.Linear(2, model_params['embedding_dim'])
        self.layers = nn.ModuleList([EncoderLayer(**model_params) for _ in range(model_params['encoder_layer_num'])])

    def forward(self, data):
        """reference TSP/models.py:145-154, inference only (training goes through TSPModel.pre_forward)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("TSP_Encoder.forward is inference-only; train through TSPModel.pre_forward")
        names = enc_host.parameter_names(L.PROBLEM_TSP, len(self.layers))
        sd = dict(self.named_parameters(prefix="encoder"))
        params = [sd[n] for n in names if n.startswith("encoder.")]
        return enc_host.encode_only(L.PROBLEM_TSP, data, None, params, len(self.layers), self.model_params['ff_hidden_dim'])


class TSP_Decoder(nn.Module):
    def __init__(self, **model_params):
        super().__init__()
        self.model_params = model_params
        e, h, d = model_params['embedding_dim'], model_params['head_num'], model_params['qkv_dim']
        if (e, h, d) != (eng.E, eng.H, eng.DK):
            raise ValueError(f"embedding_dim / head_num / qkv_dim = {(e, h, d)}: libelg_hip.so is built for "
                             f"{(eng.E, eng.H, eng.DK)} only (the reference's config.yml)")
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

    def fold_local(self):
        """Folded local-policy tables (engine.fold_local_tables), or None without the ensemble head."""
        mp = self.model_params
        has_local = bool(mp['ensemble'] and self.local)
        return self.local_policy_0.folded_tables(int(mp['local_size'][0])) if has_local else None

    def set_tables(self, encoded_nodes, tables, loc):
        mp = self.model_params
        has_local = bool(mp['ensemble'] and self.local)
        self.policy = eng.Policy(tables, loc, int(mp['local_size'][0]), float(mp['xi']), float(mp['logit_clipping']), 1.0,
                                 has_local, bool(mp['distance_penalty']), bool(mp.get('euclidean', False)))
        self.k, self.v = tables["K"], tables["V"]
        self.single_head_key = encoded_nodes.transpose(1, 2)

    def set_kv(self, encoded_nodes):
        """reference decoder.set_kv on given encodings (inference only; TSPModel.pre_forward produces the same tables
        together with the encoder and carries the backward)."""
        if torch.is_grad_enabled() and encoded_nodes.requires_grad:
            raise RuntimeError("set_kv is inference-only; train through TSPModel.pre_forward")
        sd = dict(self.named_parameters(prefix="decoder"))
        names = [n for n in enc_host.parameter_names(L.PROBLEM_TSP, 0) if n.startswith("decoder.")]
        tables = enc_host.fold_only(L.PROBLEM_TSP, encoded_nodes, [sd[n] for n in names])
        self.set_tables(encoded_nodes, tables, self.fold_local())
