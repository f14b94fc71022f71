"""Host side of the native encoder: `CVRPModel.pre_forward` / `TSPModel.pre_forward` (reference CVRP/CVRPModel.py:21-34,
TSP/TSPModel.py:17-24) = encoder + `decoder.set_kv` as ONE call into libelg_hip.so each way (elg_encoder_fwd /
elg_encoder_bwd, csrc/elg_enc.hip).  The parameters stay where the reference's state_dict puts them; this module only
collects their device pointers, owns the activation workspace and hands the gradients back to autograd."""
from __future__ import annotations

import os
import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L

E = 128

# order of the flat parameter list handed to _EncodeFold (None entries are skipped for TSP / CVRP)
_CVRP_LAYER = ("Wq.weight", "Wk.weight", "Wv.weight", "multi_head_combine.weight", "multi_head_combine.bias",
               "add_n_normalization_1.norm.weight", "add_n_normalization_1.norm.bias",
               "feed_forward.W1.weight", "feed_forward.W1.bias", "feed_forward.W2.weight", "feed_forward.W2.bias",
               "add_n_normalization_2.norm.weight", "add_n_normalization_2.norm.bias")
_TSP_LAYER = ("Wq.weight", "Wk.weight", "Wv.weight", "multi_head_combine.weight", "multi_head_combine.bias",
              "addAndNormalization1.norm.weight", "addAndNormalization1.norm.bias",
              "feedForward.W1.weight", "feedForward.W1.bias", "feedForward.W2.weight", "feedForward.W2.bias",
              "addAndNormalization2.norm.weight", "addAndNormalization2.norm.bias")


def parameter_names(kind: int, n_layers: int) -> List[str]:
    """state_dict names in the order of elg_enc_weights."""
    tsp = kind == L.PROBLEM_TSP
    names = (["encoder.embedding.weight", "encoder.embedding.bias"] if tsp else
             ["encoder.embedding_depot.weight", "encoder.embedding_depot.bias",
              "encoder.embedding_node.weight", "encoder.embedding_node.bias"])
    for i in range(n_layers):
        names += [f"encoder.layers.{i}.{n}" for n in (_TSP_LAYER if tsp else _CVRP_LAYER)]
    if tsp:
        names.append("decoder.Wq_first.weight")
    names += ["decoder.Wq_last.weight", "decoder.Wk.weight", "decoder.Wv.weight",
              "decoder.multi_head_combine.weight", "decoder.multi_head_combine.bias"]
    return names


def _fill_weights(w: L.EncWeights, kind: int, n_layers: int, ptrs: Sequence[int]):
    it = iter(ptrs)
    if kind != L.PROBLEM_TSP:
        w.emb_depot_w, w.emb_depot_b = next(it), next(it)
    w.emb_w, w.emb_b = next(it), next(it)
    for i in range(n_layers):
        for f in L._LAYER_FIELDS:
            setattr(w.layer[i], f, next(it))
    if kind == L.PROBLEM_TSP:
        w.dec_Wq_first = next(it)
    w.dec_Wq_last, w.dec_Wk, w.dec_Wv, w.dec_Wc, w.dec_bc = next(it), next(it), next(it), next(it), next(it)
    assert next(it, None) is None


class _Workspace:
    """Activation workspace of one (shape, mode); `gen` detects a second training forward before the backward."""
    _cache: Dict[tuple, "_Workspace"] = {}

    def __init__(self, n, dev):
        self.buf = torch.empty(n, device=dev, dtype=torch.float32)
        self.gen = 0

    @classmethod
    def get(cls, key, n, dev):
        ws = cls._cache.get(key)
        if ws is None or ws.buf.numel() < n:
            ws = cls._cache[key] = _Workspace(n, dev)
        return ws


def _precision(precision) -> int:
    """0 = f32 (parity mode), 1 = the bf16 throughput mode (elg_encoder_args.precision); None: the engine's mode
    (engine.FWD_PRECISION / ELG_FWD_MODE, the switch the rollout obeys)."""
    if precision is None:
        from . import engine as _eng          # (lazy: engine does not import this module)
        precision = _eng.FWD_PRECISION
    return int(precision)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class _EncodeFold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, n_layers, ff, eps, train, precision, xy, demand, *params):
        dev = xy.device
        B, N1, _ = xy.shape
        tsp = kind == L.PROBLEM_TSP
        lib = L.lib()
        n_ws = int(lib.elg_encoder_ws_floats(B, N1, n_layers, ff, int(train)))
        ws = _Workspace.get((B, N1, n_layers, ff, train, str(dev)), n_ws, dev)
        a = L.EncoderArgs()
        a.problem, a.B, a.N1, a.n_layers, a.ff_hidden, a.save, a.eps = kind, B, N1, n_layers, ff, int(train), eps
        a.precision = precision
        if precision == 1:
            # the bf16 mode exists in the fused per-instance kernels only (csrc/elg_enc_fused.hip::enc_fused_ok): say so where the
            # per-GEMM f32 path runs instead, like the rollout does (engine._note_f32_fallback, once per case)
            fused = (4 <= N1 <= 128 and 128 <= ff <= 1024 and ff % 128 == 0 and os.environ.get("ELG_ENC_FUSED", "1")[:1] != "0")
            if not fused:
                from . import engine as _eng
                _eng._note_f32_fallback(f"encoder at N + 1 = {N1}, ff_hidden_dim = {ff}"
                                        + (" with ELG_ENC_FUSED=0" if os.environ.get("ELG_ENC_FUSED", "1")[:1] == "0" else "")
                                        + " (the per-GEMM encoder path has no bf16 instantiation)")
        a.xy, a.demand = _ptr(xy), _ptr(demand)
        _fill_weights(a.W, kind, n_layers, [p.data_ptr() for p in params])
        enc, K, V, PK, Q1 = (torch.empty(B, N1, E, device=dev) for _ in range(5))
        Q2 = torch.empty(B, N1, E, device=dev) if tsp else None
        pb = torch.empty(B, N1, device=dev)
        wl = None if tsp else torch.empty(E, device=dev)
        a.enc, a.K, a.V, a.PK, a.Q1, a.Q2 = _ptr(enc), _ptr(K), _ptr(V), _ptr(PK), _ptr(Q1), _ptr(Q2)
        a.pb, a.wl = _ptr(pb), _ptr(wl)
        a.ws, a.ws_floats = _ptr(ws.buf), ws.buf.numel()
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            L.check(lib.elg_encoder_fwd(C.byref(a), stream), "elg_encoder_fwd")
        if train:
            ws.gen += 1
            ctx.args, ctx.ws, ctx.gen, ctx.tsp = a, ws, ws.gen, tsp
            # Buffers of `a` that the backward reads (elg_encoder_bwd: xy, demand, enc).  The INPUTS may sit in ctx; the
            # output `enc` must go through save_for_backward: an output stored on ctx is a reference cycle (tensor ->
            # grad_fn -> ctx -> tensor) that only Python's cyclic collector frees -- 16.5 MB of tables per training step
            # piled up until a long run died with an out-of-memory error.
            ctx.keep = (xy, demand)
            ctx.save_for_backward(*params, enc)
            # an output nobody differentiated (the training step never uses `enc` itself) arrives in backward as None, not as a
            # zero tensor autograd would have to fill first: elg_encoder_bwd takes NULL for every cotangent that is absent
            ctx.set_materialize_grads(False)
        return enc, K, V, PK, pb, Q1, Q2, wl

    @staticmethod
    def backward(ctx, g_enc, gK, gV, gPK, gpb, gQ1, gQ2, gwl):
        params = ctx.saved_tensors[:-1]                                 # (+ enc, kept alive for the kernel)
        ws = ctx.ws
        if ws.gen != ctx.gen:
            raise RuntimeError("elg_amd.encoder: the activation workspace was overwritten by a later training forward of "
                               "the same shape before this backward ran")
        dev = params[0].device
        a = ctx.args
        lib = L.lib()
        ba = L.EncoderBwdArgs()
        ba.fwd = a

        def cg(t):
            return None if t is None else t.contiguous().float()
        g_enc, gK, gV, gPK, gpb, gQ1, gQ2, gwl = (cg(t) for t in (g_enc, gK, gV, gPK, gpb, gQ1, gQ2, gwl))
        if gpb is not None and gPK is None:
            gPK = torch.zeros(a.B, a.N1, E, device=dev)
        ba.g_enc, ba.gK, ba.gV, ba.gPK, ba.gpb = _ptr(g_enc), _ptr(gK), _ptr(gV), _ptr(gPK), _ptr(gpb)
        ba.gQ1, ba.gQ2, ba.gwl = _ptr(gQ1), _ptr(gQ2), _ptr(gwl)
        sizes = [p.numel() for p in params]
        flat = torch.zeros(sum(sizes), device=dev)
        grads, ptrs, off = [], [], 0
        for p, n in zip(params, sizes):
            g = flat[off:off + n].view_as(p)
            grads.append(g)
            ptrs.append(g.data_ptr())
            off += n
        _fill_weights(ba.G, a.problem, a.n_layers, ptrs)
        n2 = int(lib.elg_encoder_bwd_ws_floats(a.B, a.N1, a.n_layers, a.ff_hidden))
        ws2 = _Workspace.get(("bwd", a.B, a.N1, a.n_layers, a.ff_hidden, str(dev)), n2, dev)
        ba.ws2, ba.ws2_floats = _ptr(ws2.buf), ws2.buf.numel()
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            L.check(lib.elg_encoder_bwd(C.byref(ba), stream), "elg_encoder_bwd")
        return (None, None, None, None, None, None, None, None, *grads)


def encode_and_fold(kind: int, xy: torch.Tensor, demand: Optional[torch.Tensor], params: Sequence[torch.Tensor],
                    n_layers: int, ff_hidden: int, eps: float = 1e-5, precision=None):
    """-> (encoded_nodes, tables) with tables = {K, V, PK, pb, Q1, Q2, wl} (engine.Policy layout).  GPU only."""
    if not xy.is_cuda:
        raise RuntimeError("elg_amd: the encoder runs on the GPU only -- the HIP path has no CPU fallback")
    for p in params:
        if p.dtype != torch.float32 or not p.is_contiguous() or p.device != xy.device:
            raise ValueError("encoder parameters must be contiguous fp32 tensors on the problem's device")
    xy = xy.contiguous().float()
    demand = None if demand is None else demand.contiguous().float()
    train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    enc, K, V, PK, pb, Q1, Q2, wl = _EncodeFold.apply(kind, n_layers, ff_hidden, eps, train, _precision(precision), xy, demand, *params)
    return enc, dict(K=K, V=V, PK=PK, pb=pb, Q1=Q1, Q2=Q2, wl=wl)


def encode_only(kind: int, xy: torch.Tensor, demand: Optional[torch.Tensor], enc_params: Sequence[torch.Tensor],
                n_layers: int, ff_hidden: int, eps: float = 1e-5, precision=None) -> torch.Tensor:
    """Encoder without the decoder tables (inference): `enc_params` = the encoder.* entries of parameter_names()."""
    if not xy.is_cuda:
        raise RuntimeError("elg_amd: the encoder runs on the GPU only -- the HIP path has no CPU fallback")
    dev = xy.device
    xy = xy.contiguous().float()
    demand = None if demand is None else demand.contiguous().float()
    B, N1, _ = xy.shape
    lib = L.lib()
    ws = _Workspace.get((B, N1, n_layers, ff_hidden, False, str(dev)), int(lib.elg_encoder_ws_floats(B, N1, n_layers, ff_hidden, 0)), dev)
    a = L.EncoderArgs()
    a.problem, a.B, a.N1, a.n_layers, a.ff_hidden, a.save, a.eps = kind, B, N1, n_layers, ff_hidden, 0, eps
    a.precision = _precision(precision)
    a.xy, a.demand = _ptr(xy), _ptr(demand)
    ptrs = [p.data_ptr() for p in enc_params] + [0] * (6 if kind == L.PROBLEM_TSP else 5)
    _fill_weights(a.W, kind, n_layers, ptrs)
    enc = torch.empty(B, N1, E, device=dev)
    a.enc, a.ws, a.ws_floats = _ptr(enc), _ptr(ws.buf), ws.buf.numel()
    with torch.cuda.device(dev):
        L.check(lib.elg_encoder_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "elg_encoder_fwd")
    return enc


def fold_only(kind: int, enc: torch.Tensor, dec_params: Sequence[torch.Tensor], precision=None) -> Dict[str, torch.Tensor]:
    """`decoder.set_kv(encoded_nodes)` on given encodings (reference models.py:300-308, TSP/models.py:231-241), inference
    only: the table part of elg_encoder_fwd (n_layers = 0).  dec_params = the decoder.* entries of parameter_names()."""
    if not enc.is_cuda:
        raise RuntimeError("elg_amd: set_kv runs on the GPU only -- the HIP path has no CPU fallback")
    dev = enc.device
    enc = enc.detach().contiguous().float()
    B, N1, _ = enc.shape
    tsp = kind == L.PROBLEM_TSP
    a = L.EncoderArgs()
    a.problem, a.B, a.N1, a.n_layers, a.ff_hidden, a.save, a.eps = kind, B, N1, 0, 512, 0, 1e-5
    a.precision = _precision(precision)
    ptrs = [0] * (2 if tsp else 4) + [p.data_ptr() for p in dec_params]
    _fill_weights(a.W, kind, 0, ptrs)
    K, V, PK, Q1 = (torch.empty(B, N1, E, device=dev) for _ in range(4))
    Q2 = torch.empty(B, N1, E, device=dev) if tsp else None
    pb = torch.empty(B, N1, device=dev)
    wl = None if tsp else torch.empty(E, device=dev)
    a.enc, a.K, a.V, a.PK, a.Q1, a.Q2, a.pb, a.wl = _ptr(enc), _ptr(K), _ptr(V), _ptr(PK), _ptr(Q1), _ptr(Q2), _ptr(pb), _ptr(wl)
    with torch.cuda.device(dev):
        L.check(L.lib().elg_encoder_fwd(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "elg_encoder_fwd")
    return dict(K=K, V=V, PK=PK, pb=pb, Q1=Q1, Q2=Q2, wl=wl)


def _teardown():
    # (no device synchronisation on an exit path: see engine._teardown)
    _Workspace._cache.clear()


import atexit  # noqa: E402

atexit.register(_teardown)
