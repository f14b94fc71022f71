"""ctypes binding of libelg_hip.so (include/elg_hip.h).  No CPU fallback: if the library is
missing the import fails loudly -- build it with `python -m elg_amd.build`."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ELG_HIP_LIB: another build of the same ABI (A/B timing of kernel versions, tools/time_coop_variants.py); default: the in-tree library
LIB_PATH = os.environ.get("ELG_HIP_LIB") or os.path.join(HERE, "libelg_hip.so")

ELG_OK, ELG_EINVAL, ELG_ELAUNCH, ELG_ENOTIMPL = 0, -1, -2, -3
PROBLEM_CVRP, PROBLEM_TSP = 0, 1
MODE_GREEDY, MODE_SAMPLE, MODE_FORCED = 0, 1, 2
# elg_rollout_last_kernel(): which construction kernel the last elg_rollout_fwd of this thread launched (include/elg_hip.h)
KERNEL_WAVE, KERNEL_COOP, KERNEL_COOP_SPLIT, KERNEL_STREAM, KERNEL_XL, KERNEL_XM, KERNEL_COOP_WIDE = 1, 2, 3, 4, 5, 6, 7

LOC_ROWS = 64
LOC_LA, LOC_LT, LOC_LAV, LOC_LCV, LOC_LWC, LOC_LBC, LOC_LWE, LOC_LPE, LOC_SIZE = \
    0, 16, 272, 368, 2416, 3440, 3472, 3568, 5616

MAX_ENS = 4
MAX_LOCAL_SIZE = 63             # ELG_SLOT_MAX - 1 (csrc/elg_rollout.h): k nearest neighbours + the depot slot, one slot per lane
ROWS_LOCAL_SIZE = 47            # ELG_SLOT_STRIDE - 1: what the matrix-core kernels and the saved training rows are built for
_vp = C.c_void_p


class RolloutArgs(C.Structure):
    _fields_ = [
        ("problem", C.c_int32), ("B", C.c_int32), ("M", C.c_int32), ("N1", C.c_int32), ("K", C.c_int32),
        ("Tmax", C.c_int32), ("mode", C.c_int32), ("Tforced", C.c_int32), ("has_local", C.c_int32),
        ("has_penalty", C.c_int32), ("max_steps", C.c_int32), ("do_decode", C.c_int32), ("do_update", C.c_int32),
        ("use_state", C.c_int32), ("waves", C.c_int32), ("tiles", C.c_int32), ("lds_stage", C.c_int32),
        ("dump_T", C.c_int32),
        ("xi", C.c_float), ("clip", C.c_float), ("inv_ens", C.c_float), ("variant", C.c_int32),
        ("dump_logits", C.c_int32), ("euclidean", C.c_int32),
        ("ens", C.c_int32), ("Kens", C.c_int32 * MAX_ENS), ("precision", C.c_int32),
        ("seed", C.c_uint64),
        ("Kmat", _vp), ("Vmat", _vp), ("PK", _vp), ("pb", _vp), ("Q1", _vp), ("Q2", _vp), ("wl", _vp),
        ("xy", _vp), ("demand", _vp), ("nbr_idx", _vp), ("nbr_dist", _vp), ("nbr_theta", _vp), ("loc", _vp),
        ("starts", _vp), ("forced", _vp), ("uniforms", _vp),
        ("st_cur", _vp), ("st_cnt", _vp), ("st_fin", _vp), ("st_first", _vp), ("st_load", _vp), ("st_len", _vp),
        ("st_vis", _vp),
        ("actions", _vp), ("probs", _vp), ("reward", _vp), ("tlen", _vp), ("full_probs", _vp),
        ("trA", _vp), ("trPC", _vp), ("trCsel", _vp), ("trQ", _vp), ("trO", _vp), ("trLoad", _vp), ("trSlot", _vp), ("trF", _vp), ("trMask", _vp), ("scratch", _vp), ("trLse", _vp),
    ]


class BwdArgs(C.Structure):
    _fields_ = [
        ("fwd", RolloutArgs), ("T", C.c_int32), ("member", C.c_int32),
        ("gprob", _vp), ("rowA", _vp), ("rowDL", _vp), ("rowQ", _vp), ("rowO", _vp), ("rowLoad", _vp),
        ("rowDU", _vp), ("gloc", _vp), ("time_major", C.c_int32), ("local_only", C.c_int32),
        ("row_stride", C.c_int64),
    ]


ENC_MAX_LAYERS = 8
_LAYER_FIELDS = ("Wq", "Wk", "Wv", "Wc", "bc", "g1", "b1", "W1", "bf1", "W2", "bf2", "g2", "b2")


class EncLayer(C.Structure):
    _fields_ = [(n, _vp) for n in _LAYER_FIELDS]


class EncWeights(C.Structure):
    _fields_ = [("emb_depot_w", _vp), ("emb_depot_b", _vp), ("emb_w", _vp), ("emb_b", _vp),
                ("layer", EncLayer * ENC_MAX_LAYERS),
                ("dec_Wq_first", _vp), ("dec_Wq_last", _vp), ("dec_Wk", _vp), ("dec_Wv", _vp), ("dec_Wc", _vp),
                ("dec_bc", _vp)]


class DecoderBwdArgs(C.Structure):
    _fields_ = [("problem", C.c_int32), ("B", C.c_int32), ("M", C.c_int32), ("N1", C.c_int32), ("T", C.c_int32),
                ("Tcap_actions", C.c_int32), ("first_decode_step", C.c_int32), ("inv_ens", C.c_float), ("Rcap", C.c_int64),
                ("gprob", _vp), ("pval", _vp), ("tlen", _vp), ("actions", _vp), ("trPC", _vp), ("trCsel", _vp), ("trQ", _vp),
                ("trO", _vp), ("trLoad", _vp), ("trSlot", _vp), ("trA", _vp), ("trMask", _vp), ("trLse", _vp), ("Kmat", _vp), ("Vmat", _vp),
                ("PK", _vp), ("dK", _vp), ("dV", _vp), ("dPK", _vp), ("dpb", _vp), ("dQ1", _vp), ("dQ2", _vp), ("dwl", _vp),
                ("rowDU", _vp), ("dO", _vp), ("idx_prev", _vp), ("idx_first", _vp), ("rowW", _vp),
                ("T_dev", _vp), ("gprob_T", C.c_int32), ("tables_frozen", C.c_int32),
                ("mask_words", C.c_int32), ("mfma_mode", C.c_int32), ("ws", _vp), ("ws_floats", C.c_int64)]


class LocalWeights(C.Structure):
    _fields_ = [(n, _vp) for n in ("init_emb_w", "init_emb_b", "cur_token_emb", "Wq", "Wk", "Wv", "combine_w", "combine_b")]


class EncoderArgs(C.Structure):
    _fields_ = [("problem", C.c_int32), ("B", C.c_int32), ("N1", C.c_int32), ("n_layers", C.c_int32),
                ("ff_hidden", C.c_int32), ("save", C.c_int32), ("eps", C.c_float), ("precision", C.c_int32),
                ("xy", _vp), ("demand", _vp), ("W", EncWeights),
                ("enc", _vp), ("K", _vp), ("V", _vp), ("PK", _vp), ("pb", _vp), ("Q1", _vp), ("Q2", _vp), ("wl", _vp),
                ("ws", _vp), ("ws_floats", C.c_int64)]


class EncoderBwdArgs(C.Structure):
    _fields_ = [("fwd", EncoderArgs), ("g_enc", _vp), ("gK", _vp), ("gV", _vp), ("gPK", _vp), ("gpb", _vp),
                ("gQ1", _vp), ("gQ2", _vp), ("gwl", _vp), ("G", EncWeights), ("ws2", _vp), ("ws2_floats", C.c_int64)]


EXPORTS = ["elg_version", "elg_last_error", "elg_aug8", "elg_dist_matrix", "elg_nbr_tables", "elg_route_length",
           "elg_rollout_fwd", "elg_rollout_bwd", "elg_glimpse_rows_bwd", "elg_glimpse_bwd_fused", "elg_gemm_f32",
           "elg_gemm_f32_batched", "elg_pomo_loss", "elg_pomo_loss_grad", "elg_adam_step", "elg_local_bwd_rows",
           "elg_add_instnorm_fwd", "elg_add_instnorm_bwd",
           "elg_encoder_ws_floats", "elg_encoder_fwd", "elg_encoder_bwd_ws_floats", "elg_encoder_bwd",
           "elg_local_fold_fwd", "elg_local_fold_bwd", "elg_check_feasible", "elg_rollout_stats", "elg_decoder_bwd",
           "elg_decoder_bwd_ws_floats",
           "elg_rollout_scratch_floats", "elg_rollout_last_kernel"]

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension is mandatory (no CPU fallback). "
                              "Build it with `python -m elg_amd.build`.")
        # ONE HIP runtime per process: PyTorch bundles its own libamdhip64.so.7; load it first so that
        # libelg_hip.so's DT_NEEDED libamdhip64.so.7 resolves to the same runtime (shared streams,
        # device pointers) instead of pulling /opt/rocm's copy next to it.
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB_PATH)
        L.elg_version.restype = C.c_char_p
        L.elg_last_error.restype = C.c_char_p
        i, f = C.c_int, C.c_void_p
        L.elg_aug8.argtypes = [f, f, i, i, f]
        L.elg_dist_matrix.argtypes = [f, f, i, i, f]
        L.elg_nbr_tables.argtypes = [f, f, f, f, i, i, f]
        L.elg_route_length.argtypes = [f, f, f, i, i, i, i, i, f]
        L.elg_rollout_fwd.argtypes = [C.POINTER(RolloutArgs), f]
        L.elg_rollout_last_kernel.argtypes = []
        L.elg_rollout_last_kernel.restype = C.c_int
        L.elg_rollout_bwd.argtypes = [C.POINTER(BwdArgs), f]
        L.elg_glimpse_rows_bwd.argtypes = [f, f, f, f, f, f, f, i, i, i, C.c_int64, C.c_int64, f]
        L.elg_glimpse_bwd_fused.argtypes = [f, f, f, f, f, f, f, f, f, f, i, i, i, C.c_int64, C.c_int64, C.c_int64, i, f]
        L.elg_gemm_f32.argtypes = [f, f, f, f, i, i, i, i, i, i, i, i, i, i, f, f]
        i64, fl = C.c_int64, C.c_float
        L.elg_gemm_f32_batched.argtypes = [f, f, f, i, i, i, i, i, i, i, i, i, i, i64, i64, i64, i64, i64, i64, fl, f]
        L.elg_gemm_f32_batched.restype = C.c_int
        L.elg_pomo_loss.argtypes = [f, f, i, i, i, i64, i64, f, f, f, f, f, f]
        L.elg_pomo_loss_grad.argtypes = [f, f, f, i, i, i, i64, i64, C.c_float, f, f, f, f, f, f]
        L.elg_add_instnorm_fwd.argtypes = [f, f, f, f, f, f, f, i, i, i, fl, f]
        L.elg_add_instnorm_bwd.argtypes = [f, f, f, f, f, f, f, i, i, i, f]
        L.elg_local_bwd_rows.argtypes = [f, f, f, f, f, i, i, i64, i, f, i, i, f]
        L.elg_adam_step.argtypes = [f, f, f, i, f, f, f, i64, fl, fl, fl, fl, fl, i64, fl, f]
        L.elg_encoder_ws_floats.argtypes = [i, i, i, i, i]
        L.elg_encoder_ws_floats.restype = i64
        L.elg_rollout_scratch_floats.argtypes = [i, i, i, i]
        L.elg_rollout_scratch_floats.restype = i64
        L.elg_encoder_bwd_ws_floats.argtypes = [i, i, i, i]
        L.elg_encoder_bwd_ws_floats.restype = i64
        L.elg_encoder_fwd.argtypes = [C.POINTER(EncoderArgs), f]
        L.elg_encoder_bwd.argtypes = [C.POINTER(EncoderBwdArgs), f]
        L.elg_local_fold_fwd.argtypes = [C.POINTER(LocalWeights), i, i, i, f, f]
        L.elg_local_fold_bwd.argtypes = [C.POINTER(LocalWeights), i, i, i, f, C.POINTER(LocalWeights), f]
        L.elg_check_feasible.argtypes = [f, i64, f, i, i, i, f, f]
        L.elg_rollout_stats.argtypes = [f, f, i, i, i, f, f, f]
        L.elg_decoder_bwd.argtypes = [C.POINTER(DecoderBwdArgs), f]
        L.elg_decoder_bwd_ws_floats.argtypes = [i, i64, i]
        L.elg_decoder_bwd_ws_floats.restype = i64
        L.elg_decoder_bwd.restype = C.c_int
        L.elg_check_feasible.restype = C.c_int
        L.elg_rollout_stats.restype = C.c_int
        L.elg_local_fold_fwd.restype = C.c_int
        L.elg_local_fold_bwd.restype = C.c_int
        L.elg_encoder_fwd.restype = C.c_int
        L.elg_encoder_bwd.restype = C.c_int
        for n in ("elg_aug8", "elg_dist_matrix", "elg_nbr_tables", "elg_route_length", "elg_rollout_fwd",
                  "elg_rollout_bwd", "elg_glimpse_rows_bwd", "elg_glimpse_bwd_fused", "elg_gemm_f32",
                  "elg_pomo_loss", "elg_pomo_loss_grad", "elg_adam_step", "elg_local_bwd_rows",
                  "elg_add_instnorm_fwd", "elg_add_instnorm_bwd"):
            getattr(L, n).restype = C.c_int
        _lib = L
    return _lib


def check(code: int, what: str):
    """Map ELG_E* codes to the Python exceptions the reference's protocol raises."""
    if code == ELG_OK:
        return
    msg = f"{what}: {lib().elg_last_error().decode()}"
    if code == ELG_EINVAL:
        raise ValueError(msg)
    if code == ELG_ENOTIMPL:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
