"""The C-ABI library builds for gfx950, loads, exports every symbol include/elg_hip.h declares, and the
ctypes mirrors of its structs have the C layout (checked against gcc's sizeof/offsetof).  No GPU needed."""
import ctypes as C
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "elg_hip.h")


def test_build_and_exports():
    from elg_amd import build
    lib_path = build.build()
    assert os.path.exists(lib_path)
    from elg_amd import _lib
    L = _lib.lib()
    declared = set(re.findall(r"^\s*(?:const\s+char\*|int|int64_t)\s+(elg_\w+)\s*\(", open(HDR).read(), re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.elg_version()
    # the shared object really contains gfx950 device code
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", lib_path], capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out


def test_struct_layout_matches_c():
    from elg_amd import _lib
    pairs = [("elg_rollout_args", _lib.RolloutArgs, ["problem", "variant", "dump_logits", "euclidean", "ens", "Kens", "seed", "Kmat", "loc", "st_vis", "full_probs", "trMask", "scratch", "trLse"]),
             ("elg_bwd_args", _lib.BwdArgs, ["T", "member", "gprob", "rowA", "rowLoad", "gloc", "row_stride"]),
             ("elg_enc_layer", _lib.EncLayer, ["Wq", "bc", "W1", "b2"]),
             ("elg_enc_weights", _lib.EncWeights, ["emb_depot_w", "emb_w", "layer", "dec_Wq_first", "dec_bc"]),
             ("elg_encoder_args", _lib.EncoderArgs, ["problem", "eps", "xy", "W", "enc", "Q2", "wl", "ws", "ws_floats"]),
             ("elg_encoder_bwd_args", _lib.EncoderBwdArgs, ["fwd", "g_enc", "gpb", "gwl", "G", "ws2", "ws2_floats"]),
             ("elg_decoder_bwd_args", _lib.DecoderBwdArgs, ["problem", "inv_ens", "Rcap", "gprob", "trLse", "Kmat", "dwl", "rowDU", "rowW", "T_dev", "gprob_T", "tables_frozen", "mask_words", "mfma_mode", "ws", "ws_floats"]),
             ("elg_local_weights", _lib.LocalWeights, ["init_emb_w", "cur_token_emb", "combine_b"])]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "elg_hip.h"\nint main(){\n'
    for cname, _, fields in pairs:
        src += f'printf("%zu\\n", sizeof({cname}));\n'
        for f in fields:
            src += f'printf("%zu\\n", offsetof({cname}, {f}));\n'
    src += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = [int(x) for x in subprocess.check_output([exe], text=True).split()]
    i = 0
    for cname, ct, fields in pairs:
        assert out[i] == C.sizeof(ct), cname
        i += 1
        for f in fields:
            assert out[i] == getattr(ct, f).offset, (cname, f)
            i += 1


def test_loc_layout_constants():
    from elg_amd import _lib
    txt = open(HDR).read()
    for name in ("LA", "LT", "LAV", "LCV", "LWC", "LBC", "LWE", "LPE", "SIZE", "ROWS"):
        v = int(re.search(rf"#define ELG_LOC_{name}\s+(\d+)", txt).group(1))
        assert v == getattr(_lib, f"LOC_{name}"), name


def test_no_cpu_fallback():
    """The product path refuses to run without the GPU instead of silently computing on the host."""
    import pytest
    import torch
    from elg_amd import engine as eng
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    with pytest.raises(RuntimeError):
        CVRPEnv(4, "cpu")
    with pytest.raises(RuntimeError):
        eng.nbr_tables(torch.rand(1, 5, 2))
