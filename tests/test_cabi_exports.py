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
    declared = set(re.findall(r"^\s*(?:const\s+char\*|int)\s+(elg_\w+)\s*\(", open(HDR).read(), re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.elg_version()
    # the shared object really contains gfx950 device code
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", lib_path], capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out


def test_struct_layout_matches_c():
    from elg_amd import _lib
    fields = ["problem", "seed", "Kmat", "loc", "st_vis", "full_probs"]
    bfields = ["T", "gprob", "rowA", "rowLoad", "gloc"]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "elg_hip.h"\nint main(){\n'
    src += 'printf("%zu %zu\\n", sizeof(elg_rollout_args), sizeof(elg_bwd_args));\n'
    for f in fields:
        src += f'printf("%zu\\n", offsetof(elg_rollout_args, {f}));\n'
    for f in bfields:
        src += f'printf("%zu\\n", offsetof(elg_bwd_args, {f}));\n'
    src += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe], text=True).split()
    sizes = [int(x) for x in out]
    assert sizes[0] == C.sizeof(_lib.RolloutArgs) and sizes[1] == C.sizeof(_lib.BwdArgs)
    i = 2
    for f in fields:
        assert sizes[i] == getattr(_lib.RolloutArgs, f).offset, f
        i += 1
    for f in bfields:
        assert sizes[i] == getattr(_lib.BwdArgs, f).offset, f
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
