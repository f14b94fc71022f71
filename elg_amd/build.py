"""Build libelg_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

`python -m elg_amd.build` or `elg_amd.build.build()`.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libelg_hip.so")
SOURCES = ["elg_fwd.hip", "elg_fwd_coop.hip", "elg_bwd.hip", "elg_gemm.hip", "elg_train.hip", "elg_local.hip", "elg_encoder.hip", "elg_enc.hip", "elg_enc_fused.hip", "elg_fold.hip", "elg_dbwd.hip"]
HEADERS = ["elg_common.h", "elg_rollout.h", "elg_coop.h", "elg_bwd_internal.h", "elg_bf16.h", "elg_enc_internal.h", os.path.join("..", "..", "include", "elg_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
         "-Wno-unused-value"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    cmd = [_hipcc(), *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = LIB + ".sha"
    dg = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dg:
        return LIB
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if verbose:
        print(f"[elg_amd.build] hipcc {' '.join(srcs)} -> {LIB}", file=sys.stderr)
    with cf.ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        objs = list(ex.map(_compile, srcs))
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp, "w") as f:
        f.write(dg)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
