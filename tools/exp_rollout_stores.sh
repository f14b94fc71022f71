#!/bin/bash
# Builds diagnostic copies of the library with saved-row stores of the cooperative rollout kernel compiled out (-DELG_EXP_SKIP=bits:
# 1 trQ, 2 trSlot / trF, 4 trO, 8 trPC / trCsel) -> tools/_diag/libelg_skip<bits>.so.  Run in the build container, then on the GPU:
#   for b in 0 1 2 3 15; do ELG_HIP_LIB=tools/_diag/libelg_skip$b.so python tools/time_coop_variants.py; done
# The training rows of such a build are incomplete: timing only, never a result.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_diag
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wno-unused-value"
OBJS=$(ls elg_amd/csrc/*.o | grep -v elg_fwd.o)
for b in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DELG_EXP_SKIP=$b -c elg_amd/csrc/elg_fwd.hip -o /tmp/elg_fwd_skip$b.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_diag/libelg_skip$b.so $OBJS /tmp/elg_fwd_skip$b.o
  echo tools/_diag/libelg_skip$b.so
done
