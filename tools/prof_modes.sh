#!/bin/bash
# per-kernel averages of the training step for a list of ELG_BWD_MFMA_MODE values:  bash tools/prof_modes.sh "0 1 2"
cd /tmp && export TMPDIR=/tmp
for m in $1; do
  export ELG_BWD_MFMA_MODE=$m
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_mode$m -o m$m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_mode$m.log 2>&1
  echo "== mode $m"; grep '"metric"' $GRAFT_REPO_ROOT/gpurun_out/prof_mode$m.log | cut -c1-160
  db=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_mode$m -name '*.db' | head -1)
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db $GRAFT_REPO_ROOT/gpurun_out/mode${m}_kernel_stats.csv 25 | grep -i "glimpse\|pointer_bwd\|local_bwd\|rollout_fwd" | cut -c1-60,78-140
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mode$m
done
