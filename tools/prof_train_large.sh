#!/bin/bash
# per-kernel averages of a training step above 128 nodes:  bash tools/prof_train_large.sh "cvrp 200 32"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tl -o tl -- python3 $GRAFT_REPO_ROOT/tools/time_train_large.py $1 rows > $GRAFT_REPO_ROOT/gpurun_out/prof_tl.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_tl.log | cut -c1-300
db=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_tl -name '*.db' | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db $GRAFT_REPO_ROOT/gpurun_out/train_large_kernel_stats.csv 6 | head -24 | cut -c1-70,78-140
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tl
