#!/bin/bash
# usage: run_prof.sh TAG  -> encoder tests, then rocprof of bench
TAG=$1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
grep '"metric"' $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log | cut -c1-200
