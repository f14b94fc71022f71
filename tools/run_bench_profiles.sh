#!/bin/bash
# Round bench line + rocprof summaries (run on the GPU box):  bash tools/run_bench_profiles.sh rNN
TAG=$1
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
tail -c 3000 gpurun_out/${TAG}_bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_bench -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_secondary -o sec -- python3 $R/tools/secondary_only.py > $R/gpurun_out/prof_${TAG}_secondary.log 2>&1
tail -2 $R/gpurun_out/prof_${TAG}_secondary.log
