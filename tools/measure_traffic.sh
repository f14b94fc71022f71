#!/bin/bash
# HBM traffic of the persistent rollout kernel ON THE BENCH WORKLOAD ITSELF (bench.py's own steps), collected as the guide
# prescribes: one rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE cannot share a pass), --kernel-trace only.
#   bash tools/measure_traffic.sh        (on the GPU box)   -> gpurun_out/roofline_traffic.json  (copy to profiles/)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/traffic_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/traffic_$ctr -o pmc -- python3 $R/bench.py --steps 10 --warmup 3 \
      --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $R/gpurun_out/traffic_$ctr.log 2> $R/gpurun_out/traffic_$ctr.err
  tail -c 300 $R/gpurun_out/traffic_$ctr.log
done
python3 $R/tools/make_traffic_json.py $R/gpurun_out/roofline_traffic.json $R/gpurun_out/traffic_FETCH_SIZE $R/gpurun_out/traffic_WRITE_SIZE \
    $R/gpurun_out/traffic_FETCH_SIZE.log
