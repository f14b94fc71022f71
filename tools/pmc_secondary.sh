#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the secondary workloads (TSP-500 batch 16 pomo 500; VRPLIB X-n1001 x8), one counter per
# run:  bash tools/pmc_secondary.sh TAG   -> gpurun_out/pmc_TAG_sec_{FETCH_SIZE,WRITE_SIZE}/
TAG=$1
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sec_${grp} -o pmc -- python3 $GRAFT_REPO_ROOT/tools/secondary_only.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sec_${grp}.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sec_${grp}.log
done
