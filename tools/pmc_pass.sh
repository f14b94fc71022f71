#!/bin/bash
# Separate rocprofv3 --pmc passes (one counter group per run, no trace flags besides --kernel-trace) over a few training steps.
#   bash tools/pmc_pass.sh TAG   -> gpurun_out/pmc_TAG_{fetch,write,valu}/...
TAG=$1
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${name} -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_kernels.py 64 train 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${name}.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${name}.log
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_FETCH_SIZE
