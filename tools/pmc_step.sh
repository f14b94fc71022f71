#!/bin/bash
# SQ counter passes over a few whole training steps (every kernel of the step): bash tools/pmc_step.sh TAG
TAG=$1
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for grp in "$P1" "$P2" "$P3"; do
  name=$(echo abc | cut -c$((i+1)))
  rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_step_${name} -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_kernels.py 64 train 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_step_${name}.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_step_${name}.log
  i=$((i+1))
done
