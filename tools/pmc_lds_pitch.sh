#!/bin/bash
# LDS bank-conflict counters of the cooperative rollout for two builds of the library (the shipped pitches and a variant built with
# -DELG_CO_QP=... -DELG_CO_SP=... -DELG_CO_XPAD=... -DELG_CL_P=... -DELG_CL_Q=...): one rocprofv3 --pmc pass each over
# tools/time_coop_variants.py (same seeds, same tours).   bash tools/pmc_lds_pitch.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export ELG_HIP_LIB=$lib
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$tag -o pmc -- python3 $GRAFT_REPO_ROOT/tools/time_coop_variants.py 64 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$tag.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$tag.json $(find $GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$tag -name "*.db") | grep coop
done
