#!/bin/bash
# bf16 pointer backward + the whole-step data-parallel test + full step timelines (every kernel).  Run on the GPU box.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chk2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_glue.py tests/test_gpu_coop.py tests/test_gpu_zz_dp.py -q -x 2>&1 | grep -v "Warning\|pickle.load\|^$" > $O/pytest.log
tail -5 $O/pytest.log
timeout 200 python tools/time_step_modes.py 200 2>&1 | grep "ms/step" > $O/step_ms.txt
cat $O/step_ms.txt
timeout 300 rocprofv3 --kernel-trace -d $O/prof_f32 -o b -- python3 tools/prof_kernels.py 64 train 25 > $O/prof_f32.log 2>&1
python3 tools/step_timeline.py $(ls $O/prof_f32/*.db | head -1) 0 > $O/timeline_f32_all.txt
ELG_FWD_MODE=bf16 timeout 300 rocprofv3 --kernel-trace -d $O/prof_bf16 -o b -- python3 tools/prof_kernels.py 64 train 25 > $O/prof_bf16.log 2>&1
python3 tools/step_timeline.py $(ls $O/prof_bf16/*.db | head -1) 0 > $O/timeline_bf16_all.txt
rm -rf $O/prof_f32 $O/prof_bf16
head -3 $O/timeline_bf16_all.txt
