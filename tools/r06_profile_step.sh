#!/bin/bash
# kernel trace of the training step in both modes (rocprofv3 --kernel-trace --stats), summaries under gpurun_out/r06/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf $O/prof_f32 $O/prof_bf16
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_f32 -o b -- python3 tools/prof_kernels.py 64 train 25 > $O/prof_f32.log 2>&1
python3 tools/rocpd_stats.py $(ls $O/prof_f32/*.db | head -1) $O/${TAG:-r06}_f32_step_kernel_stats.csv 25 > $O/${TAG:-r06}_f32_step_kernel_stats.txt
python3 tools/step_timeline.py $(ls $O/prof_f32/*.db | head -1) 15 > $O/${TAG:-r06}_step_timeline_f32.txt
ELG_FWD_MODE=bf16 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o b -- python3 tools/prof_kernels.py 64 train 25 > $O/prof_bf16.log 2>&1
python3 tools/rocpd_stats.py $(ls $O/prof_bf16/*.db | head -1) $O/${TAG:-r06}_bf16_step_kernel_stats.csv 25 > $O/${TAG:-r06}_bf16_step_kernel_stats.txt
python3 tools/step_timeline.py $(ls $O/prof_bf16/*.db | head -1) 15 > $O/${TAG:-r06}_step_timeline_bf16.txt
rm -rf $O/prof_f32 $O/prof_bf16
head -14 $O/${TAG:-r06}_f32_step_kernel_stats.txt; head -12 $O/${TAG:-r06}_bf16_step_kernel_stats.txt
