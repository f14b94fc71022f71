#!/bin/bash
# Round-5 measurement records, all from one box: run on the GPU box (gpurun), results under gpurun_out/r05/ -> copy to profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
# 1. the GPU gate
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "Warning\|pickle.load\|^$" > $O/r05_pytest_gpu.log
tail -1 $O/r05_pytest_gpu.log
# 2. the bench line (driver-equivalent run)
timeout 900 python bench.py > $O/bench.out 2> $O/bench.err; tail -1 $O/bench.out > $O/r05_bench_line.json
# 3. kernel trace of the bench's own steps
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python3 bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $O/prof_bench.log 2>&1
python3 tools/rocpd_stats.py $(ls $O/prof_bench/*.db | head -1) $O/r05_bench_kernel_stats.csv 30 > $O/r05_bench_kernel_stats.txt
python3 tools/step_timeline.py $(ls $O/prof_bench/*.db | head -1) 15 > $O/r05_step_timeline_f32.txt
# 4. the same for the bf16 mode's step
ELG_FWD_MODE=bf16 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o b -- python3 tools/prof_kernels.py 64 train 25 > $O/prof_bf16.log 2>&1
python3 tools/rocpd_stats.py $(ls $O/prof_bf16/*.db | head -1) $O/r05_bf16_step_kernel_stats.csv 25 > $O/r05_bf16_step_kernel_stats.txt
python3 tools/step_timeline.py $(ls $O/prof_bf16/*.db | head -1) 15 > $O/r05_step_timeline_bf16.txt
# 5. encoder alone: times, in-kernel phase clock, micro-benchmarks
(timeout 100 python tools/time_encoder.py; ELG_FWD_MODE=bf16 timeout 100 python tools/time_encoder.py; ELG_ENC_FUSED=0 timeout 100 python tools/time_encoder.py) 2>&1 | grep us > $O/r05_encoder_alone.txt
timeout 200 python tools/stamp_enc.py 2>&1 | grep -v amdgpu > $O/r05_encoder_phase_clock.txt
timeout 100 tools/_bench/ldbench > $O/r05_ldbench.txt 2>&1
timeout 100 tools/_bench/atbench > $O/r05_atbench.txt 2>&1
# 6. HBM traffic of the rollout launch (two PMC passes, bounded)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/traffic_$ctr
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/traffic_$ctr -o pmc -- python3 $R/bench.py --steps 10 --warmup 3 \
      --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $R/gpurun_out/traffic_$ctr.log 2> $R/gpurun_out/traffic_$ctr.err
done
python3 tools/make_traffic_json.py $O/roofline_traffic.json $R/gpurun_out/traffic_FETCH_SIZE $R/gpurun_out/traffic_WRITE_SIZE $R/gpurun_out/traffic_FETCH_SIZE.log
rm -rf $O/prof_bench $O/prof_bf16
ls $O
