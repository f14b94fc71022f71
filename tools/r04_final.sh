#!/bin/bash
# round-4 record: GPU gate, bench line, kernel stats of the bench command, traffic of the rollout kernel, secondary stats
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r04_pytest_gpu.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r04_pytest_gpu.log
tail -4 gpurun_out/r04_pytest_gpu.log
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err; tail -c 600 gpurun_out/r04_bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_bench -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $R/gpurun_out/prof_r04_bench.log 2>&1
python3 $R/tools/rocpd_stats.py $R/gpurun_out/prof_r04_bench/bench_results.db $R/gpurun_out/r04_bench_kernel_stats.csv 25 | head -16
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_secondary -o sec -- python3 $R/tools/secondary_only.py > $R/gpurun_out/prof_r04_secondary.log 2>&1
python3 $R/tools/rocpd_stats.py $R/gpurun_out/prof_r04_secondary/sec_results.db $R/gpurun_out/r04_secondary_kernel_stats.csv | head -8
bash $R/tools/measure_traffic.sh
