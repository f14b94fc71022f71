#!/bin/bash
# round-4 first GPU pass: gate tests, bench line, kernel stats, traffic
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_pytest_gpu.log
tail -5 gpurun_out/r04_pytest_gpu.log
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err; tail -c 1500 gpurun_out/r04_bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_bench -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $R/gpurun_out/prof_r04_bench.log 2>&1
bash $R/tools/measure_traffic.sh
