#!/bin/bash
# focused gate + A/B of the f32 glimpse backward (new kernel vs ELG_GLIMPSE_F32_OLD=1)
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_glue.py tests/test_gpu_backward.py tests/test_gpu_protocol.py tests/test_gpu_coop.py tests/test_gpu_train_large.py -x -q > gpurun_out/r04_ab_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_ab_pytest.log
tail -4 gpurun_out/r04_ab_pytest.log
for v in new old; do
  if [ $v = old ]; then export ELG_GLIMPSE_F32_OLD=1; else unset ELG_GLIMPSE_F32_OLD; fi
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > gpurun_out/r04_ab_$v.json 2> gpurun_out/r04_ab_$v.err
  python -c "import json; d=json.loads([l for l in open('gpurun_out/r04_ab_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
unset ELG_GLIMPSE_F32_OLD
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_ab -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $R/gpurun_out/prof_r04_ab.log 2>&1
python3 $R/tools/rocpd_stats.py $R/gpurun_out/prof_r04_ab/bench_results.db $R/gpurun_out/r04_ab_kernel_stats.csv 25 | head -24
