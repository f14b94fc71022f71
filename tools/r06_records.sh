#!/bin/bash
# Round-6 measurement records, all from one box: run on the GPU box (gpurun), results under gpurun_out/r06rec/ -> copy to profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06rec
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
# 1. the GPU gate + smoke
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "Warning\|pickle.load\|^$" > $O/r06_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep graft >> $O/r06_pytest_gpu.log
tail -2 $O/r06_pytest_gpu.log
cp $R/gpurun_out/parity_r06.json $O/parity_r06.json 2>/dev/null
# 2. HBM traffic of the rollout launch (two PMC passes, bounded) -- BEFORE the bench line, which reads the record
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/traffic_$ctr
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/traffic_$ctr -o pmc -- python3 $R/bench.py --steps 10 --warmup 3 \
      --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $R/gpurun_out/traffic_$ctr.log 2> $R/gpurun_out/traffic_$ctr.err
done
python3 tools/make_traffic_json.py $O/roofline_traffic.json $R/gpurun_out/traffic_FETCH_SIZE $R/gpurun_out/traffic_WRITE_SIZE $R/gpurun_out/traffic_FETCH_SIZE.log | cut -c1-300
cp $O/roofline_traffic.json $R/profiles/roofline_traffic.json
# 3. the bench line (driver-equivalent run)
timeout 900 python bench.py > $O/bench.out 2> $O/bench.err; tail -1 $O/bench.out > $O/r06_bench_line.json
python3 - <<PY
import json
o = json.load(open("$O/r06_bench_line.json"))
print("bench:", o["value"], o["ms_per_step"], o.get("value_fast"), o.get("value_bf16"), o["roofline"]["frac"], o["roofline"]["launch_ms"],
      {k: v for k, v in o["config"]["per_rank"].items() if "host" in k}, "traffic" , (o["roofline"]["traffic"] or {}).get("over_algorithmic"))
PY
# 4. kernel trace of the bench's own steps + step timelines in both modes
rm -rf $O/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python3 bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-secondary --no-fast --sustain-s 0 > $O/prof_bench.log 2>&1
python3 tools/rocpd_stats.py $(ls $O/prof_bench/*.db | head -1) $O/r06_bench_kernel_stats.csv 30 > $O/r06_bench_kernel_stats.txt
python3 tools/step_timeline.py $(ls $O/prof_bench/*.db | head -1) 15 > $O/r06_step_timeline_f32.txt
rm -rf $O/prof_bf16
ELG_FWD_MODE=bf16 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o b -- python3 tools/prof_kernels.py 64 train 25 > $O/prof_bf16.log 2>&1
python3 tools/rocpd_stats.py $(ls $O/prof_bf16/*.db | head -1) $O/r06_bf16_step_kernel_stats.csv 25 > $O/r06_bf16_step_kernel_stats.txt
python3 tools/step_timeline.py $(ls $O/prof_bf16/*.db | head -1) 15 > $O/r06_step_timeline_bf16.txt
head -8 $O/r06_bench_kernel_stats.txt
# 5. SQ counters of the training step (three passes, SQ block only)
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32"
i=0; dbs=""
for grp in "$P1" "$P2" "$P3"; do
  name=$(echo abc | cut -c$((i+1)))
  rm -rf $R/gpurun_out/pmc_r06_step_${name}
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_r06_step_${name} -o pmc -- python3 $R/tools/prof_kernels.py 64 train 4 > $R/gpurun_out/pmc_r06_step_${name}.log 2>&1
  dbs="$dbs $(find $R/gpurun_out/pmc_r06_step_${name} -name '*.db' | head -1)"
  i=$((i+1))
done
python3 tools/rocpd_pmc.py $O/r06_pmc_sq_step.json $dbs | head -4 | cut -c1-400
# 6. encoder alone, cooperative kernel A/B (lockstep vs split-group)
(timeout 100 python tools/time_encoder.py; ELG_FWD_MODE=bf16 timeout 100 python tools/time_encoder.py) 2>&1 | grep us > $O/r06_encoder_alone.txt
for k in lockstep split; do echo "== kernel $k"; ELG_COOP_KERNEL=$k timeout 120 python tools/time_coop_variants.py 2>&1 | grep train=; ELG_FWD_MODE=bf16 ELG_COOP_KERNEL=$k timeout 120 python tools/time_coop_variants.py 2>&1 | grep train= | sed 's/^/bf16 /'; done > $O/r06_coop_split_vs_lockstep_final.txt
rm -rf $O/prof_bench $O/prof_bf16
ls $O
