#!/bin/bash
# A/B of the cooperative rollout kernels (lockstep vs split-group) on the GPU box: parity tests, then launch times.
O=gpurun_out/r06; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_coop.py tests/test_gpu_logits.py tests/test_gpu_bench_shape.py tests/test_gpu_protocol.py tests/test_gpu_backward.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -15) > $O/coop2_pytest.log
cat $O/coop2_pytest.log
: > $O/coop2_ab.txt
for k in lockstep split; do
  for st in ${STAGGERS:-3}; do
    [ $k = lockstep ] && [ $st != 3 ] && continue
    echo "== kernel $k stagger $st" >> $O/coop2_ab.txt
    ELG_COOP_KERNEL=$k ELG_COOP_STAGGER=$st timeout 120 python tools/time_coop_variants.py 2>&1 | grep train= >> $O/coop2_ab.txt
    ELG_FWD_MODE=bf16 ELG_COOP_KERNEL=$k ELG_COOP_STAGGER=$st timeout 120 python tools/time_coop_variants.py 2>&1 | grep train= | sed 's/^/bf16 /' >> $O/coop2_ab.txt
  done
done
cat $O/coop2_ab.txt
