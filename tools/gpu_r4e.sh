#!/bin/bash
# full GPU suite, deconv micro-benchmark A/B (gather = round 3's dispatch), then the step A/B of the fused pool forward
tag=${1:-r4e}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "^E  |passed|failed|FAILED|Error" gpurun_out/${tag}_pytest.log | head -30
[ $rc -ne 0 ] && exit $rc
for v in gather new gather new; do
  timeout -k 10 120 python tools/bench_kernels.py deconv --reps 50 --only "upcat" --lib tools/_build/libmi355_unet_$v.so 2>&1 | grep "deconv" | sed "s/^/[$v] /"
done | tee gpurun_out/${tag}_deconv.txt
bash tools/ab_flags.sh 3 "" "--separate-pool" "--separate-pool --eager-pool-bwd --immediate-reduce" "--lib tools/_build/libmi355_unet_gather.so" "--lib tools/_build/libmi355_unet_new.so" 2>&1 | tee gpurun_out/${tag}_ab.txt
