#!/bin/bash
# conv_lowg_kernel's LDS write order (round 4): parity, micro-benchmark A/B (plain = round 3's order), LDS counters, step A/B
tag=${1:-r4c}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "k3_ or conv_fwd_bwd" > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "^E  |passed|failed|FAILED|Error" gpurun_out/${tag}_pytest.log | head -30
[ $rc -ne 0 ] && exit $rc
for v in plain new plain new; do
  for only in "@16^3" "@8^3"; do
    timeout -k 10 120 python tools/bench_kernels.py conv --reps 50 --only "$only" --lib tools/_build/libmi355_unet_$v.so 2>&1 | grep "conv fwd" | sed "s/^/[$v] /"
  done
done | tee gpurun_out/${tag}_micro.txt
PMC_DTYPE=bf16 bash tools/pmc_conv.sh 0 gpurun_out/${tag}_pmc "512->256 @16^3" sq > gpurun_out/${tag}_pmc.txt 2>&1; tail -30 gpurun_out/${tag}_pmc.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > /dev/null 2>&1
for r in 0 1 2; do
  for v in plain new; do
    ms=$(python bench.py --steps 100 --no-cpu-baseline --no-probe --lib tools/_build/libmi355_unet_$v.so 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'],3))")
    echo "round $r [$v] $ms ms"
  done
done | tee gpurun_out/${tag}_ab.txt
