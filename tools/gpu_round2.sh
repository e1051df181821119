#!/bin/bash
# usage: tools/gpu_round2.sh <tag>   -- tests, ablation variants of the marching convs, step bench, PMC passes
tag=$1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/${tag}_pytest.log
tail -8 gpurun_out/${tag}_pytest.log
{
for only in "96->32" "64->64 @64" "128->64 @64"; do
  echo "== $only: shipped"; timeout -k 10 120 python tools/bench_kernels.py conv --reps 30 --only "$only" 2>&1 | grep conv
  for v in nostore nodmaact nodmaw nobarrier nomem; do
    [ -f tools/_build/libmi355_unet_$v.so ] || continue
    echo "== $only: $v"; timeout -k 10 120 python tools/bench_kernels.py conv --reps 30 --only "$only" --lib tools/_build/libmi355_unet_$v.so 2>&1 | grep conv
  done
done
for only in "32->32 @128" "32->96"; do
  echo "== $only: shipped"; timeout -k 10 120 python tools/bench_kernels.py conv --reps 30 --only "$only" 2>&1 | grep conv
  for v in marchnodma marchnomem; do
    [ -f tools/_build/libmi355_unet_$v.so ] || continue
    echo "== $only: $v"; timeout -k 10 120 python tools/bench_kernels.py conv --reps 30 --only "$only" --lib tools/_build/libmi355_unet_$v.so 2>&1 | grep conv
  done
done
} > gpurun_out/${tag}_ablate.log 2>&1
cat gpurun_out/${tag}_ablate.log
timeout -k 10 300 python tools/bench_kernels.py conv > gpurun_out/${tag}_kernels.log 2>&1
cat gpurun_out/${tag}_kernels.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
bash tools/pmc_conv.sh -1 gpurun_out/${tag}_pmc_96to32 "96->32" sq > gpurun_out/${tag}_pmc.log 2>&1
grep -A9 "marchg" gpurun_out/${tag}_pmc.log | head -24
