#!/bin/bash
# usage: tools/gpu_variants.sh "<only pattern>" name1 name2 ... : interleaved kernel timings of library variants (tools/_build/libmi355_unet_<name>.so)
only=$1; shift
for r in 0 1; do
  echo "round $r shipped: $(timeout -k 10 120 python tools/bench_kernels.py conv --reps 50 --only "$only" 2>&1 | grep conv)"
  for v in "$@"; do
    echo "round $r $v: $(timeout -k 10 120 python tools/bench_kernels.py conv --reps 50 --only "$only" --lib tools/_build/libmi355_unet_$v.so 2>&1 | grep conv)"
  done
done
