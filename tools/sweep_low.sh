#!/bin/bash
# plan sweep of the low-level convolutions with the diagnostic build (MI355_CONV_SHAPE / _CT / _KSPLIT)
LIB=tools/_build/libmi355_unet_diag.so
run() { # only shapes...
  only=$1; shift
  echo "== $only"
  echo "default: $(python tools/bench_kernels.py conv --reps 30 --only "$only" --lib $LIB 2>&1 | grep conv | awk '{print $(NF-8), $(NF-7), $(NF-6), $(NF-5), "plan", $(NF-3)}')"
  for sh in "$@"; do for ct in 1 2; do for ks in 0 2 4 8 16; do
    echo "shape=$sh ct=$ct ksplit=$ks: $(MI355_CONV_SHAPE=$sh MI355_CONV_CT=$ct MI355_CONV_KSPLIT=$ks python tools/bench_kernels.py conv --reps 30 --only "$only" --lib $LIB 2>&1 | grep conv | awk '{print $(NF-8), $(NF-7), $(NF-6), $(NF-5), "plan", $(NF-3)}')"
  done; done; done
}
run "128->256 @16" 1 12
run "512->256 @16" 1 12
run "256->128 @16" 1 12
run "256->256 @16" 1 12
run "512->512 @8" 2 13
