#!/bin/bash
# plan sweep of the low-level convolutions with the diagnostic build (MI355_CONV_CT / _KSPLIT overrides; tools/build_diag.sh)
LIB=tools/_build/libmi355_unet_diag.so
for only in "128->256 @16" "512->256 @16" "256->128 @16" "256->256 @16" "512->512 @8"; do
  echo "== $only"
  echo "default: $(python tools/bench_kernels.py conv --reps 30 --only "$only" --lib $LIB 2>&1 | grep conv | awk '{print $(NF-8), $(NF-7), $(NF-6), $(NF-5), "plan", $(NF-3)}')"
  for ct in 1 2; do for ks in 1 2 4 8 16; do
    echo "ct=$ct ksplit=$ks: $(MI355_CONV_CT=$ct MI355_CONV_KSPLIT=$ks python tools/bench_kernels.py conv --reps 30 --only "$only" --lib $LIB 2>&1 | grep conv | awk '{print $(NF-8), $(NF-7), $(NF-6), $(NF-5), "plan", $(NF-3)}')"
  done; done
done
