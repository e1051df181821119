#!/bin/bash
# usage: tools/gpu_r4c_160.sh <tag>: kernel statistics of the eager 1x24x160^3 step (BASELINE configs[4]), bf16 and fp8
tag=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for dt in bf16 fp8; do
  OUT=$ROOT/gpurun_out/${tag}_stats160_$dt; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --size 160 --dtype $dt --steps 30 --warmup 5 --settle-s 0 --no-graph --no-cpu-baseline --no-probe > $OUT/run.log 2>&1 ) || exit 1
  cp $OUT/p_kernel_stats.csv gpurun_out/${tag}_${dt}_160_kernel_stats.csv
  python3 tools/prof_summary.py $OUT 35 30 > gpurun_out/${tag}_${dt}_160_summary.txt; head -12 gpurun_out/${tag}_${dt}_160_summary.txt
  rm -f $OUT/p_kernel_trace.csv
done
