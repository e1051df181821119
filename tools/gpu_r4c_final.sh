#!/bin/bash
# usage: tools/gpu_r4c_final.sh <tag>: artefacts of round 4's third session -- the bench lines (gpu_artifacts.sh part b) and the kernel
# statistics of the eager step; the convolution sources are unchanged since r04b, so its PMC traffic records stay valid (bench.py checks the hash)
tag=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/gpu_artifacts.sh $tag b
OUT=$ROOT/gpurun_out/${tag}_stats; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --steps 40 --warmup 5 --settle-s 0 --no-graph --no-cpu-baseline --no-probe > $OUT/run.log 2>&1 )
cp $OUT/p_kernel_stats.csv gpurun_out/${tag}_bf16_default_kernel_stats.csv
python3 tools/prof_summary.py $OUT 45 30 > gpurun_out/${tag}_bf16_default_summary.txt; head -9 gpurun_out/${tag}_bf16_default_summary.txt
rm -f $OUT/p_kernel_trace.csv
