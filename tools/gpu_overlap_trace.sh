#!/bin/bash
# usage: tools/gpu_overlap_trace.sh <tag>: kernel trace of the one-rank RCCL rehearsal; does the all-reduce kernel overlap the backward kernels?
tag=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${tag}_otrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --force-collectives --steps 4 --warmup 2 --settle-s 0 --no-cpu-baseline --no-probe > $OUT/run.log 2>&1
echo "rocprof rc=$?"
cd $ROOT
python3 tools/overlap_report.py $OUT > gpurun_out/${tag}_overlap_force_collectives.txt 2>&1
cat gpurun_out/${tag}_overlap_force_collectives.txt
rm -f $OUT/*kernel_trace.csv
