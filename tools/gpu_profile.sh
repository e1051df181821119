#!/bin/bash
# usage: tools/gpu_profile.sh <tag> : rocprofv3 kernel trace + stats of the eager step; per-launch listing of one step
tag=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${tag}_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --steps 3 --warmup 2 --settle-s 0 --no-graph --no-cpu-baseline --no-probe > $OUT/run.log 2>&1
echo "rocprof rc=$?"
cd $ROOT
python3 tools/step_trace.py $OUT gpurun_out/${tag}_step_trace.txt
python3 tools/prof_summary.py $OUT 5 45 > gpurun_out/${tag}_prof_summary.txt 2>&1
head -60 gpurun_out/${tag}_prof_summary.txt
