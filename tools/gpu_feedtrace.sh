#!/bin/bash
tag=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${tag}_feedtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --fresh-batch --steps 20 --warmup 3 --settle-s 0 --no-cpu-baseline --no-probe > $OUT/run.log 2>&1
echo "rc=$?"; tail -2 $OUT/run.log | cut -c1-300
cd $ROOT; python3 tools/feed_trace.py $OUT | tee gpurun_out/${tag}_feedtrace.txt | head -60
rm -f $OUT/*kernel_trace.csv      # (large)
