#!/bin/bash
# usage: tools/gpu_profile_graph.sh <tag> : rocprofv3 kernel trace of hipGraph REPLAYS of the step; per-launch listing + per-kernel table
tag=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${tag}_gtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --steps 6 --warmup 2 --settle-s 0 --no-cpu-baseline --no-probe > $OUT/run.log 2>&1
echo "rocprof rc=$?"
cd $ROOT
python3 tools/step_trace.py $OUT gpurun_out/${tag}_graph_step_trace.txt
python3 - <<PY
import collections
tot = collections.defaultdict(lambda: [0, 0.0])
span = None
for l in open("gpurun_out/${tag}_graph_step_trace.txt"):
    if l.startswith("#"):
        print(l.strip()); continue
    p = l.split()
    name = " ".join(p[7:]).split("<")[0]
    tot[name][0] += 1; tot[name][1] += float(p[3])
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:8.3f} ms {n:5d} launches avg {t / n:8.1f} us  {k}")
PY
rm -f $OUT/*kernel_trace.csv
