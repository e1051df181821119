#!/bin/bash
# HBM-side traffic per launch of the kernels of the eager training step, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (kernel trace only), gfx950 correction (FETCH_SIZE counts 16-B/lane streams at half
# their bytes) applied in the summary.  Writes profiles/<tag>_traffic.json (read by bench.py: `roofline.traffic`, valid
# while the kernel sources keep the recorded hash) and profiles/<tag>_pmc_step_traffic.txt.
# usage: tools/pmc_traffic.sh <tag> [size] [dtype]      (run on the GPU box, from the repository root)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}; SIZE=${2:-128}; DTYPE=${3:-bf16}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT $ROOT/profiles
cd /tmp && export TMPDIR=/tmp
for name in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $name --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 $ROOT/bench.py --size $SIZE --dtype $DTYPE --steps 2 --warmup 1 --settle-s 0 --no-graph --no-cpu-baseline --no-probe > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -n 3 $OUT/$name.log; exit 1; }
done
cd $ROOT && python3 tools/pmc_traffic.py $OUT $TAG $SIZE $DTYPE
