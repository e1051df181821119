#!/bin/bash
# HBM-side traffic of the kernels of one eager training step (separate --pmc passes, as the guide prescribes).
# usage: tools/pmc_step.sh <outdir>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for name in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $name --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-probe > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in acc.items():
    f = cs.get("FETCH_SIZE", [0]); w = cs.get("WRITE_SIZE", [0])
    rows.append((sum(f) * 2 + sum(w), k, len(f), sum(f) / max(1, len(f)), sum(w) / max(1, len(w))))
rows.sort(reverse=True)
print("kernel | launches | FETCH_SIZE KB/launch (raw; x2 on gfx950 for 16-B/lane streams) | WRITE_SIZE KB/launch")
for tot, k, n, fa, wa in rows[:25]:
    print(f"{k:60s} {n:5d} {fa:14.1f} {wa:14.1f}")
PY
