#!/bin/bash
# LazyPool (max-pool backward inside the norm backward kernels): tests, LDS probe with the halo read patterns, step A/B
tag=${1:-r4d}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "maxpool or norm or deferred or hipgraph or golden or sinks or oracle" > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "^E  |passed|failed|FAILED|Error" gpurun_out/${tag}_pytest.log | head -30
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 tests/diag/lds_rw_bench > gpurun_out/${tag}_lds_rw.txt 2>&1; cat gpurun_out/${tag}_lds_rw.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_ldspmc -o lds -f csv -- $GRAFT_REPO_ROOT/tests/diag/lds_rw_bench > $GRAFT_REPO_ROOT/gpurun_out/${tag}_ldspmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY | tee gpurun_out/${tag}_ldspmc.txt
import csv, glob, collections
for f in glob.glob("gpurun_out/${tag}_ldspmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k, {c: round(x) for c, x in v.items()})
PY
bash tools/ab_flags.sh 3 "" "--eager-pool-bwd" 2>&1 | tee gpurun_out/${tag}_ab.txt
