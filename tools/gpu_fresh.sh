#!/bin/bash
tag=$1
run() { name=$1; shift; timeout -k 10 200 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${tag}_$name.json") if l.startswith("{")][-1]); print("$name", round(d["ms_per_step"], 3), round(d["value"], 2), d.get("nondefault"))
except Exception as e: print("$name", "ERR", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
B="python bench.py --no-cpu-baseline --no-probe --steps 100"
run resident $B
run fresh_hostsync $B --fresh-batch
run fresh_gpuwait $B --fresh-batch --fresh-diag gpu_wait
run fresh_y_only $B --fresh-batch --fresh-diag y_only
run fresh_hostsync2 $B --fresh-batch
