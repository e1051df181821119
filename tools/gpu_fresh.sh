#!/bin/bash
tag=$1
run() { name=$1; shift; timeout -k 10 200 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${tag}_$name.json") if l.startswith("{")][-1]); print("$name", round(d["ms_per_step"], 3), round(d["value"], 2), d.get("nondefault"))
except Exception as e: print("$name", "ERR", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_modules.py -q -k "two_input_sets" 2>&1 | grep -E "^E |assert|Error|passed|failed" | head -20
B="python bench.py --no-cpu-baseline --no-probe --steps 100"
run resident $B
run fresh $B --fresh-batch
run fresh_after0 $B --fresh-batch --fresh-copy-after 0
run fresh_after1 $B --fresh-batch --fresh-copy-after 1
run fresh_after2 $B --fresh-batch --fresh-copy-after 2
run fresh_after3 $B --fresh-batch --fresh-copy-after 3
