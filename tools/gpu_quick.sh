#!/bin/bash
# usage: tools/gpu_quick.sh <tag> [pytest -k expr]: selected tests (or all), step bench
tag=$1; shift
mkdir -p gpurun_out
if [ -n "$1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "$1" > gpurun_out/${tag}_pytest.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1
fi
echo "pytest rc=$?"; grep -E "^E  |passed|failed|FAILED|Error" gpurun_out/${tag}_pytest.log | head -30
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${tag}_bench.json") if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("bench", round(d["ms_per_step"], 3), round(d["value"], 2), r.get("kernel"), r.get("frac"))
    for f in r.get("conv_families", [])[:4]: print("   ", f["plan"], f["launches_per_step"], round(f["ms_per_step"], 3), round(f["tflops"]))
    print("   hbm", {k: (round(v["achieved"]), round(v["total_ms"], 3)) for k, v in d.get("roofline_hbm", {}).get("kernels", {}).items()})
except Exception as e:
    print("bench ERR", e); print(open("gpurun_out/${tag}_bench.err").read()[-1500:])
PY
