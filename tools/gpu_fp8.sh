#!/bin/bash
# fp8 round: the fp8 tests, then 160^3 bench lines bf16 / fp8 interleaved
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fp8.py -x -q > gpurun_out/fp8_tests.log 2>&1; rc=$?; tail -15 gpurun_out/fp8_tests.log
[ $rc -ne 0 ] && exit $rc
for r in 0 1; do
  for dt in bf16 fp8; do
    timeout -k 10 400 python bench.py --size 160 --steps 60 --dtype $dt --no-cpu-baseline > gpurun_out/fp8r_${dt}_$r.json 2> gpurun_out/fp8r_${dt}_$r.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/fp8r_${dt}_$r.json") if l.startswith("{")][-1])
    print("$dt round $r", round(d["ms_per_step"], 3), "ms", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
except Exception as e:
    print("$dt", "ERR", e); print(open("gpurun_out/fp8r_${dt}_$r.err").read()[-1500:])
PY
  done
done
