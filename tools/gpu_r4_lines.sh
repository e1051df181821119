#!/bin/bash
# usage: tools/gpu_r4_lines.sh <tag>: the round's extra bench lines (160^3 bf16 / fp8, the reference's 8 x 64^3 batch, forced collectives)
tag=$1
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 "$@" > gpurun_out/${tag}_bench_$name.json 2> gpurun_out/${tag}_bench_$name.err; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${tag}_bench_$name.json") if l.startswith("{")][-1]); r = d.get("roofline", {})
    print("$name", round(d["ms_per_step"], 3), round(d["value"], 2), r.get("kernel"), r.get("frac"))
except Exception as e: print("$name", "ERR", e); print(open("gpurun_out/${tag}_bench_$name.err").read()[-600:])
PY
}
run bf16_160 python bench.py --size 160 --steps 100 --no-cpu-baseline
run fp8_160 python bench.py --size 160 --steps 100 --dtype fp8 --no-cpu-baseline
run bf16_8x64 python bench.py --size 64 --batch 8 --steps 100 --no-cpu-baseline
run bf16_force_collectives python bench.py --force-collectives --no-cpu-baseline
