#!/bin/bash
# usage: tools/gpu_round3.sh <tag> : tests, kernel micro-benchmarks, step bench (bf16 128, fp8 / bf16 160), per-launch trace
tag=$1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/${tag}_pytest.log
tail -4 gpurun_out/${tag}_pytest.log
timeout -k 10 300 python tools/bench_kernels.py conv > gpurun_out/${tag}_kernels.log 2>&1
timeout -k 10 300 python tools/bench_kernels.py wgrad >> gpurun_out/${tag}_kernels.log 2>&1
cat gpurun_out/${tag}_kernels.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --size 160 --steps 50 > gpurun_out/${tag}_bench_bf16_160.json 2> gpurun_out/${tag}_bench_bf16_160.err
timeout -k 10 300 python bench.py --no-cpu-baseline --size 160 --steps 50 --dtype fp8 > gpurun_out/${tag}_bench_fp8_160.json 2> gpurun_out/${tag}_bench_fp8_160.err
python - <<PY
import json
for f in ("bench", "bench_bf16_160", "bench_fp8_160"):
    try:
        d = json.loads([l for l in open("gpurun_out/${tag}_%s.json" % f) if l.startswith("{")][-1])
        r = d.get("roofline", {})
        print(f, round(d["ms_per_step"], 3), round(d["value"], 2), r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
bash tools/gpu_profile.sh ${tag} > gpurun_out/${tag}_profile.log 2>&1
head -12 gpurun_out/${tag}_prof_summary.txt
