#!/bin/bash
# usage: tools/gpu_artifacts.sh <tag> [a|b]  -- the measurement artefacts of a round (copied from gpurun_out/ into profiles/ afterwards);
# part a = PMC traffic passes + kernel statistics + replay trace + PMC of the marching kernels, part b = the bench lines (a gpurun call is
# limited to 20 minutes: two calls); no part = both
tag=$1
part=${2:-ab}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
if [[ $part == *a* ]]; then
bash tools/pmc_traffic.sh $tag 128 bf16 > gpurun_out/${tag}_pmc_traffic.log 2>&1; echo "traffic rc=$?"; tail -3 gpurun_out/${tag}_pmc_traffic.log
cp profiles/${tag}_bf16_128_traffic.json profiles/${tag}_bf16_128_pmc_step_traffic.txt gpurun_out/ 2>/dev/null
bash tools/pmc_traffic.sh $tag 160 fp8 > gpurun_out/${tag}_pmc_traffic_fp8_160.log 2>&1; echo "traffic fp8 160 rc=$?"; tail -3 gpurun_out/${tag}_pmc_traffic_fp8_160.log
cp profiles/${tag}_fp8_160_traffic.json profiles/${tag}_fp8_160_pmc_step_traffic.txt gpurun_out/ 2>/dev/null
fi
run() { name=$1; shift; timeout -k 10 400 "$@" > gpurun_out/${tag}_bench_$name.json 2> gpurun_out/${tag}_bench_$name.err; python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${tag}_bench_$name.json") if l.startswith("{")][-1]); r = d.get("roofline", {})
    print("$name", round(d["ms_per_step"], 3), round(d["value"], 2), r.get("kernel"), r.get("frac"), r.get("traffic_over_algorithmic"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("$name", "ERR", e); print(open("gpurun_out/${tag}_bench_$name.err").read()[-600:])
PY
}
if [[ $part == *b* ]]; then
run bf16_default python bench.py
run bf16_gen_only python bench.py --workload gen_only --no-cpu-baseline
run bf16_160 python bench.py --size 160 --steps 100 --no-cpu-baseline
run fp8_160 python bench.py --size 160 --steps 100 --dtype fp8 --no-cpu-baseline
run bf16_8x64 python bench.py --size 64 --batch 8 --steps 100 --no-cpu-baseline
run bf16_fresh_batch python bench.py --fresh-batch --no-cpu-baseline
run bf16_force_collectives python bench.py --force-collectives --no-cpu-baseline
run bf16_2rank_gloo python bench.py --gpus 2 --backend gloo --steps 20 --no-cpu-baseline
fi
if [[ $part == *a* ]]; then
# kernel stats of the eager step
OUT=$ROOT/gpurun_out/${tag}_stats; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --steps 40 --warmup 5 --settle-s 0 --no-graph --no-cpu-baseline --no-probe > $OUT/run.log 2>&1 )
cp $OUT/p_kernel_stats.csv gpurun_out/${tag}_bf16_default_kernel_stats.csv
python3 tools/prof_summary.py $OUT 45 30 > gpurun_out/${tag}_bf16_default_summary.txt; head -9 gpurun_out/${tag}_bf16_default_summary.txt
rm -f $OUT/p_kernel_trace.csv
bash tools/gpu_profile_graph.sh ${tag} > gpurun_out/${tag}_graph_profile.txt 2>&1
bash tools/pmc_conv.sh -1 gpurun_out/${tag}_pmc_marchg "96->32" sq > /dev/null 2>&1; cp gpurun_out/${tag}_pmc_marchg/summary.txt gpurun_out/${tag}_pmc_conv_marchg_raw.txt
bash tools/pmc_conv.sh -1 gpurun_out/${tag}_pmc_march "32->32 @128" sq > /dev/null 2>&1; cp gpurun_out/${tag}_pmc_march/summary.txt gpurun_out/${tag}_pmc_conv_march_raw.txt
python3 - <<PY
import csv, glob
for k in ("marchg", "march"):
    try:
        rows = list(csv.DictReader(open(glob.glob("gpurun_out/${tag}_pmc_%s/SQ1/*kernel_trace.csv" % k)[0])))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "conv_march" in r["Kernel_Name"]]
        print(k, "kernel duration in the SQ1 pass: mean %.1f us over %d dispatches" % (sum(d) / len(d), len(d)))
    except Exception as e: print(k, "ERR", e)
PY
fi
