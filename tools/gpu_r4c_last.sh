#!/bin/bash
# usage: tools/gpu_r4c_last.sh <tag>: after the planner change (conv_api.hip: lowg rows up to 20): full suite, PMC traffic records for the
# new source hash (bf16 128^3, fp8 160^3), the bench lines
tag=$1
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && { grep -E "^E  |FAILED" gpurun_out/${tag}_pytest.log | head -20; exit 1; }
bash tools/pmc_traffic.sh $tag 128 bf16 > gpurun_out/${tag}_pmc_traffic.log 2>&1; echo "traffic rc=$?"; tail -2 gpurun_out/${tag}_pmc_traffic.log
cp profiles/${tag}_bf16_128_traffic.json profiles/${tag}_bf16_128_pmc_step_traffic.txt gpurun_out/ 2>/dev/null
bash tools/pmc_traffic.sh $tag 160 fp8 > gpurun_out/${tag}_pmc_traffic_fp8_160.log 2>&1; echo "traffic fp8 160 rc=$?"; tail -2 gpurun_out/${tag}_pmc_traffic_fp8_160.log
cp profiles/${tag}_fp8_160_traffic.json profiles/${tag}_fp8_160_pmc_step_traffic.txt gpurun_out/ 2>/dev/null
bash tools/gpu_artifacts.sh $tag b
[ -f tools/_build/libmi355_unet_diag_cur.so ] && { SIZE=160 bash tools/ab_flags.sh 2 "--size 160 --lib tools/_build/libmi355_unet_diag_cur.so" "--size 160" > gpurun_out/${tag}_ab_160_prev.txt 2>&1; cat gpurun_out/${tag}_ab_160_prev.txt; }
