#!/bin/bash
# final call of the session: soak (bf16, fp8) + the bench lines of the final sources
tag=${1:-r04b}
mkdir -p gpurun_out
( timeout -k 10 200 python tools/soak.py --dtype bf16 --steps 1500 --every 250; timeout -k 10 200 python tools/soak.py --dtype fp8 --steps 1500 --every 250 ) > gpurun_out/${tag}_soak.txt 2>&1
echo "soak rc=$?"; tail -4 gpurun_out/${tag}_soak.txt
bash tools/gpu_artifacts.sh $tag b
