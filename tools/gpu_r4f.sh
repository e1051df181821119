#!/bin/bash
tag=${1:-r04b}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_artifacts.sh $tag b
