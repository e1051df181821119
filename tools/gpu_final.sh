#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_final.sh [tag] -- the full -m gpu suite, then the round artefacts (tools/gpu_artifacts.sh <tag>)
tag=${1:-r04}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_artifacts.sh $tag
