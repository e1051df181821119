#!/bin/bash
tag=${1:-r4l}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "conv_fwd_bwd or deconv or golden or sinks or deferred or hipgraph_replayed or fp8" > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|FAILED|Error" gpurun_out/${tag}_pytest.log | head -20
bash tools/ab_flags.sh 3 "" "--dgrad-first" 2>&1 | tee gpurun_out/${tag}_ab.txt
