#!/bin/bash
# usage: tools/gpu_r4h.sh <tag>: the full -m gpu suite, the step bench, a replay trace (third session of round 4)
tag=$1
bash tools/gpu_quick.sh $tag || exit 1
grep -q "failed\|error" gpurun_out/${tag}_pytest.log && exit 1
bash tools/gpu_profile_graph.sh $tag > gpurun_out/${tag}_graph_summary.txt 2>&1
grep -n "small\|^#" gpurun_out/${tag}_graph_step_trace.txt | head -30
