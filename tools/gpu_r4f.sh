#!/bin/bash
tag=${1:-r4n}
mkdir -p gpurun_out
bash tools/ab_flags.sh 3 "--lib tools/_build/libmi355_unet_diag.so" "--lib tools/_build/libmi355_unet_l64.so" "--lib tools/_build/libmi355_unet_l128.so" 2>&1 | tee gpurun_out/${tag}_ab.txt
