#!/bin/bash
tag=${1:-r4o}
mkdir -p gpurun_out
python bench.py --size 160 --dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-probe > /dev/null 2>&1
for r in 0 1 2; do
  for cfg in "--dtype bf16 --lib tools/_build/libmi355_unet_diag.so" "--dtype fp8 --lib tools/_build/libmi355_unet_diag.so" "--dtype fp8 --lib tools/_build/libmi355_unet_f8a.so"; do
    v=$(python bench.py --size 160 --steps 60 --no-cpu-baseline --no-probe $cfg 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'],3))")
    echo "round $r [$cfg] $v ms"
  done
done | tee gpurun_out/${tag}_ab.txt
