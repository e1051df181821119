#!/bin/bash
# Interleaved A/B of bench.py FLAG sets inside one gpurun call: tools/ab_flags.sh <rounds> "<flags A>" "<flags B>" ...
R=$1; shift
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > /dev/null 2>&1     # warm the box up
for ((r = 0; r < R; r++)); do
  for cfg in "$@"; do
    v=$(python bench.py --steps 100 --no-cpu-baseline --no-probe $cfg 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'],3))")
    echo "round $r [$cfg] $v ms"
  done
done
