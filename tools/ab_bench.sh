#!/bin/bash
# Interleaved A/B of bench.py configurations inside ONE gpurun call (boxes differ by several % and drift while
# warming up): tools/ab_bench.sh <rounds> "<env A>" "<env B>" ...   e.g.  tools/ab_bench.sh 4 "MI355_CONV_CT=2" ""
R=$1; shift
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > /dev/null 2>&1     # warm the box up
for ((r = 0; r < R; r++)); do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))")
    echo "round $r [$cfg] $v ms"
  done
done
