#!/bin/bash
# Interleaved A/B of planner knobs (diagnostic build) in the 1x24xSIZE^3 step (SIZE env, default 160): tools/ab_env_160.sh <rounds> <lib> "<env A>" "<env B>" ...
R=$1; LIB=$2; shift; shift
python bench.py --size ${SIZE:-160} --steps 20 --warmup 5 --no-cpu-baseline --no-probe --lib $LIB > /dev/null 2>&1
for ((r = 0; r < R; r++)); do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --size ${SIZE:-160} --steps 60 --no-cpu-baseline --no-probe --lib $LIB 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'],3))")
    echo "round $r [$cfg] $v ms"
  done
done
