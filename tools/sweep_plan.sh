#!/bin/bash
# Sweeps of the convolution planner's constants in the FULL step (diagnostic build, tools/build_diag.sh): interleaved bench.py runs.
# usage: tools/sweep_plan.sh <rounds> VAR=v1,v2,... [VAR2=...]   (each variable swept alone against the defaults)
LIB=tools/_build/libmi355_unet_diag.so
R=$1; shift
run() { python bench.py --steps 100 --no-cpu-baseline --no-probe --lib $LIB 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'],3))"; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --lib $LIB > /dev/null 2>&1
for ((r = 0; r < R; r++)); do
  echo "round $r defaults $(run) ms"
  for spec in "$@"; do
    var=${spec%%=*}; vals=${spec#*=}
    for v in ${vals//,/ }; do export $var=$v; echo "round $r $var=$v $(run) ms"; unset $var; done
  done
done
