#!/bin/bash
# Timing-only variants of the C-ABI library (tools/_build/, never shipped): the other objects come from the diagnostic build
# (tools/build_diag.sh), conv_api.o is recompiled per variant.  usage: tools/build_variants.sh name1="-DX -DY" name2="-DZ" ...
set -e
cd "$(dirname "$0")/.."
[ -f tools/_build/runtime.o ] || bash tools/build_diag.sh
OTHERS=""
for f in runtime wgrad elementwise upcat dti patches metrics augment; do OTHERS="$OTHERS tools/_build/$f.o"; done
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMI355_DIAG $flags -c unet_bssfp_amd/csrc/conv_api.hip -o tools/_build/conv_api_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/libmi355_unet_$name.so tools/_build/conv_api_$name.o $OTHERS && echo built $name ) &
done
wait
