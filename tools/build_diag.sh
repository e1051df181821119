#!/bin/bash
# Diagnostic build of the C-ABI library (plan overrides via MI355_CONV_SHAPE / _CT / _KSPLIT): tools/_build/, never shipped.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMI355_DIAG $MI355_DIAG_FLAGS"
OBJS=""
for f in runtime conv_api wgrad elementwise upcat dti patches metrics augment; do
  /opt/rocm/bin/hipcc $FLAGS -c unet_bssfp_amd/csrc/$f.hip -o tools/_build/$f.o &
  OBJS="$OBJS tools/_build/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/libmi355_unet_diag${MI355_DIAG_SUFFIX}.so $OBJS
echo built tools/_build/libmi355_unet_diag${MI355_DIAG_SUFFIX}.so
