#!/bin/bash
# One gpurun call = tests + kernel micro-benchmarks + step bench (a box costs minutes to acquire: batch the work).
# usage: tools/gpu_round.sh <tag> [pytest-args...]
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/${tag}_pytest.log
tail -5 gpurun_out/${tag}_pytest.log
timeout -k 10 300 python tools/bench_kernels.py conv > gpurun_out/${tag}_kernels.log 2>&1
echo "kernels rc=$?"
if [ -f tools/_build/libmi355_unet_diag.so ]; then
  MI355_CONV_SHAPE=9 timeout -k 10 300 python tools/bench_kernels.py conv --lib tools/_build/libmi355_unet_diag.so > gpurun_out/${tag}_kernels_shape9.log 2>&1
  echo "kernels(shape 9) rc=$?"
fi
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
cat gpurun_out/${tag}_kernels.log
