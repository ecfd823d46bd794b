#!/bin/bash
# the whole -m gpu suite with durations, then the default bench line (run on the GPU box from the repo root)
T=${1:-full}
python3 -m pytest tests -q -m gpu --durations=30 -x > gpurun_out/r03_${T}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_${T}_gpu_tests.log
tail -45 gpurun_out/r03_${T}_gpu_tests.log
python3 bench.py > gpurun_out/r03_${T}_bench.log 2>&1; tail -c 400 gpurun_out/r03_${T}_bench.log
