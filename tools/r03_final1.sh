#!/bin/bash
# final pass 1 (GPU box): the whole -m gpu suite with durations, smoke(), the driver's bench command
python3 -m pytest tests -q -m gpu --durations=15 > gpurun_out/r03_final_gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_final_gpu_tests.log
tail -22 gpurun_out/r03_final_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final_smoke.log 2>&1; tail -2 gpurun_out/r03_final_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_final_bench_c3.log 2>&1; tail -c 300 gpurun_out/r03_final_bench_c3.log
