#!/bin/bash
# final pass 4 (GPU box): the sharded regime on one RCCL rank (configs[2] walks, configs[3] searches), then the GPU seed sweep
python3 bench.py --sharded --steps 3 --warmup 1 --cpu-seconds 8 > gpurun_out/r03_final_bench_c3_sharded.log 2>&1; tail -c 200 gpurun_out/r03_final_bench_c3_sharded.log; echo
python3 bench.py --workload c4 --sharded --steps 2 --warmup 1 --cpu-seconds 8 > gpurun_out/r03_final_bench_c4_sharded.log 2>&1; tail -c 200 gpurun_out/r03_final_bench_c4_sharded.log; echo
timeout -k 10 420 python3 tools/soak_gpu.py 300 6 > gpurun_out/r03_final_soak_gpu.log 2>&1; tail -3 gpurun_out/r03_final_soak_gpu.log
echo pass4 done
