#!/bin/bash
# final pass 3 (GPU box): the table at 1 Gb (k = 47, 3 colours, ~1.05e9 records) on the final build; then the GPU seed sweep for what time is left
python3 bench.py --genome-len 1000000000 --repeat-families 170000 --steps 5 --warmup 2 --cpu-seconds 15 > gpurun_out/r03_final_bench_c3_1Gb.log 2>&1; tail -c 400 gpurun_out/r03_final_bench_c3_1Gb.log; echo
rm -f /tmp/ldbg_bench/c3_L1000000000*
echo pass3 done
