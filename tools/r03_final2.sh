#!/bin/bash
# final pass 2 (GPU box): rocprofv3 summaries of the default bench and of the k_find bench, then the other workloads' lines
R=$(pwd)
bash tools/profile_walk.sh r03 > gpurun_out/r03_profile_walk.log 2>&1; tail -2 gpurun_out/r03_profile_walk.log
bash tools/profile_find.sh r03 > gpurun_out/r03_profile_find.log 2>&1; tail -2 gpurun_out/r03_profile_find.log
cd $R
mkdir -p gpurun_out/profiles_r03; cp profiles/r03_* gpurun_out/profiles_r03/ 2>/dev/null
cp gpurun_out/prof_r03_kt_bench.log gpurun_out/profiles_r03/r03_bench_under_rocprof.log 2>/dev/null
python3 bench.py --workload c4 --steps 5 --warmup 2 > gpurun_out/r03_final_bench_c4.log 2>&1; tail -c 200 gpurun_out/r03_final_bench_c4.log; echo
python3 bench.py --workload c4 --stopper ExplorationStopper --max-len 300 --steps 5 --warmup 2 > gpurun_out/r03_final_bench_c4_exploration.log 2>&1
python3 bench.py --workload c2 --steps 20 --warmup 3 > gpurun_out/r03_final_bench_c2.log 2>&1; tail -c 200 gpurun_out/r03_final_bench_c2.log; echo
python3 bench.py --workload c2 --lookups 16000000 --steps 10 --warmup 3 --cpu-seconds 5 > gpurun_out/r03_final_bench_c2_16M.log 2>&1
echo pass2 done
