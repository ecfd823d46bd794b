#!/bin/bash
# final pass 2b (GPU box): the rocprofv3 passes of the default bench again with --only-timed (kernel averages of the trace = the timed region's)
R=$(pwd)
bash tools/profile_walk.sh r03 > gpurun_out/r03_profile_walk.log 2>&1; tail -2 gpurun_out/r03_profile_walk.log
cd $R
mkdir -p gpurun_out/profiles_r03; cp profiles/r03_kernel_stats.csv profiles/r03_pmc.log profiles/r03_walk_traffic.json gpurun_out/profiles_r03/ 2>/dev/null
cp gpurun_out/prof_r03_kt_bench.log gpurun_out/profiles_r03/r03_bench_under_rocprof.log 2>/dev/null
echo pass2b done
