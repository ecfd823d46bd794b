#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box from the repo root): kernel trace, then one PMC pass per counter set
# (counters never share a run with the other trace domains).  Outputs: gpurun_out/prof_*/ (sqlite), read by tools/profile_walk_summary.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_kt_bench.log 2>&1
echo kt done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1
echo write done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/prof_sq1 -o s1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_sq1.log 2>&1
echo sq1 done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/prof_sq2 -o s2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_sq2.log 2>&1
echo sq2 done
