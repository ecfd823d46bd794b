#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box from the repo root): a kernel trace with stats, then one PMC pass per
# counter set (counters never share a run with the other trace domains).  CSV output under gpurun_out/prof_<tag>_*/; the summaries
# that are kept are written by tools/profile_summary.py into profiles/.
# usage: tools/profile_walk.sh <tag> [extra bench.py arguments]
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
set -e
# --only-timed: the process's kernel trace is then the timed region (engines in flight) + two warm-up batches per engine, so that the
# per-kernel averages of the trace are those of the region bench.py's own HIP events time
B="python3 $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --only-timed $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_kt -o kt -- $B > $R/gpurun_out/prof_${TAG}_kt_bench.log 2>&1
echo kt done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o f -- $B > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o w -- $B > $R/gpurun_out/prof_${TAG}_write.log 2>&1
echo write done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_sq1 -o s1 -- $B > $R/gpurun_out/prof_${TAG}_sq1.log 2>&1
echo sq1 done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_sq2 -o s2 -- $B > $R/gpurun_out/prof_${TAG}_sq2.log 2>&1
echo sq2 done
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_sq3 -o s3 -- $B > $R/gpurun_out/prof_${TAG}_sq3.log 2>&1
echo sq3 done
python3 $R/tools/profile_summary.py $TAG
