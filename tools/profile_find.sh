#!/bin/bash
# rocprofv3 passes over the configs[1] bench (k_find): kernel trace + FETCH/WRITE, at 100 k and 16 M lookups per launch
# usage: tools/profile_find.sh <tag>     (run on the GPU box from the repo root; summaries: tools/profile_summary.py <tag>_c2a / <tag>_c2b)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
set -e
for V in "a 100000" "b 16000000"; do
  set -- $V
  T=${TAG}_c2$1
  B="python3 $R/bench.py --workload c2 --lookups $2 --steps 20 --warmup 3 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_kt -o kt -- $B > $R/gpurun_out/prof_${T}_kt_bench.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${T}_fetch -o f -- $B > $R/gpurun_out/prof_${T}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${T}_write -o w -- $B > $R/gpurun_out/prof_${T}_write.log 2>&1
  python3 $R/tools/profile_summary.py $T
  echo "$T done"
done
