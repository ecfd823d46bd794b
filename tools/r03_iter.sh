#!/bin/bash
# one optimisation iteration on the GPU box: link / walk parity subset, the diag build's timers, the shipped build's bench line
set -e
T=${1:-iter}
O=gpurun_out/r03_$T.log
: > $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_walks or dense_cycles or run_steps or big_link_stores or long_walks or ref_ or dfs_dense or hash_collision" > gpurun_out/r03_${T}_tests.log 2>&1 || { tail -30 gpurun_out/r03_${T}_tests.log; exit 1; }
tail -2 gpurun_out/r03_${T}_tests.log >> $O
echo "== diag build, LDBG_WG_TIMES" >> $O
LDBG_DIAG_LIB=1 LDBG_WG_TIMES=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "wavefront|workgroups" | tail -5 | cut -c1-330 >> $O
echo "== shipped build" >> $O
python3 bench.py --steps 10 --warmup 3 --cpu-seconds 6 >> $O 2>&1
echo done >> $O
if [ -n "$2" ]; then
  python3 -m pytest tests/test_gpu_sharded_fullsize.py -x -q -m gpu --durations=10 > gpurun_out/r03_${T}_gates.log 2>&1 || true
  tail -25 gpurun_out/r03_${T}_gates.log >> $O
fi
df -h /tmp /dev/shm . 2>/dev/null | tail -4 >> $O; free -g | head -2 >> $O; nproc >> $O
