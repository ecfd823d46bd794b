#!/bin/bash
# optimisation iteration: walk parity subset, diag timers, bench with / without the second run phase
set -e
T=${1:-iter}
O=gpurun_out/r03_$T.log
: > $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_walks or dense_cycles or run_steps or big_link_stores or long_walks or ref_ or hash_collision or partition" > gpurun_out/r03_${T}_tests.log 2>&1 || { tail -30 gpurun_out/r03_${T}_tests.log; exit 1; }
tail -2 gpurun_out/r03_${T}_tests.log >> $O
echo "== diag build, LDBG_WG_TIMES" >> $O
LDBG_DIAG_LIB=1 LDBG_WG_TIMES=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "wavefront|workgroups" | tail -7 | cut -c1-330 >> $O
echo "== shipped build" >> $O
python3 bench.py --steps 10 --warmup 3 --cpu-seconds 6 >> $O 2>&1
echo '== c4' >> $O
python3 bench.py --workload c4 --steps 3 --warmup 1 --cpu-seconds 6 >> $O 2>&1
echo done >> $O
