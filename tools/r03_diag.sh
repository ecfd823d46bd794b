#!/bin/bash
# walk-kernel timers (diag build) + the shipped build's bench line (run on the GPU box from the repo root)
set -e
O=gpurun_out/r03_diag.log
: > $O
echo "== diag build, LDBG_WG_TIMES" >> $O
LDBG_DIAG_LIB=1 LDBG_WG_TIMES=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "wavefront|workgroups|^\{" | tail -12 >> $O
echo "== shipped build" >> $O
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O 2>&1
echo done >> $O
