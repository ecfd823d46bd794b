#!/bin/bash
# round-3 knob sweep of the default bench (run on the GPU box from the repo root)
set -e
O=gpurun_out/r03_sweep.log
: > $O
run() { echo "== $*" >> $O; env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O 2>&1; }
run LDBG_X=base
run LDBG_WALK_BLOCK=32
run LDBG_WALK_BLOCK=16
run LDBG_WALK_BLOCK=32 LDBG_WG_PER_CU=8
run LDBG_WG_PER_CU=4
run LDBG_LEAN_RUN=1
run LDBG_LEAN_RUN=8
run LDBG_WG_TIMES=1 LDBG_HOST_TIMES=1
echo done >> $O
