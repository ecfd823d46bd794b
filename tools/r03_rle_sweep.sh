#!/bin/bash
# k_contigs_rle launch geometry sweep (+ the 8-lane groups build's walk timers)
O=gpurun_out/r03_rlesweep.log
: > $O
run() { echo "== $*" >> $O; env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('ms/step %.2f k_walk %.2f contig %.3f ms (%.2f)'%(j['ms_per_step'], r['avg_launch_ms'], r['contig_kernel']['ms_per_step'], r['contig_kernel']['frac']))
" >> $O; }
run LDBG_RLE_BLOCK=64 LDBG_RLE_GRID=16384
run LDBG_RLE_BLOCK=64 LDBG_RLE_GRID=65536
run LDBG_RLE_BLOCK=64 LDBG_RLE_GRID=32768
run LDBG_RLE_BLOCK=128 LDBG_RLE_GRID=8192
run LDBG_RLE_BLOCK=128 LDBG_RLE_GRID=32768
run LDBG_RLE_BLOCK=256 LDBG_RLE_GRID=4096
run LDBG_RLE_BLOCK=256 LDBG_RLE_GRID=16384
run LDBG_RLE_BLOCK=256 LDBG_RLE_GRID=2048
echo "== diag" >> $O
LDBG_DIAG_LIB=1 LDBG_WG_TIMES=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "wavefront" | tail -6 | cut -c1-330 >> $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_walks or dense_cycles or run_steps or big_link_stores or long_walks or ref_ or dfs_dense or dfs_run_steps" 2>&1 | tail -2 >> $O
echo done >> $O
