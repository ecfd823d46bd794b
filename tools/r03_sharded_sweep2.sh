#!/bin/bash
# the sharded regime's round length: row slots per request x loop iterations a wavefront runs before it ends its round (LDBG_IMG_YIELD)
for cfg in "32 32" "128 128" "256 256" "512 512" "1024 1024" "256 512" "1024 4096"; do
  set -- $cfg
  LDBG_IMG_YIELD=$2 python3 bench.py --sharded --steps 2 --warmup 1 --no-cpu-baseline --chain-depth $1 --rows-per-owner 65536 > gpurun_out/r03_sh2_$1_$2.log 2>&1
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r03_sh2_$1_$2.log").read().strip().split("\n") if l.startswith("{")][-1])
    print("row slots $1, yield $2: %.3f G k-mers/s, %.1f ms/step, %d rounds, %.2f ms/round, image rows %d, parity %s" % (d["value"]/1e9, d["ms_per_step"], d["config"]["rounds_per_step"], d["config"]["ms_per_round"], d["config"]["image_rows_used"], d.get("parity")))
except Exception as ex:
    print("row slots $1, yield $2: failed", ex); print(open("gpurun_out/r03_sh2_$1_$2.log").read()[-500:])
PY
done
