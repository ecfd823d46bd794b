#!/bin/bash
# the sharded regime's round parameters (GPU box, one RCCL rank): row slots per request x requests per owner and round
for cd in 32 64 128 256; do
  for rpo in 65536 131072; do
    python3 bench.py --sharded --steps 2 --warmup 1 --no-cpu-baseline --chain-depth $cd --rows-per-owner $rpo > gpurun_out/r03_sh_${cd}_${rpo}.log 2>&1
    python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_sh_${cd}_${rpo}.log").read().strip().split("\n")[-1])
    print("chain $cd rows_per_owner $rpo: %.3f G k-mers/s, %.1f ms/step, %d rounds, %.2f ms/round, image rows %d" % (d["value"]/1e9, d["ms_per_step"], d["config"]["rounds_per_step"], d["config"]["ms_per_round"], d["config"]["image_rows_used"]))
except Exception as ex:
    print("chain $cd rows_per_owner $rpo: failed", ex)
PY
  done
done
