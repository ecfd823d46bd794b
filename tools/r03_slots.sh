#!/bin/bash
# resident workgroups per CU of one k_walk launch x engines in flight: fewer slots per launch = lanes refill from the strand queue and two
# launches are co-resident from the start
for cfg in "2 6" "2 3" "2 4" "3 2" "4 2" "1 3" "3 4"; do
  set -- $cfg
  LDBG_WG_PER_CU=$2 python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --in-flight $1 > gpurun_out/r03_slots_$1_$2.log 2>&1
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_slots_$1_$2.log").read().strip().split("\n")[-1])
    print("in flight $1, workgroups per CU $2: step %.3f ms = %.1f G k-mers/s; k_walk %.3f ms per launch (events); single batch %.3f ms; wavefronts %d" % (d["ms_per_step"], d["value"]/1e9, d["roofline"]["avg_launch_ms"], d["single_batch"]["ms_per_step"], d["roofline"]["latency_bound"]["wavefronts"]))
except Exception as ex:
    print("$1 $2 failed", ex); print(open("gpurun_out/r03_slots_$1_$2.log").read()[-600:])
PY
done
