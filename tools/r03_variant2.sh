#!/bin/bash
# link-store elements per lane in LDS (16 shipped; 14 -> 7 workgroups per CU, 12 -> 8) with two engines in flight
for v in variant14 variant; do
  LDBG_DIAG_LIB=$v python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --in-flight 2 > gpurun_out/r03_v2_$v.log 2>&1
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_v2_$v.log").read().strip().split("\n")[-1])
    print("$v", d["library"], ": step %.3f ms = %.1f G k-mers/s; k_walk %.3f ms per launch (events); single batch %.3f ms, k_walk %.3f; wavefronts %d" % (d["ms_per_step"], d["value"]/1e9, d["roofline"]["avg_launch_ms"], d["single_batch"]["ms_per_step"], d["single_batch"]["k_walk_ms"], d["roofline"]["latency_bound"]["wavefronts"]))
except Exception as ex:
    print("$v failed", ex); print(open("gpurun_out/r03_v2_$v.log").read()[-800:])
PY
done
