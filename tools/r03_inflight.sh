#!/bin/bash
# engines in flight (GPU box): the default bench line at 1, 2, 3, 4 engines
for n in 2 1 3 4; do
  python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --in-flight $n > gpurun_out/r03_inflight_$n.log 2>&1
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_inflight_$n.log").read().strip().split("\n")[-1])
    print("in flight $n: step %.3f ms = %.1f G k-mers/s; k_walk %.3f ms per launch (events), rle %.3f; single batch %.3f ms, k_walk %.3f" % (d["ms_per_step"], d["value"]/1e9, d["roofline"]["avg_launch_ms"], d["roofline"]["contig_kernel"]["ms_per_step"], d["single_batch"]["ms_per_step"], d["single_batch"]["k_walk_ms"]))
except Exception as ex:
    print("in flight $n failed", ex); print(open("gpurun_out/r03_inflight_$n.log").read()[-1500:])
PY
done
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r03c_gpu_parity.log 2>&1; tail -3 gpurun_out/r03c_gpu_parity.log
