#!/bin/bash
# feasibility: two independent bench processes on ONE GPU at the same time (how much of a launch's tail can another batch fill?)
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03_solo.log 2>&1
python3 bench.py --steps 400 --warmup 3 --no-cpu-baseline > gpurun_out/r03_two_a.log 2>&1 &
PA=$!
python3 bench.py --steps 400 --warmup 3 --no-cpu-baseline > gpurun_out/r03_two_b.log 2>&1 &
PB=$!
wait $PA; wait $PB
python3 - <<'PY'
import json
for f in ("solo", "two_a", "two_b"):
    d = json.loads(open("gpurun_out/r03_%s.log" % f).read().strip().split("\n")[-1])
    print(f, "step %.3f ms" % d["ms_per_step"], "k_walk %.3f ms" % d["roofline"]["avg_launch_ms"], "rle %.3f" % d["roofline"]["contig_kernel"]["ms_per_step"])
PY
