#!/bin/bash
# tuning variant against the shipped build (GPU box): LDBG_DIAG_LIB=variant loads corticall_amd/_build_variant/libldbg.so
for cfg in "base" "variant" "variant6"; do
  case $cfg in
    base) E="";;
    variant) E="LDBG_DIAG_LIB=variant";;
    variant6) E="LDBG_DIAG_LIB=variant LDBG_WG_PER_CU=6";;
  esac
  env $E python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_variant_$cfg.log 2>&1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r03_variant_$cfg.log").read().strip().split("\n")[-1])
print("$cfg", d["library"], "step %.3f ms" % d["ms_per_step"], "k_walk %.3f ms" % d["roofline"]["avg_launch_ms"], "wavefronts", d["roofline"]["latency_bound"]["wavefronts"],
      "busiest", d["roofline"]["latency_bound"]["busiest_wavefront_iterations"], "us/iter %.2f" % d["roofline"]["latency_bound"]["us_per_iteration_of_the_busiest_wavefront"])
PY
done
