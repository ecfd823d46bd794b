#!/usr/bin/env python3
"""Summaries of the rocprofv3 passes of tools/profile_walk.sh, written into profiles/ (tracked):
   profiles/<tag>_kernel_stats.csv   per-kernel calls / total / average / min / max duration (from the kernel trace)
   profiles/<tag>_pmc.log            per-dispatch counter values of the walk-path kernels
   profiles/<tag>_walk_traffic.json  HBM bytes per k_walk launch (FETCH_SIZE + WRITE_SIZE), stamped with the library build"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out_dir = os.path.join(ROOT, "profiles")
WALK_KERNELS = ("k_walk", "k_expand_paths", "k_contigs", "k_zero16", "k_walk_lengths", "k_dfs", "k_path_lengths", "k_find")


def rows(pattern):
    for fn in glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True):
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def short(name):
    m = re.search(r"ldbg::(\w+)", name)
    return m.group(1) if m else name.split("(")[0][:40]


# ---- kernel trace -> stats
dur = {}
for r in rows("prof_%s_kt/**/*kernel_trace.csv" % tag):
    try:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    except (KeyError, ValueError):
        continue
    dur.setdefault(r.get("Kernel_Name", "?"), []).append(d)
tot = sum(sum(v) for v in dur.values()) or 1.0
with open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w") as f:
    f.write("Name,Calls,TotalDurationMs,AverageMs,MinMs,MaxMs,Percentage\n")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%.4f,%.4f,%.4f,%.4f,%.2f\n' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))

# ---- counters
pmc = {}
for sub in ("fetch", "write", "sq1", "sq2", "sq3"):
    for r in rows("prof_%s_%s/**/*counter_collection.csv" % (tag, sub)):
        k = short(r.get("Kernel_Name", "?"))
        if not any(k.startswith(w) for w in WALK_KERNELS):
            continue
        pmc.setdefault((k, r.get("Counter_Name", "?")), []).append(float(r.get("Counter_Value", "0")))
with open(os.path.join(out_dir, "%s_pmc.log" % tag), "w") as f:
    f.write("# tools/profile_walk.sh %s: rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --only-timed (c2 passes: tools/profile_find.sh) ; one pass per counter set\n" % tag)
    f.write("# kernel counter dispatches mean_per_dispatch (FETCH_SIZE / WRITE_SIZE in KB)\n")
    for (k, c), v in sorted(pmc.items()):
        f.write("%s %s %d %.3f\n" % (k, c, len(v), sum(v) / len(v)))


def mean(k, c):
    v = pmc.get((k, c))
    return sum(v) / len(v) if v else None


lib = None
for line in open(os.path.join(ROOT, "gpurun_out", "prof_%s_kt_bench.log" % tag), errors="replace") if os.path.exists(os.path.join(ROOT, "gpurun_out", "prof_%s_kt_bench.log" % tag)) else []:
    if line.startswith("{") and '"library"' in line:
        lib = json.loads(line).get("library")
fk, wk = mean("k_walk", "FETCH_SIZE"), mean("k_walk", "WRITE_SIZE")
if fk is not None and wk is not None:
    json.dump({
        "library": lib, "kernel": "k_walk", "read_bytes_per_launch": int(fk * 1024), "write_bytes_per_launch": int(wk * 1024),
        "hbm_bytes_per_launch": int((fk + wk) * 1024),
        "other_kernels_bytes_per_launch": {k: int(((mean(k, "FETCH_SIZE") or 0) + (mean(k, "WRITE_SIZE") or 0)) * 1024) for k in ("k_expand_paths", "k_contigs", "k_zero16")},
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/%s_pmc.log)" % tag,
        "caveat": "MI355X_MICROARCH.md, HBM: FETCH_SIZE under-reports wide coalesced streams by 2x (k_expand_paths / k_contigs read such streams: "
                  "double their FETCH_SIZE share before comparing); k_walk issues narrow random accesses, reported uncorrected",
    }, open(os.path.join(out_dir, "%s_walk_traffic.json" % tag), "w"), indent=1)
print("profiles written for", tag, "kernels:", len(dur), "counter series:", len(pmc))
