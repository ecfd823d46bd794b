#!/usr/bin/env python3
"""More seeds of the randomised parity cases than the test suite runs (on the GPU box: python tools/soak_gpu.py [first_seed] [count])."""
import pathlib
import sys
import tempfile
import time
import traceback

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import corticall_amd as ca  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from tests import parity_cases as pc  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
orc.build(); orc.lib()
lib = ca.default_lib()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    import random
    r = random.Random(seed)
    k = r.choice([9, 15, 21, 31, 32, 47])
    for name, fn in (("dfs_run_steps", lambda t: pc.case_dfs_run_steps(orc, lib, t, seed)), ("run_steps", lambda t: pc.case_run_steps(orc, lib, t, seed)),
                     ("dfs_dense", lambda t: pc.case_dfs_dense(orc, lib, t, seed)), ("dense_cycles", lambda t: pc.case_dense_cycles(orc, lib, t, seed)),
                     ("findtips", lambda t: pc.case_findtips(orc, lib, t, k, seed, seed % 2 == 0)),
                     ("partition", lambda t: pc.case_partition(orc, lib, t, r.choice([21, 31, 47]), seed, seed % 2 == 1)),
                     ("dfs_rules", lambda t: pc.case_dfs_rules(orc, lib, t, r.choice([21, 31]), seed, seed % 2 == 0)),      # (k = 9: some rules fork without end in the checker too)
                     ("facade", lambda t: pc.case_facade(orc, lib, t, r.choice([21, 31, 47]), seed, seed % 2 == 0)),
                     ("random_walks", lambda t: pc.case_random_walks(orc, lib, t, r.choice([21, 31, 47, 64]), seed, seed % 2 == 0))):
        tmp = pathlib.Path(tempfile.mkdtemp(prefix="soak_"))
        try:
            fn(tmp)
            print("ok", name, seed, "%.0f s" % (time.time() - t0), flush=True)
        except Exception:
            bad += 1
            print("FAILED", name, seed, flush=True)
            traceback.print_exc()
print("soak done:", bad, "failures")
sys.exit(1 if bad else 0)
