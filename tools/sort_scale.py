import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import corticall_amd as ca
from corticall_amd.distributed import ctx_header
from corticall_amd.partition import Sort
src = "/tmp/ldbg_bench/c3_L23332839_k47_s50000_r0.ctx"
raw = np.fromfile(src, dtype=np.uint8)
h = ctx_header(raw); rec = 8*h["W"] + 5*h["C"]
body = raw[h["data_offset"]:].reshape(-1, rec)
perm = np.random.default_rng(5).permutation(len(body))
np.concatenate([raw[:h["data_offset"]], body[perm].reshape(-1)]).tofile("/tmp/ldbg_bench/c3_shuffled.ctx")
ca.profile_reset()
t = time.time()
n = Sort("/tmp/ldbg_bench/c3_shuffled.ctx", "/tmp/ldbg_bench/c3_resorted.ctx").execute()
dt = time.time() - t
ms, _ = ca.profile_get("sort")
same = (np.fromfile("/tmp/ldbg_bench/c3_resorted.ctx", dtype=np.uint8) == raw).all()
print("records", n, "wall %.2f s" % dt, "device passes %.1f ms" % ms, "identical to the sorted original:", bool(same))
t = time.time()
words = np.ascontiguousarray(body[perm][:, :16]).view("<u8").reshape(-1, 2)
order = np.lexsort((words[:, 1], words[:, 0]))
print("numpy lexsort of the same keys (1 core): %.2f s" % (time.time() - t))
