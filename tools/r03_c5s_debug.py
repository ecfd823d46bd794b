"""where does the 70 Mb k = 63 graph go wrong? (run on the GPU box)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth
from corticall_amd.distributed import ctx_header
d = "/tmp/ldbg_bench"; os.makedirs(d, exist_ok=True)
prefix = os.path.join(d, "c5s_L70000000_k63_s50000")
if not os.path.exists(prefix + ".ctx"):
    t = time.time()
    st = synth.generate(prefix, 70_000_000, 63, colours=3, with_links=True, seed=0xC0FFEE05, n_chrom=8, n_repeat_families=12000, repeat_copies=4, repeat_len=(50, 300), n_seeds=50000, threads=16)
    print("generated in", time.time() - t, st, flush=True)
raw = np.memmap(prefix + ".ctx", dtype=np.uint8, mode="r")
h = ctx_header(raw); rec = 8 * h["W"] + 5 * h["C"]
n = (raw.size - h["data_offset"]) // rec
r = raw[h["data_offset"]:h["data_offset"] + n * rec].reshape(n, rec)
keys = np.ascontiguousarray(r[:, :16]).view("<u8").reshape(-1, 2)
hi, lo = keys[:, 0], keys[:, 1]
bad = np.nonzero((hi[1:] < hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] <= lo[:-1])))[0]
print("file:", raw.size, "bytes,", n, "records; unsorted pairs", len(bad), bad[:5], "zero keys", int(((hi == 0) & (lo == 0)).sum()), flush=True)
import corticall_amd as ca
try:
    g = ca.CortexGraph(prefix + ".ctx")
    print("resident open ok:", g.getNumRecords(), flush=True)
    g.close()
except Exception as ex:
    print("resident open FAILED:", str(ex)[:200], flush=True)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
# the steps of ShardedCortexGraph.__init__ by hand
flat = torch.from_numpy(np.ascontiguousarray(r).reshape(-1)).cuda()
print("uploaded", flat.numel(), "bytes; nonzero tail:", int(flat[-1000:].ne(0).sum()), flush=True)
got = torch.empty_like(flat)
dist.all_to_all_single(got, flat, output_split_sizes=[flat.numel()], input_split_sizes=[flat.numel()])
torch.cuda.synchronize()
same = bool((got == flat).all().item())
print("all_to_all_single of", flat.numel(), "bytes identical:", same, flush=True)
if not same:
    diff = (got != flat).nonzero()
    print("first differing byte", int(diff[0]), "count", diff.numel(), flush=True)
back = got.cpu().numpy().reshape(-1, rec)
print("download identical:", bool((back == r).all()), flush=True)
image = np.concatenate([np.asarray(raw[:h["data_offset"]]), np.ascontiguousarray(back).reshape(-1)])
try:
    g = ca.CortexGraph(prefix + "#mem", image=image)
    print("open from memory ok:", g.getNumRecords(), flush=True)
except Exception as ex:
    print("open from memory FAILED:", str(ex)[:200], flush=True)
dist.destroy_process_group()
