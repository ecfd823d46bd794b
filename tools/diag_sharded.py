import os, sys, random, faulthandler, time
faulthandler.enable()
sys.path.insert(0,'/root/repo')
import numpy as np
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29612"
import torch.distributed as dist
import torch; dist.init_process_group("nccl", rank=0, world_size=1)
from oracle import pyoracle as orc
import corticall_amd as ca
from tools import synth
from corticall_amd import CortexLinks
from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine
lib=ca.default_lib()
L=int(sys.argv[1]); nseeds=int(sys.argv[2])
os.makedirs('/tmp/ldbg_bench', exist_ok=True)
prefix='/tmp/ldbg_bench/mid_L%d'%L
if not os.path.exists(prefix+'.ctx'):
    synth.generate(prefix, L, 47, colours=3, with_links=True, seed=0xC0FFEE03, n_chrom=2, n_repeat_families=L//6000, repeat_copies=4, repeat_len=(50,300), n_seeds=2000, threads=8)
seeds=np.fromfile(prefix+'.seeds',dtype=np.uint8).reshape(-1,47)
t0=time.time()
sg=ShardedCortexGraph(prefix+'.ctx',lib=lib); sg.build_neighbour_index()
print('shard ok %.1fs'%(time.time()-t0), flush=True)
links=CortexLinks(prefix+'.ctp.gz', sg.shard, lib=lib)
e=ShardedTraversalEngine(sg,[0],links=[links],max_branch_length=75000, rows_per_owner=int(sys.argv[3]) if len(sys.argv)>3 else 4096)
mine=[s.tobytes().decode() for s in seeds[:nseeds]]
t0=time.time()
got=e.walk_batch(mine)
print('walk %.1fs rounds %d traversed %d image rows %d'%(time.time()-t0, e.rounds, e.kmers_traversed, e.image_rows_used), flush=True)
og=orc.Graph(prefix+'.ctx',tuned=True); ol=orc.Links(prefix+'.ctp.gz')
oe=orc.Engine(og,[0],links=[ol],max_length=75000)
bad=0
for s,c in zip(mine,got):
    x=oe.walk(s)[0]
    if x!=c:
        bad+=1
        if bad<4:
            i=x.find(s); j=c.find(s)
            print('MISMATCH', len(x), len(c), 'seed at', i, j, 'left same' if x[:i][-min(i,j):]==c[:j][-min(i,j):] else 'left differs', 'right same' if x[i:i+min(len(x)-i,len(c)-j)]==c[j:j+min(len(x)-i,len(c)-j)] else 'right differs', 'left len', i, j, 'right len', len(x)-i, len(c)-j)
print('bad',bad,'of',len(mine),'oracle traversed',oe.kmers_traversed())
