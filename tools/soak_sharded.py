#!/usr/bin/env python3
"""Randomised walks and searches over a sharded table's image on one rank (RCCL), against the CPU checker: varying k, request
capacity, row slots per request and graph (python tools/soak_sharded.py [first_seed] [count], on the GPU box)."""
import os
import pathlib
import random
import sys
import tempfile
import traceback

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch.distributed as dist  # noqa: E402

import corticall_amd as ca  # noqa: E402
from corticall_amd import CortexLinks  # noqa: E402
from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from tests import parity_cases as pc  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
orc.build(); orc.lib()
lib = ca.default_lib()
bad = 0
for seed in range(first, first + count):
    rng = random.Random(9000 + seed)
    k = rng.choice([21, 31, 32, 47, 63])
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="soaksh_"))
    try:
        base = pc.genome_with_repeats(rng, rng.randint(1500, 5000), n_rep=6, rep_len=(k // 2 + 1, 3 * k), copies=(2, 4))
        kid = pc.mutate(rng, base, snv=0.01, indel=0.003)
        p = str(tmp / "g.ctx")
        orc.build_graph(p, [("kid", [kid]), ("mom", [base])], k)
        og = orc.Graph(p, tuned=True)
        rl = max(3 * k, 60)
        lp = str(tmp / "kid.ctp.gz")
        orc.build_links(og, lp, "kid", [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), max(1, rl // 4))] + [kid[-rl:]])
        ol = orc.Links(lp)
        sg = ShardedCortexGraph(p, lib=lib)
        links = [CortexLinks(lp, sg.shard, lib=lib)]
        kmers = [og.record_string(i).split()[0] for i in range(og.N)]
        seeds = rng.sample(kmers, min(300, len(kmers)))
        seeds = [s if rng.random() < 0.5 else orc.revcomp(s) for s in seeds] + [pc.rand_seq(rng, k), "N" * k]
        cap, depth, ML = rng.choice([64, 256, 4096]), rng.choice([1, 4, 16, 64]), rng.choice([600, 3000, 75000])
        for trav, direction, op, with_links in (([0], 0, 0, True), ([0, 1], 2, 0, rng.random() < 0.5), ([1], 1, 1, False)):
            oe = orc.Engine(og, trav, links=[ol] if with_links else [], op_and=(op == 1), direction=direction, stopper="ContigStopper", max_length=ML)
            it0 = oe.kmers_traversed()
            exp = [oe.walk(s)[0] for s in seeds]
            e = ShardedTraversalEngine(sg, trav, links=links if with_links else (), direction=direction, op=op, max_branch_length=ML, rows_per_owner=cap,
                                       chain_depth=depth, check_every=rng.choice([1, 4, 16]))
            got = e.walk_batch(seeds)
            assert got == exp, ("walks", k, trav, cap, depth)
            assert e.kmers_traversed == oe.kmers_traversed() - it0
            e.close()
        # searches towards a sink downstream
        pos = rng.sample(range(0, len(kid) - k - 300), 40)
        sources = [kid[q:q + k] for q in pos]
        sinks = [[kid[q + d:q + d + k]] for q, d in ((q, rng.randint(20, 280)) for q in pos)]
        for stopper, direction in (("DestinationStopper", 1), ("ExplorationStopper", 0)):
            oe = orc.Engine(og, [0], links=[ol], direction=direction, max_length=300, stopper=stopper)
            it0 = oe.kmers_traversed()
            e = ShardedTraversalEngine(sg, [0], links=links, direction=direction, max_branch_length=300, stopping_rule=stopper, rows_per_owner=cap,
                                       chain_depth=depth, check_every=4)
            got = e.dfs_batch(sources, sinks)
            for s_, sk, gi in zip(sources, sinks, got):
                r = oe.dfs(s_, sk)
                assert (gi is None) == r.is_null, (stopper, s_)
                if gi is not None:
                    assert [(km, rec >= 0, ci, ix) for km, rec, ci, ix in gi.vertex_tuples()] == [(km, rec >= 0, ci, ix) for km, rec, ci, ix in r.vertices()]
                    assert gi.edge_tuples() == r.edges()
                r.free()
            assert e.dfs_kmers_traversed == oe.kmers_traversed() - it0
            e.close()
        sg.close()
        print("ok", seed, "k", k, "cap", cap, "depth", depth, "maxLength", ML, flush=True)
    except Exception:
        bad += 1
        print("FAILED", seed, flush=True)
        traceback.print_exc()
print("soak done:", bad, "failures")
dist.destroy_process_group()
sys.exit(1 if bad else 0)
