"""Parity gates of the HASH-SHARDED regime at workload size (one RCCL rank: the GPU box has one GPU; two and three ranks run on gloo
in tests/test_distributed.py).
 (a) configs[2]'s files, 50,000 seeds: ShardedTraversalEngine.walk_batch — link-guided ContigStopper walks over the local image
 (b) configs[3] as BASELINE.json names it: DestinationStopper dfs over the hash-sharded table, 50,000 sources
 (c) the scaled stand-in for configs[4]: a k = 63, 3-colour graph as large as the gate's time allows, link-guided sharded walks
The whole batch is held to the resident engine (itself gated against the oracle in tests/test_gpu_fullsize.py) and to
size-independent properties, a RANDOM sample to the oracle bit for bit (TraversalEngine.java:241-279, 356-482)."""
import hashlib
import os
import time

import numpy as np
import pytest
import torch  # noqa: F401  (before libldbg)

pytestmark = pytest.mark.gpu

K, L, NSEEDS = 47, 23332839, 50000
K5, L5 = 63, int(os.environ.get("LDBG_C5_SCALED_LEN", "70000000"))


def _bench_dir():
    d = os.environ.get("LDBG_BENCH_DIR", "/tmp/ldbg_bench")
    os.makedirs(d, exist_ok=True)
    return d


@pytest.fixture(scope="module")
def workload():
    """configs[2]'s files (the ones bench.py and tests/test_gpu_fullsize.py use)"""
    import json
    from tools import synth
    prefix = os.path.join(_bench_dir(), "c3_L%d_k%d_s%d_r0" % (L, K, NSEEDS))
    if not (os.path.exists(prefix + ".ctx") and os.path.exists(prefix + ".json")):
        st = synth.generate(prefix, L, K, colours=3, with_links=True, seed=0xC0FFEE03, n_chrom=14, n_repeat_families=4000,
                            repeat_copies=4, repeat_len=(50, 300), n_seeds=NSEEDS, threads=min(16, os.cpu_count() or 1))
        json.dump(st, open(prefix + ".json", "w"))
    return prefix


@pytest.fixture(scope="module")
def workload_c5s():
    """configs[4] scaled to one GPU's test budget: k = 63, 3 colours, child links, 50,000 seeds"""
    from tools import synth
    import json
    import shutil
    prefix = os.path.join(_bench_dir(), "c5s_L%d_k%d_s%d" % (L5, K5, NSEEDS))
    if not (os.path.exists(prefix + ".ctx") and os.path.exists(prefix + ".json")):
        need = int(L5 * 1.05) * (16 + 15) + (1 << 28)            # the graph file: one 31-byte record per k-mer
        free = shutil.disk_usage(_bench_dir()).free
        assert free > need, "not enough room for the scaled configs[4] graph in %s: %d MB free, %d MB needed" % (_bench_dir(), free >> 20, need >> 20)
        st = synth.generate(prefix, L5, K5, colours=3, with_links=True, seed=0xC0FFEE05, n_chrom=8, n_repeat_families=4000 * max(1, L5 // L),
                            repeat_copies=4, repeat_len=(50, 300), n_seeds=NSEEDS, threads=min(16, os.cpu_count() or 1))
        size = os.path.getsize(prefix + ".ctx")
        assert size > st["n_records"] * 31 and (size - st["n_records"] * 31) < 4096, "the generator wrote %d bytes for %d records" % (size, st["n_records"])
        json.dump(st, open(prefix + ".json", "w"))
    return prefix


@pytest.fixture()
def rccl_one_rank():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


from tests.fullsize_util import oracle_sample  # noqa: E402


def _resident(prefix, direction, stopper, max_len=75000):
    from corticall_amd import OR, CortexGraph, CortexLinks, TraversalEngineFactory
    g = CortexGraph(prefix + ".ctx")
    links = CortexLinks(prefix + ".ctp.gz", g)
    e = (TraversalEngineFactory().traversalColors(g.getColorForSampleName("child")).traversalDirection(direction).combinationOperator(OR)
         .stoppingRule(stopper).maxBranchLength(max_len).graph(g).links(links).make())
    return g, links, e


def _sharded_walk_gate(orc, prefix, k, chain_depth=32, oracle_budget=35):
    from corticall_amd import BOTH, ContigStopper, CortexLinks
    from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine
    seeds = np.fromfile(prefix + ".seeds", dtype=np.uint8).reshape(-1, k)
    # the resident engine on the whole table: the reference point for EVERY contig of the batch
    g, _, e = _resident(prefix, BOTH, ContigStopper)
    arena, offs, wl = e.walk_batch_arrays(seeds)
    trav = e.kmers_traversed
    raw = arena.tobytes()
    del arena
    e.close()
    g.close()
    sg = ShardedCortexGraph(prefix + ".ctx", device=0)
    sg.build_neighbour_index()
    links = [CortexLinks(prefix + ".ctp.gz", sg.shard)]
    se = ShardedTraversalEngine(sg, [0], links=links, rows_per_owner=65536, check_every=16, chain_depth=chain_depth)
    mine = [s.tobytes().decode() for s in seeds]
    got = se.walk_batch(mine)
    assert se.rounds > 0 and se.kmers_traversed == trav
    # (1) every contig of the batch equals the resident engine's; walk lengths too
    assert len(got) == len(seeds)
    digest = hashlib.sha256()
    for i, c in enumerate(got):
        assert c.encode() == raw[offs[i]:offs[i + 1]], i
        digest.update(c.encode())
    assert (np.asarray(se.walk_lengths) == wl).all()
    # (2) idempotence from an empty image
    again = se.walk_batch(mine)
    d2 = hashlib.sha256()
    for c in again:
        d2.update(c.encode())
    assert d2.hexdigest() == digest.hexdigest() and se.kmers_traversed == trav
    # (3) shape
    lens = np.array([len(c) for c in got])
    assert ((lens == 0) | (lens == wl + k - 1)).all()
    rng = np.random.default_rng(4242 + k)
    sample = [int(i) for i in rng.choice(len(seeds), 4000, replace=False)]
    for i in sample[:2000]:
        assert not got[i] or mine[i] in got[i]
    # (4) a random sample bit for bit against the oracle

    def make():
        return orc.Engine(orc.Graph(prefix + ".ctx", tuned=True), [0], links=[orc.Links(prefix + ".ctp.gz")], stopper="ContigStopper")

    def check(oe, i):
        exp, nv = oe.walk(mine[i])
        assert got[i] == exp and wl[i] == nv, i
    checked = oracle_sample(make, sample, check, oracle_budget, 24)
    se.close()
    sg.close()
    return checked


@pytest.mark.timeout(600)
def test_sharded_walks_c3_one_rank(orc, workload, rccl_one_rank):
    """(a) configs[2] over the hash-sharded table: 50,000 link-guided walks, image empty at the start of the batch"""
    assert _sharded_walk_gate(orc, workload, K) >= 24


@pytest.mark.timeout(600)
def test_sharded_dfs_c4_one_rank(orc, workload, rccl_one_rank):
    """(b) configs[3] as BASELINE names it: 50,000 DestinationStopper searches (FORWARD, Call.java:759-779) towards the child k-mer
    200-2,000 bp downstream on the source's own contig, over the hash-sharded table"""
    from corticall_amd import BOTH, FORWARD, ContigStopper, CortexLinks, DestinationStopper
    from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine
    seeds = np.fromfile(workload + ".seeds", dtype=np.uint8).reshape(-1, K)
    n = len(seeds)
    g, links, we = _resident(workload, BOTH, ContigStopper)
    arena, offs, _ = we.walk_batch_arrays(seeds)
    rng = np.random.default_rng(0xC0FFEE05)
    sink = np.empty_like(seeds)
    for i in range(n):
        c = arena[offs[i]:offs[i + 1]]
        p = c.tobytes().find(seeds[i].tobytes()) if len(c) >= K else -1
        if p < 0:
            sink[i] = seeds[(i + 1) % n]
            continue
        q = min(len(c) - K, p + int(rng.integers(200, 2001)))
        sink[i] = c[q:q + K]
    we.close()
    # resident searches: sizes of every graph of the batch
    from corticall_amd import OR, TraversalEngineFactory
    re_ = (TraversalEngineFactory().traversalColors(g.getColorForSampleName("child")).traversalDirection(FORWARD).combinationOperator(OR)
           .stoppingRule(DestinationStopper).graph(g).links(links).make())
    rb = re_.dfs_batch_arrays(np.ascontiguousarray(seeds).reshape(-1), n, np.ascontiguousarray(sink).reshape(-1), np.arange(n + 1, dtype=np.int64))
    r_trav = re_.dfs_kmers_traversed
    r_sizes = np.zeros((n, 3), dtype=np.int64)
    for i in range(n):
        gi = rb.graph(i)
        if gi is not None:
            r_sizes[i] = (1, gi.nv, gi.ne)
    sources = [s.tobytes().decode() for s in seeds]
    sinks = [[s.tobytes().decode()] for s in sink]
    sg = ShardedCortexGraph(workload + ".ctx", device=0)
    sg.build_neighbour_index()
    se = ShardedTraversalEngine(sg, [0], links=[CortexLinks(workload + ".ctp.gz", sg.shard)], direction=FORWARD, stopping_rule=DestinationStopper,
                                rows_per_owner=65536, check_every=16, chain_depth=32)
    graphs = se.dfs_batch(sources, sinks)
    assert se.rounds > 0 and se.dfs_kmers_traversed == r_trav
    # (1) every search of the batch: reached or not, vertices, edges — as on the resident table
    s_sizes = np.zeros((n, 3), dtype=np.int64)
    for i, gi in enumerate(graphs):
        if gi is not None:
            s_sizes[i] = (1, gi.nv, gi.ne)
    assert (s_sizes == r_sizes).all()
    assert s_sizes[:, 0].sum() > 0.5 * n
    sample = [int(i) for i in rng.choice(n, 4000, replace=False)]
    # (2) a sample of graphs vertex by vertex and edge by edge (insertion order) against the resident engine's
    norm = lambda gi: [(km, rec >= 0, ci, ix) for km, rec, ci, ix in gi.vertex_tuples()]
    for i in sample[:400]:
        if graphs[i] is None:
            continue
        assert norm(graphs[i]) == norm(rb.graph(i)) and graphs[i].edge_tuples() == rb.graph(i).edge_tuples()
        assert graphs[i].vertex_tuples()[0][0] == sources[i]
    # (3) a random sample against the oracle

    def make():
        return orc.Engine(orc.Graph(workload + ".ctx", tuned=True), [0], links=[orc.Links(workload + ".ctp.gz")], stopper="DestinationStopper", direction=orc.FORWARD)

    def check(oe, i):
        r = oe.dfs(sources[i], sinks[i])
        try:
            assert (graphs[i] is None) == r.is_null, i
            if graphs[i] is not None:
                assert norm(graphs[i]) == [(km, rec >= 0, ci, ix) for km, rec, ci, ix in r.vertices()] and graphs[i].edge_tuples() == r.edges(), i
        finally:
            r.free()
    assert oracle_sample(make, sample, check, 35, 24) >= 24
    se.close()
    sg.close()
    re_.close()
    g.close()


@pytest.mark.timeout(900)
def test_sharded_link_walks_k63_scaled_c5(orc, workload_c5s, rccl_one_rank):
    """(c) configs[4] scaled: k = 63 (two words, the last one full), 3 colours, child links, 50,000 link-guided walks over the sharded table"""
    assert _sharded_walk_gate(orc, workload_c5s, K5) >= 24
