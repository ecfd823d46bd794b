"""Parity gate at BASELINE.json's full size (configs[2]: 23.3 Mb, 3 colours, k = 47, child links, 50,000 seeds): the oracle
cannot walk all of it in test time, so the full batch is held to size-independent properties and a RANDOM sample of it to
the oracle bit for bit."""
import hashlib
import os
import time

import numpy as np
import pytest
import torch  # noqa: F401  (before libldbg)

pytestmark = pytest.mark.gpu

K, L, NSEEDS = 47, 23332839, 50000


@pytest.fixture(scope="module")
def workload():
    from tools import synth
    d = os.environ.get("LDBG_BENCH_DIR", "/tmp/ldbg_bench")
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "c3_L%d_k%d_s%d_r0" % (L, K, NSEEDS))     # the files bench.py uses
    if not (os.path.exists(prefix + ".ctx") and os.path.exists(prefix + ".json")):
        import json
        t = time.time()
        st = synth.generate(prefix, L, K, colours=3, with_links=True, seed=0xC0FFEE03, n_chrom=14, n_repeat_families=4000,
                            repeat_copies=4, repeat_len=(50, 300), n_seeds=NSEEDS, threads=min(16, os.cpu_count() or 1))
        st["gen_seconds"] = round(time.time() - t, 1)
        json.dump(st, open(prefix + ".json", "w"))
    return prefix


@pytest.mark.timeout(900)
def test_full_size_walks(orc, workload):
    import corticall_amd as ca
    from corticall_amd import BOTH, OR, ContigStopper, CortexGraph, CortexLinks, TraversalEngineFactory
    assert ca.default_lib().device_count() >= 1
    g = CortexGraph(workload + ".ctx")
    links = CortexLinks(workload + ".ctp.gz", g)
    e = (TraversalEngineFactory().traversalColors(g.getColorForSampleName("child")).traversalDirection(BOTH).combinationOperator(OR)
         .stoppingRule(ContigStopper).graph(g).links(links).make())
    seeds = np.fromfile(workload + ".seeds", dtype=np.uint8).reshape(-1, K)
    arena, offs, wl = e.walk_batch_arrays(seeds)
    trav = e.kmers_traversed
    digest = hashlib.sha256(arena.tobytes()).hexdigest()
    # (1) idempotence: the same batch again gives the same bytes and the same amount of work
    arena2, offs2, wl2 = e.walk_batch_arrays(seeds)
    assert hashlib.sha256(arena2.tobytes()).hexdigest() == digest and (offs2 == offs).all() and (wl2 == wl).all()
    assert e.kmers_traversed == trav
    # (2) shape: a non-empty contig has walk_len + k - 1 bases, contains its seed, and is over ACGT
    lens = offs[1:] - offs[:-1]
    assert ((lens == 0) | (lens == wl + K - 1)).all()
    assert set(np.unique(arena)) <= set(b"ACGT")
    raw = arena.tobytes()
    rng = np.random.default_rng(12345)
    sample = rng.choice(len(seeds), 3000, replace=False)
    for i in sample:
        c = raw[offs[i]:offs[i + 1]]
        assert not c or seeds[i].tobytes() in c
    # (3) every k-mer of a sampled contig is a record of the graph (the last one may be a neighbour without a record)
    for i in sample[:300]:
        c = np.frombuffer(raw[offs[i]:offs[i + 1]], dtype=np.uint8)
        if len(c) < K:
            continue
        win = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(c, K))
        if len(win) > 4000:
            win = win[rng.choice(len(win), 4000, replace=False)]
        idx, _, _ = g.find_batch(win, with_payload=False)
        assert (idx < 0).sum() <= 2
    # (4) bit-exact against the oracle on random seeds (tuned search: same results as the faithful mode, checked in
    #     tests/test_oracle_golden.py, but fast enough for long walks), on several host threads: as many as 30 s allow, never fewer than 40
    from tests.fullsize_util import oracle_sample

    def make():
        return orc.Engine(orc.Graph(workload + ".ctx", tuned=True), [0], links=[orc.Links(workload + ".ctp.gz")], stopper="ContigStopper")

    def check(oe, i):
        exp, nv = oe.walk(seeds[i].tobytes().decode())
        assert raw[offs[i]:offs[i + 1]].decode() == exp and wl[i] == nv, i
    assert oracle_sample(make, [int(i) for i in sample], check, 30, 40) >= 40
    g.close()


# ---------------------------------------------------------------------------------------------------- configs[1]
@pytest.fixture(scope="module")
def workload_c2():
    """configs[1]: 10 Mb uniform genome, k = 31, ONE colour — the files bench.py --workload c2 uses"""
    from tools import synth
    d = os.environ.get("LDBG_BENCH_DIR", "/tmp/ldbg_bench")
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "c2_L%d_k%d" % (10_000_000, 31))
    if not os.path.exists(prefix + ".ctx"):
        synth.generate(prefix, 10_000_000, 31, colours=1, with_links=False, seed=0xC0FFEE01, n_chrom=1, n_repeat_families=0, repeat_copies=0,
                       n_indels=0, n_dnm=0, n_tandem=0, n_seeds=100000, threads=min(16, os.cpu_count() or 1))
    return prefix


@pytest.mark.timeout(900)
def test_full_size_lookups_and_walks_c2(orc, workload_c2):
    """100,000 random-access lookups (50 % present in a random orientation, 50 % random k-mers) and 10,000 ContigStopper walks
    on the 10 Mb single-colour k = 31 table: membership properties on all of them, a random sample bit-exact against the oracle
    in its FAITHFUL mode (ASCII 3-point search, CortexGraph.java:272-317)."""
    import corticall_amd as ca
    from corticall_amd import BOTH, OR, ContigStopper, CortexGraph, TraversalEngineFactory
    k = 31
    g = CortexGraph(workload_c2 + ".ctx")
    assert g.getNumColors() == 1 and g.getKmerSize() == k and 9_000_000 < g.getNumRecords() < 10_000_001
    present = np.fromfile(workload_c2 + ".seeds", dtype=np.uint8).reshape(-1, k)
    rng = np.random.default_rng(0xC0FFEE02)
    n = 100_000
    q = np.empty((n, k), dtype=np.uint8)
    half = n // 2
    q[:half] = present[rng.integers(0, len(present), half)]
    flip = rng.random(half) < 0.5                        # random orientation: reverse complement half of the present ones
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    q[:half][flip] = comp[q[:half][flip][:, ::-1]]
    q[half:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n - half, k))]
    is_present = np.zeros(n, dtype=bool)
    is_present[:half] = True
    perm = rng.permutation(n)
    q, is_present = q[perm], is_present[perm]
    idx, cov, edges = g.find_batch(q)
    # (1) membership: every planted k-mer is found, (2) a hit's record IS the canonical form of the query, (3) idempotence
    assert (idx[is_present] >= 0).all()
    assert (idx[~is_present] >= 0).sum() <= 2            # a uniformly random 31-mer is in a 10^7-record table with p ~ 2e-12
    hits = np.nonzero(idx >= 0)[0]
    rc = comp[q[:, ::-1]]
    canon = np.where((rc.view("S%d" % k).ravel() < q.view("S%d" % k).ravel())[:, None], rc, q)
    sample_hits = hits[rng.choice(len(hits), 2000, replace=False)]
    for i in sample_hits:
        r = g.getRecord(int(idx[i]))
        assert r.getKmerAsString().encode() == canon[i].tobytes()
        assert list(r.getCoverages()) == [int(cov[i][0])] and r.getEdges()[0] == int(edges[i][0])
    idx2, cov2, edges2 = g.find_batch(q)
    assert (idx2 == idx).all() and (cov2 == cov).all() and (edges2 == edges).all()
    # (4) records are strictly ascending, so lookups of consecutive records return consecutive indices
    some = rng.integers(0, g.getNumRecords() - 64, 50)
    for s0 in some[:10]:
        ks = [g.getRecord(int(s0) + j).getKmerAsString() for j in range(8)]
        assert ks == sorted(ks) and list(g.find_batch(ks)[0]) == list(range(int(s0), int(s0) + 8))
    # (5) the oracle in faithful mode on a random sample
    og = orc.Graph(workload_c2 + ".ctx", use_cache=True, tuned=False)
    t0, checked = time.time(), 0
    for i in rng.choice(n, 4000, replace=False):
        if time.time() - t0 > (25 if checked >= 500 else 400):
            break
        exp, ecov, eed = og.find(q[i].tobytes().decode())
        assert exp == int(idx[i]), (q[i].tobytes(), exp, int(idx[i]))
        if exp >= 0:
            assert ecov[0] == int(cov[i][0]) and eed[0] == int(edges[i][0])
        checked += 1
    assert checked >= 500
    # ---- simple walks: 10,000 ContigStopper walks (no links) from uniform records
    e = (TraversalEngineFactory().traversalColors(0).traversalDirection(BOTH).combinationOperator(OR).stoppingRule(ContigStopper).graph(g).make())
    seeds = present[rng.integers(0, len(present), 10_000)]
    arena, offs, wl = e.walk_batch_arrays(seeds)
    raw = arena.tobytes()
    lens = offs[1:] - offs[:-1]
    assert ((lens == 0) | (lens == wl + k - 1)).all() and (lens > 0).all()
    assert set(np.unique(arena)) <= set(b"ACGT")
    for i in rng.choice(len(seeds), 2000, replace=False):
        assert seeds[i].tobytes() in raw[offs[i]:offs[i + 1]]
    arena2, offs2, _ = e.walk_batch_arrays(seeds)
    assert hashlib.sha256(arena2.tobytes()).hexdigest() == hashlib.sha256(raw).hexdigest() and (offs2 == offs).all()
    from tests.fullsize_util import oracle_sample

    def make():
        return orc.Engine(orc.Graph(workload_c2 + ".ctx", tuned=True), [0], stopper="ContigStopper")

    def check(oe, i):
        exp, nv = oe.walk(seeds[i].tobytes().decode())
        assert raw[offs[i]:offs[i + 1]].decode() == exp and wl[i] == nv, i
    assert oracle_sample(make, [int(i) for i in rng.choice(len(seeds), 2000, replace=False)], check, 20, 40) >= 40
    assert e.kmers_traversed > 0
    g.close()


# ---------------------------------------------------------------------------------------------------- configs[3], one GPU
@pytest.mark.timeout(1100)
def test_full_size_dfs_c4(orc, workload):
    """configs[3] on one GPU: 50,000 DestinationStopper searches (FORWARD, Call.java:759-779) on the 23.3 Mb 3-colour k = 47 LdBG
    with child links, each towards the child k-mer 200-2,000 bp downstream on the seed's own contig.  Whole batch: idempotence
    and shape; a random sample of the graphs: vertices and edges in insertion order against the oracle."""
    import corticall_amd as ca
    from corticall_amd import BOTH, FORWARD, OR, ContigStopper, CortexGraph, CortexLinks, DestinationStopper, TraversalEngineFactory
    g = CortexGraph(workload + ".ctx")
    links = CortexLinks(workload + ".ctp.gz", g)
    seeds = np.fromfile(workload + ".seeds", dtype=np.uint8).reshape(-1, K)
    child = g.getColorForSampleName("child")
    we = (TraversalEngineFactory().traversalColors(child).traversalDirection(BOTH).combinationOperator(OR).stoppingRule(ContigStopper)
          .graph(g).links(links).make())
    arena, offs, wl = we.walk_batch_arrays(seeds)
    rng = np.random.default_rng(0xC0FFEE05)
    sink = np.empty_like(seeds)
    for i in range(len(seeds)):
        c = arena[offs[i]:offs[i + 1]]
        p = c.tobytes().find(seeds[i].tobytes()) if len(c) >= K else -1
        if p < 0:
            sink[i] = seeds[(i + 1) % len(seeds)]
            continue
        dd = int(rng.integers(200, 2001))
        qq = min(len(c) - K, p + dd)
        sink[i] = c[qq:qq + K]
    sink_off = np.arange(len(seeds) + 1, dtype=np.int64)
    sink_buf = np.ascontiguousarray(sink).reshape(-1)
    src = np.ascontiguousarray(seeds).reshape(-1)
    e = (TraversalEngineFactory().traversalColors(child).traversalDirection(FORWARD).combinationOperator(OR).stoppingRule(DestinationStopper)
         .graph(g).links(links).make())
    n = len(seeds)

    def sizes(b):
        out = np.zeros((n, 3), dtype=np.int64)
        for i in range(n):
            gi = b.graph(i)
            if gi is not None:
                out[i] = (1, gi.nv, gi.ne)
        return out
    b1 = e.dfs_batch_arrays(src, n, sink_buf, sink_off)
    t1 = e.dfs_kmers_traversed
    s1 = sizes(b1)
    b2 = e.dfs_batch_arrays(src, n, sink_buf, sink_off)
    assert e.dfs_kmers_traversed == t1 and (sizes(b2) == s1).all()           # idempotence
    assert s1[:, 0].sum() > 0.5 * n                                         # most sinks are reached
    found = np.nonzero(s1[:, 0])[0]
    assert (s1[found, 2] >= s1[found, 1] - 1).all()                          # a connected graph: |E| >= |V| - 1
    sample = rng.choice(n, 4000, replace=False)
    for i in sample[:300]:                                                   # the source is vertex 0 and the sink is a vertex
        gi = b1.graph(int(i))
        if gi is None:
            continue
        vt = gi.vertex_tuples()
        assert vt[0][0] == seeds[i].tobytes().decode() and any(v[0] == sink[i].tobytes().decode() for v in vt)
        assert gi.vertex_tuples() == b2.graph(int(i)).vertex_tuples() and gi.edge_tuples() == b2.graph(int(i)).edge_tuples()
    from tests.fullsize_util import oracle_sample
    mine = {int(i): b1.graph(int(i)) for i in sample[:1500]}        # (graph handles are made on this thread; the checks only read them)
    tuples = {i: (None if gi is None else (gi.vertex_tuples(), gi.edge_tuples())) for i, gi in mine.items()}

    def make():
        return orc.Engine(orc.Graph(workload + ".ctx", tuned=True), [0], links=[orc.Links(workload + ".ctp.gz")], stopper="DestinationStopper", direction=orc.FORWARD)

    def check(oe, i):
        r = oe.dfs(seeds[i].tobytes().decode(), [sink[i].tobytes().decode()])
        try:
            assert (tuples[i] is None) == r.is_null, i
            if tuples[i] is not None:
                assert tuples[i][0] == r.vertices() and tuples[i][1] == r.edges(), i
        finally:
            r.free()
    assert oracle_sample(make, list(tuples), check, 30, 40) >= 40
    g.close()
