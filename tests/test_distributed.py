"""N > 1 paths on CPU: two gloo ranks, kernels through the TEST-ONLY host simulation (tests/hostsim).
 * hash-sharded table: every rank holds its shard; findRecord batches are routed with all-to-all exchanges and
   must answer exactly like the oracle on the whole graph
 * replicas: seeds partitioned over the ranks, contigs gathered — identical to a single-process run"""
import os
import random
import socket

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    import torch.distributed as dist
    if os.environ.get("LDBG_TEST_DUMP"):          # debugging aid: where is a rank that hangs?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LDBG_TEST_DUMP"]), exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _sharded_worker(rank, world, port, paths):
    dist = _init(rank, world, port)
    try:
        from corticall_amd.distributed import ShardedCortexGraph
        from oracle import pyoracle as orc
        from tests import hostsim
        lib = hostsim.load(rebuild=False)
        for path in paths:
            og = orc.Graph(path, tuned=True)
            sg = ShardedCortexGraph(path, lib=lib, chunk_records=1000)
            assert sg.getNumRecords() == og.N
            n_mine = sg.shard.getNumRecords()
            rng = random.Random(17 + rank)
            kmers = [og.record_string(i).split()[0] for i in rng.sample(range(og.N), min(og.N, 300))]
            qs = [k if rng.random() < 0.5 else orc.revcomp(k) for k in kmers]
            qs += ["".join(rng.choice("ACGT") for _ in range(og.k)) for _ in range(100)] + ["N" * og.k]
            if rank == 1:
                qs = qs[:57]                       # ragged batches; the exchange must cope
            found, cov, edges, owner, lidx = sg.find_batch(qs)
            for i, q in enumerate(qs):
                eidx, ecov, eedges = og.find(q)
                assert bool(found[i]) == (eidx >= 0), (path, q)
                if eidx >= 0:
                    assert ecov == [int(c) for c in cov[i]] and eedges == [int(x) for x in edges[i]], (q, ecov, cov[i])
                    assert 0 <= lidx[i] and 0 <= owner[i] < world
                else:
                    assert lidx[i] == -1 and not cov[i].any() and not edges[i].any()
            # an empty batch on one rank while the other asks
            found, *_ = sg.find_batch(qs[:5] if rank == 0 else [])
            assert len(found) == (5 if rank == 0 else 0)
            # the shards partition the table
            import torch
            t = torch.tensor([n_mine])
            dist.all_reduce(t)
            assert int(t.item()) == og.N and (og.N < 20 or 0 < n_mine < og.N)
            sg.close()
    finally:
        dist.destroy_process_group()


def _replica_worker(rank, world, port, ctx, ctp, seeds, expected):
    _init(rank, world, port)
    import torch.distributed as dist
    try:
        from corticall_amd import BOTH, OR, ContigStopper, CortexGraph, CortexLinks, TraversalEngineFactory
        from corticall_amd.distributed import gather_strings, partition
        from tests import hostsim
        lib = hostsim.load(rebuild=False)
        g = CortexGraph(ctx, lib=lib)
        e = (TraversalEngineFactory(lib=lib).traversalColors(0).traversalDirection(BOTH).combinationOperator(OR)
             .stoppingRule(ContigStopper).graph(g).links(CortexLinks(ctp, g)).maxBranchLength(300).make())
        first, cnt = partition(len(seeds), rank, world)
        contigs, _ = e.walk_batch(seeds[first:first + cnt]) if cnt else ([], None)
        everything = gather_strings(contigs)
        assert everything == expected
    finally:
        dist.destroy_process_group()


def _spawn(fn, args, world=2):
    import torch.multiprocessing as mp
    from tests import hostsim
    hostsim.build()          # once, here: the workers only load it
    mp.spawn(fn, args=(world, _free_port()) + tuple(args), nprocs=world, join=True)


def test_partition_covers_everything():
    from corticall_amd.distributed import partition
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            spans = [partition(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1


def test_pack_kmers_matches_library(orc):
    from corticall_amd.distributed import pack_kmers
    rng = random.Random(3)
    for k in (5, 31, 32, 33, 47, 64, 65):
        ks = ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(20)]
        a = np.frombuffer("".join(ks).encode(), dtype=np.uint8).reshape(-1, k)
        w = pack_kmers(a, k)
        for i, s in enumerate(ks):
            assert [int(x) for x in w[i]] == orc.encode_kmer(s)
    w, valid = pack_kmers(np.frombuffer(b"ACNGTACGTT", dtype=np.uint8).reshape(2, 5), 5, return_valid=True)
    assert list(valid) == [False, True] and int(w[0][0]) == 0                   # validity travels beside the words (Q4)


@pytest.mark.timeout(300)
def test_sharded_find_two_ranks(orc, tmp_path):
    from tests import parity_cases as pc
    rng = random.Random(11)
    paths = [os.path.join(GOLDEN, "two_short_contigs.ctx")]
    for k, ncol in ((21, 1), (47, 3)):
        haps = [("s%d" % c, [pc.rand_seq(rng, 1500)]) for c in range(ncol)]
        p = str(tmp_path / ("sh%d.ctx" % k))
        orc.build_graph(p, haps, k)
        paths.append(p)
    tiny = str(tmp_path / "tiny.ctx")                 # 2 records: Q1 — nothing is ever found
    orc.build_graph(tiny, [("a", ["ACGTA"])], 4)
    paths.append(tiny)
    _spawn(_sharded_worker, (paths,))


@pytest.mark.timeout(300)
def test_replicas_two_ranks(orc, tmp_path):
    from tests import hostsim, parity_cases as pc
    rng = random.Random(5)
    lib = hostsim.load()
    g1 = pc.genome_with_repeats(rng, 1200)
    cs = pc.Case(orc, tmp_path, lib, [("a", [g1])], 21, link_samples=["a"], name="rep")
    seeds = rng.sample(cs.all_kmers(), 41)
    oe, e = cs.engines(trav=[0], links=["a"], max_len=300)
    expected, _ = e.walk_batch(seeds)
    km = np.frombuffer("".join(seeds).encode(), dtype=np.uint8).reshape(len(seeds), 21)
    arena, offs, _ = oe.walk_batch(km)
    assert expected == [arena.tobytes()[offs[i]:offs[i + 1]].decode() for i in range(len(seeds))]
    _spawn(_replica_worker, (cs.path, str(tmp_path / "rep.a.ctp.gz"), seeds, expected))


def _sharded_walk_worker(rank, world, port, path, link_path, seeds, cfgs, expected, image_rows=None):
    dist = _init(rank, world, port)
    try:
        from corticall_amd import CortexLinks
        from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine, gather_strings, partition
        from tests import hostsim
        lib = hostsim.load(rebuild=False)
        sg = ShardedCortexGraph(path, lib=lib, chunk_records=700)
        sg.build_neighbour_index(chunk_records=300)
        links = CortexLinks(link_path, sg.shard, lib=lib) if link_path else None      # every rank: the link file against its own shard
        first, cnt = partition(len(seeds), rank, world)
        for ci, (trav, direction, op, max_len, with_links) in enumerate(cfgs):
            e = ShardedTraversalEngine(sg, trav, links=[links] if (with_links and links) else (), direction=direction, op=op,
                                       max_branch_length=max_len, rows_per_owner=64 if ci % 2 else 4096, check_every=4, image_rows=image_rows)
            for rep in range(2):                                # the second batch starts from an empty image again
                mine = e.walk_batch(seeds[first:first + cnt])
                got = gather_strings(mine)
                for i, s in enumerate(seeds):
                    assert got[i] == expected[ci][0][i], (cfgs[ci], s, got[i], expected[ci][0][i])
                import torch
                t = torch.tensor([e.kmers_traversed])
                dist.all_reduce(t)
                assert int(t.item()) == expected[ci][1], (int(t.item()), expected[ci][1])
                assert e.rounds > 0
            if image_rows:       # the image was too small for the batch: it has grown (on every rank together), nobody hung, results unchanged
                assert e.image_grown >= 1 and e.image_rows > image_rows
            e.close()
        sg.close()
    finally:
        dist.destroy_process_group()


def _sharded_walk_case(orc, tmp_path, k, with_links, world=2, image_rows=None, n_cfgs=None):
    from tests import parity_cases as pc
    rng = random.Random(100 + k + (7 if with_links else 0))
    base = pc.genome_with_repeats(rng, 900, n_rep=5, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
    kid = pc.mutate(rng, base, snv=0.01, indel=0.003)
    dad = pc.mutate(rng, base, snv=0.02, indel=0.003)
    path = str(tmp_path / "sw.ctx")
    orc.build_graph(path, [("kid", [kid]), ("mom", [base]), ("dad", [dad])], k)
    og = orc.Graph(path, tuned=True)
    link_path, ol = None, None
    if with_links:
        rl = max(3 * k, 60)
        reads = [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), max(1, rl // 4))] + [kid[-rl:]]
        link_path = str(tmp_path / "sw.kid.ctp.gz")
        orc.build_links(og, link_path, "kid", reads)
        ol = orc.Links(link_path)
    kmers = [og.record_string(i).split()[0] for i in range(og.N)]
    seeds = rng.sample(kmers, 60)
    seeds = [s if rng.random() < 0.5 else orc.revcomp(s) for s in seeds] + [pc.rand_seq(rng, k), "N" * k, kid[:k], kid[-k:]]
    ML = 400 if with_links else 75000       # (a link-guided walk circles a tandem repeat until maxLength)
    cfgs = [([0], 0, 0, ML, with_links), ([0], 1, 1, ML, with_links), ([1], 2, 0, ML, with_links), ([0, 2], 0, 0, ML, with_links), ([0], 0, 0, 9, with_links),
            ([0], 0, 0, ML, False)]
    expected = []
    for trav, direction, op, max_len, wl in cfgs:
        oe = orc.Engine(og, trav, links=[ol] if (wl and ol) else [], op_and=(op == 1), direction=direction, max_length=max_len, stopper="ContigStopper")
        it0 = oe.kmers_traversed()
        contigs = [oe.walk(s)[0] for s in seeds]
        expected.append((contigs, oe.kmers_traversed() - it0))
    if n_cfgs:
        cfgs, expected = cfgs[:n_cfgs], expected[:n_cfgs]
    _spawn(_sharded_walk_worker, (path, link_path, seeds, cfgs, expected, image_rows), world=world)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("k", [21, 32, 47])
def test_sharded_walks_two_ranks(orc, tmp_path, k):
    """ContigStopper walks over a table hash-sharded over two ranks == the oracle's walks on the whole graph: contigs and the
    number of k-mers traversed (odd and even k: palindromic k-mers, all-bits k-mers)"""
    _sharded_walk_case(orc, tmp_path, k, with_links=False)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("k", [21, 47, 64])
def test_sharded_link_walks_two_ranks(orc, tmp_path, k):
    """link-guided walks (TraversalEngine.java:241-279, 548-597: link store, junction choices, copies of revisited vertices, the
    walk that circles a repeat until maxLength) over the sharded table, rows fetched from the owning rank on demand"""
    _sharded_walk_case(orc, tmp_path, k, with_links=True)


@pytest.mark.timeout(900)
def test_sharded_link_walks_three_ranks(orc, tmp_path):
    _sharded_walk_case(orc, tmp_path, 31, with_links=True, world=3)


@pytest.mark.timeout(900)
def test_sharded_walks_image_overflow_two_ranks(orc, tmp_path):
    """an image that is too small for the batch: every rank sees the overflow flag in the same round (it travels in the all-reduced
    round statistics), leaves the rounds, doubles its image and walks the batch again — no rank waits for a row that can never come"""
    _sharded_walk_case(orc, tmp_path, 21, with_links=True, image_rows=40, n_cfgs=2)


def _sharded_dfs_worker(rank, world, port, path, link_path, sources, sinks, cfgs, expected, roi_path=None, use=None, image_rows=None, poison=None):
    dist = _init(rank, world, port)
    try:
        import corticall_amd as ca
        from corticall_amd import CortexLinks
        from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine, partition
        from tests import hostsim
        lib = hostsim.load(rebuild=False)
        sg = ShardedCortexGraph(path, lib=lib, chunk_records=700)
        links = CortexLinks(link_path, sg.shard, lib=lib) if link_path else None
        rois = ca.CortexGraph(roi_path, lib=lib) if roi_path else None
        all_sources, all_sinks = sources, sinks
        for ci, (stopper, trav, direction, max_len, with_links) in enumerate(cfgs):
            idx = use[ci] if use else list(range(len(all_sources)))
            sources, sinks = [all_sources[i] for i in idx], [all_sinks[i] for i in idx]
            first, cnt = partition(len(sources), rank, world)
            e = ShardedTraversalEngine(sg, trav, links=[links] if (with_links and links) else (), direction=direction, max_branch_length=max_len,
                                       stopping_rule=stopper, rows_per_owner=64 if ci % 2 else 2048, check_every=4,
                                       rois=rois if stopper.startswith(("Novel", "Nahr", "BubbleOpening")) else None,
                                       joining_colors=[1, 2] if stopper.startswith(("Novel", "Nahr", "BubbleOpening")) else (), image_rows=image_rows)
            if poison is not None:
                # ONE rank's batch holds a source the rule dereferences a missing record for (NullPointerException in the reference): that rank
                # raises it, every other rank raises PeerRankFailed — nobody is left waiting in a collective — and the next batch runs as usual
                from corticall_amd import _native
                from corticall_amd.distributed import PeerRankFailed
                bad_s = sources[first:first + cnt] + ([poison] if rank == 1 else [])
                bad_k = sinks[first:first + cnt] + ([[]] if rank == 1 else [])
                with pytest.raises(_native.JavaNullPointerException if rank == 1 else PeerRankFailed) as ei:
                    e.dfs_batch(bad_s, bad_k)
                assert "NullPointer" in str(ei.value) if rank == 1 else "peer rank failed" in str(ei.value), str(ei.value)
            got = e.dfs_batch(sources[first:first + cnt], sinks[first:first + cnt])
            for j, gi in enumerate(got):
                exp = expected[ci][first + j]
                if exp is None:
                    assert gi is None, (cfgs[ci], sources[first + j])
                    continue
                assert gi is not None, (cfgs[ci], sources[first + j])
                vt = [(km, rec >= 0, ci_, ix) for km, rec, ci_, ix in gi.vertex_tuples()]
                assert vt == exp[0] and gi.edge_tuples() == exp[1], (cfgs[ci], sources[first + j], len(vt), len(exp[0]))
                assert gi.walk_contig(sources[first + j], trav[0]) == exp[2]
            import torch
            t = torch.tensor([e.dfs_kmers_traversed])
            dist.all_reduce(t)
            assert int(t.item()) == expected[ci][-1], (cfgs[ci], int(t.item()), expected[ci][-1])
            assert e.rounds > 0
            if image_rows:
                assert e.image_grown >= 1 and e.image_rows > image_rows
            e.close()
        sg.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("k,mode", [(21, "plain"), (32, "plain"), (21, "overflow"), (21, "poison")])
def test_sharded_dfs_two_ranks(orc, tmp_path, k, mode):
    """dfs(source, sinks) with a stopping rule over the sharded table (TraversalEngine.java:64-106, 356-482): DestinationStopper towards a
    sink downstream (the gap-closing configuration, Call.java:759-779), ExplorationStopper, ContigStopper — graphs (vertices and edges in
    insertion order), toWalk/toContig of them, and the k-mers traversed against the oracle on the whole graph"""
    from tests import parity_cases as pc
    rng = random.Random(300 + k)
    base = pc.genome_with_repeats(rng, 900, n_rep=5, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
    kid = pc.mutate(rng, base, snv=0.01, indel=0.003)
    dad = pc.mutate(rng, base, snv=0.02, indel=0.003)
    path = str(tmp_path / "sd.ctx")
    orc.build_graph(path, [("kid", [kid]), ("mom", [base]), ("dad", [dad])], k)
    og = orc.Graph(path, tuned=True)
    rl = max(3 * k, 60)
    link_path = str(tmp_path / "sd.kid.ctp.gz")
    orc.build_links(og, link_path, "kid", [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), max(1, rl // 4))] + [kid[-rl:]])
    ol = orc.Links(link_path)
    pos = rng.sample(range(0, len(kid) - k - 200), 36)
    sources = [kid[p:p + k] for p in pos] + [pc.rand_seq(rng, k), "N" * k]
    sinks = [[kid[p + d:p + d + k]] for p, d in ((p, rng.randint(20, 180)) for p in pos)] + [[kid[5:5 + k]], []]
    sources[3] = orc.revcomp(sources[3])
    sinks[5] = [sinks[5][0], "N" * k, pc.rand_seq(rng, k)]
    cfgs = [("DestinationStopper", [0], 1, 400, True), ("DestinationStopper", [0], 0, 400, False), ("ExplorationStopper", [0], 0, 150, True),
            ("ContigStopper", [0, 1], 0, 300, True), ("DestinationStopper", [1], 2, 120, True),
            # rules that consult the ROI graph (the child's novel k-mers), which every rank holds whole
            ("NovelContinuationStopper", [0], 0, 200, True), ("NovelKmerLimitedContigStopper", [0], 0, 200, False), ("NahrStopper", [0], 0, 200, True)]
    parents = set()
    for h in (base, dad):
        parents |= {orc.canonical(h[i:i + k]) for i in range(len(h) - k + 1)}
    novel = [kid[i:i + k] for i in range(len(kid) - k + 1) if orc.canonical(kid[i:i + k]) not in parents]
    roi_path = str(tmp_path / "sd.rois.ctx")
    orc.build_graph(roi_path, [("kid", novel or [kid[:k]])], k)
    oroi = orc.Graph(roi_path, tuned=True)
    sources = sources + novel[:6]
    sinks = sinks + [[] for _ in novel[:6]]
    expected, use = [], []
    in_graph = [i for i in range(len(sources)) if og.find(sources[i])[0] >= 0]
    for stopper, trav, direction, max_len, wl in cfgs:
        with_roi = stopper.startswith(("Novel", "Nahr", "BubbleOpening"))
        oe = orc.Engine(og, trav, links=[ol] if wl else [], direction=direction, max_length=max_len, stopper=stopper,
                        rois=oroi if with_roi else None, joining_colors=[1, 2] if with_roi else ())
        it0 = oe.kmers_traversed()
        per = []
        # (the ROI rules dereference the record of the vertex they stand on: NullPointerException for a source that is not in the graph)
        use.append(in_graph if with_roi else list(range(len(sources))))
        for s_, sk in ((sources[i], sinks[i]) for i in use[-1]):
            r = oe.dfs(s_, sk)
            if r.is_null:
                per.append(None)
            else:
                per.append(([(km, rec >= 0, ci, ix) for km, rec, ci, ix in r.vertices()], r.edges(), r.walk(s_, trav[0])))
            r.free()
        per.append(oe.kmers_traversed() - it0)
        expected.append(per)
    if mode == "overflow":        # an image of 40 rows: the searches fill it, every rank gives the batch up in the same round, grows, runs it again
        keep = [0, 2, 5]
        _spawn(_sharded_dfs_worker, (path, link_path, sources, sinks, [cfgs[i] for i in keep], [expected[i] for i in keep], roi_path, [use[i] for i in keep], 40))
    elif mode == "poison":        # a rank-local NullPointerException (a ROI rule on a source without a record) must fail the batch on every rank
        keep = [7]
        _spawn(_sharded_dfs_worker, (path, link_path, sources, sinks, [cfgs[i] for i in keep], [expected[i] for i in keep], roi_path, [use[i] for i in keep], None,
                                     pc.rand_seq(random.Random(4242), k)))
    else:
        _spawn(_sharded_dfs_worker, (path, link_path, sources, sinks, cfgs, expected, roi_path, use))
