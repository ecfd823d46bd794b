"""THE parity gate: every case of tests/parity_cases.py through the HIP library (libldbg.so, gfx950)
on a real MI355X, compared bit-exactly with the CPU oracle.  Run with `pytest -m gpu`."""
import pytest
import torch  # noqa: F401  (before libldbg: both bring a HIP runtime; torch's must be the one that initialises first)

from tests import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import corticall_amd as ca
    l = ca.default_lib()
    assert l.device_count() >= 1, "no MI355X visible: the product has no CPU fallback"
    return l


def test_fixture_graph(orc, lib, tmp_path): pc.case_fixture_graph(orc, lib, tmp_path)


@pytest.mark.parametrize("k,ncol", [(5, 1), (21, 2), (31, 3), (32, 1), (33, 2), (47, 3), (63, 3), (64, 1), (65, 2), (95, 1)])
def test_random_find(orc, lib, tmp_path, k, ncol): pc.case_random_find(orc, lib, tmp_path, k, ncol)


@pytest.mark.parametrize("k", [32, 64, 96])
def test_all_bits_kmers(orc, lib, tmp_path, k): pc.case_all_bits_kmers(orc, lib, tmp_path, k)


def test_q1_tiny(orc, lib, tmp_path): pc.case_q1_tiny(orc, lib, tmp_path)
def test_unsorted_rejected(orc, lib, tmp_path): pc.case_unsorted_rejected(orc, lib, tmp_path)
def test_record_count_guard(orc, lib, tmp_path, monkeypatch): pc.case_record_count_guard(orc, lib, tmp_path, monkeypatch)
def test_rejected_open_frees_device_memory(lib, tmp_path): pc.case_rejected_open_frees_device_memory(lib, tmp_path)
def test_ref_short_contig_reconstruction(orc, lib, tmp_path): pc.test_ref_short_contig_reconstruction(orc, lib, tmp_path)
def test_ref_recruitment(orc, lib, tmp_path): pc.test_ref_recruitment(orc, lib, tmp_path)
def test_ref_cycles_without_and_with_links(orc, lib, tmp_path): pc.test_ref_cycles_without_and_with_links(orc, lib, tmp_path)
def test_ref_iterate_fwd_rev(orc, lib, tmp_path): pc.test_ref_iterate_fwd_rev(orc, lib, tmp_path)
def test_ref_go_forward_and_backward(orc, lib, tmp_path): pc.test_ref_go_forward_and_backward(orc, lib, tmp_path)
def test_ref_link_guided_walk(orc, lib, tmp_path): pc.test_ref_link_guided_walk(orc, lib, tmp_path)


@pytest.mark.parametrize("k,seed,links", [(9, 1, False), (9, 2, True), (21, 3, False), (31, 4, True), (47, 5, True), (63, 6, False), (33, 7, True), (32, 8, False), (64, 9, False), (65, 10, False)])
def test_random_walks(orc, lib, tmp_path, k, seed, links): pc.case_random_walks(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("seed", range(8))
def test_dense_cycles(orc, lib, tmp_path, seed): pc.case_dense_cycles(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("seed", range(10))
def test_run_steps(orc, lib, tmp_path, seed): pc.case_run_steps(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 109, 231])      # (109: siblings that cut one stretch in different places, tools/soak_gpu.py; 231: a branch that opens INSIDE a stretch an ancestor crossed, tests/test_soak_hostsim.py)
def test_dfs_run_steps(orc, lib, tmp_path, seed): pc.case_dfs_run_steps(orc, lib, tmp_path, seed)


def test_long_walks(orc, lib, tmp_path): pc.case_long_walks(orc, lib, tmp_path)


def test_link_formats(orc, lib, tmp_path): pc.case_link_formats(orc, lib, tmp_path)


def test_sort(orc, lib, tmp_path): pc.case_sort(orc, lib, tmp_path)
def test_sort_rewrites_header(orc, lib, tmp_path): pc.case_sort_rewrites_header(orc, lib, tmp_path)
def test_join(orc, lib, tmp_path): pc.case_join(orc, lib, tmp_path)


def test_collection(orc, lib, tmp_path): pc.case_collection(orc, lib, tmp_path)


def test_ref_collection(orc, lib, tmp_path): pc.test_ref_collection(orc, lib, tmp_path)


def test_big_link_stores(orc, lib, tmp_path): pc.case_big_link_stores(orc, lib, tmp_path)


def test_hash_collision(orc, lib, tmp_path): pc.case_hash_collision(orc, lib, tmp_path)


@pytest.mark.parametrize("k,seed,links", [(9, 2, False), (21, 3, False), (31, 4, True), (47, 5, True)])
def test_dfs_rules(orc, lib, tmp_path, k, seed, links): pc.case_dfs_rules(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("seed", range(4))
def test_dfs_dense(orc, lib, tmp_path, seed): pc.case_dfs_dense(orc, lib, tmp_path, seed)


def test_ref_dfs_with_sinks(orc, lib, tmp_path): pc.test_ref_dfs_with_sinks(orc, lib, tmp_path)
def test_ref_multiple_traversal_colors(orc, lib, tmp_path): pc.test_ref_multiple_traversal_colors(orc, lib, tmp_path)


def test_dfs_packed_results(orc, lib, tmp_path): pc.case_dfs_packed_results(orc, lib, tmp_path)


def test_dfs_second_launch_without_the_index(orc, lib, tmp_path, monkeypatch):
    """a search the run steps hand back sends its chunk round again without the run index: forced here, the results must not change"""
    monkeypatch.setenv("LDBG_DFS_FORCE_RETRY", "1")
    pc.case_dfs_run_steps(orc, lib, tmp_path, 0)
    pc.case_dfs_dense(orc, lib, tmp_path, 1)


def test_factory_validation(orc, lib, tmp_path): pc.case_factory_validation(orc, lib, tmp_path)


def test_ref_fill_gaps(orc, lib, tmp_path): pc.test_ref_fill_gaps(orc, lib, tmp_path)


@pytest.mark.parametrize("seed", range(3))
def test_fill_gaps_random(orc, lib, tmp_path, seed): pc.case_fill_gaps_random(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("k,seed,links", [(21, 1, False), (31, 2, True), (47, 3, True)])
def test_partition(orc, lib, tmp_path, k, seed, links): pc.case_partition(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("k,seed,links", [(21, 1, False), (31, 2, True), (47, 3, True), (32, 4, True)])
def test_findtips(orc, lib, tmp_path, k, seed, links): pc.case_findtips(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("k,seed,links", [(21, 1, False), (31, 2, True), (47, 3, True), (32, 4, False)])
def test_facade(orc, lib, tmp_path, k, seed, links): pc.case_facade(orc, lib, tmp_path, k, seed, links)


def test_sharded_find_one_rank_rccl(orc, lib, tmp_path):
    """the exchange path of corticall_amd/distributed.py over RCCL with device buffers (one rank: this box has one GPU;
    the two-rank case runs on gloo in tests/test_distributed.py)"""
    import os
    import random
    import torch.distributed as dist
    from corticall_amd.distributed import ShardedCortexGraph
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        rng = random.Random(2)
        p = str(tmp_path / "sh.ctx")
        orc.build_graph(p, [("a", [pc.rand_seq(rng, 3000)]), ("b", [pc.rand_seq(rng, 2000)])], 47)
        og = orc.Graph(p, tuned=True)
        sg = ShardedCortexGraph(p, lib=lib)
        kmers = [og.record_string(i).split()[0] for i in rng.sample(range(og.N), 500)]
        qs = [k if rng.random() < 0.5 else orc.revcomp(k) for k in kmers] + [pc.rand_seq(rng, 47) for _ in range(200)] + ["N" * 47]
        found, cov, edges, owner, lidx = sg.find_batch(qs)
        for i, q in enumerate(qs):
            eidx, ecov, eedges = og.find(q)
            assert bool(found[i]) == (eidx >= 0)
            if eidx >= 0:
                assert ecov == [int(c) for c in cov[i]] and eedges == [int(x) for x in edges[i]] and lidx[i] == eidx
        assert len(sg.find_batch([])[0]) == 0
        sg.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,with_links", [(47, False), (47, True), (32, True), (63, True)])
def test_sharded_walks_one_rank_rccl(orc, lib, tmp_path, k, with_links):
    """walks over a sharded table's local image with device buffers and RCCL collectives, every call of a round queued on torch's
    stream (one rank here; two and three ranks on gloo in tests/test_distributed.py): link-guided and plain, odd and even k"""
    import os
    import random
    import torch.distributed as dist
    from corticall_amd import CortexLinks
    from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        rng = random.Random(8 + k)
        base = pc.genome_with_repeats(rng, 2500, n_rep=6, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
        kid = pc.mutate(rng, base, snv=0.01, indel=0.003)
        p = str(tmp_path / "sw.ctx")
        orc.build_graph(p, [("kid", [kid]), ("mom", [base])], k)
        og = orc.Graph(p, tuned=True)
        ol, lp = None, None
        if with_links:
            rl = max(3 * k, 60)
            lp = str(tmp_path / "sw.kid.ctp.gz")
            orc.build_links(og, lp, "kid", [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), max(1, rl // 4))] + [kid[-rl:]])
            ol = orc.Links(lp)
        sg = ShardedCortexGraph(p, lib=lib)
        links = [CortexLinks(lp, sg.shard, lib=lib)] if with_links else []
        kmers = [og.record_string(i).split()[0] for i in range(og.N)]
        seeds = rng.sample(kmers, 200)
        seeds = [s if rng.random() < 0.5 else orc.revcomp(s) for s in seeds] + [pc.rand_seq(rng, k), "N" * k]
        ML = 600 if with_links else 75000
        for trav, direction, op in (([0], 0, 0), ([1], 1, 1), ([0, 1], 2, 0)):
            oe = orc.Engine(og, trav, links=[ol] if ol else [], op_and=(op == 1), direction=direction, stopper="ContigStopper", max_length=ML)
            it0 = oe.kmers_traversed()
            exp = [oe.walk(s)[0] for s in seeds]
            e = ShardedTraversalEngine(sg, trav, links=links, direction=direction, op=op, max_branch_length=ML, rows_per_owner=256)
            got = e.walk_batch(seeds)
            assert got == exp
            assert e.kmers_traversed == oe.kmers_traversed() - it0
            assert e.walk_batch(seeds) == exp and e.rounds > 0       # again, from an empty image
            e.close()
        sg.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k", [21, 32])
def test_sharded_dfs_one_rank_rccl(orc, lib, tmp_path, k):
    """dfs with stopping rules over a sharded table's local image on the device, RCCL collectives (one rank; two ranks on gloo in
    tests/test_distributed.py): DestinationStopper towards a sink, ExplorationStopper, and rules that consult a ROI graph"""
    import os
    import random
    import torch.distributed as dist
    from corticall_amd import CortexGraph, CortexLinks
    from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29521")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
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
        parents = set()
        for h in (base, dad):
            parents |= {orc.canonical(h[i:i + k]) for i in range(len(h) - k + 1)}
        novel = [kid[i:i + k] for i in range(len(kid) - k + 1) if orc.canonical(kid[i:i + k]) not in parents]
        roi_path = str(tmp_path / "sd.rois.ctx")
        orc.build_graph(roi_path, [("kid", novel or [kid[:k]])], k)
        oroi, rois = orc.Graph(roi_path, tuned=True), CortexGraph(roi_path, lib=lib)
        pos = rng.sample(range(0, len(kid) - k - 200), 30)
        sources = [kid[p:p + k] for p in pos] + novel[:6]
        sinks = [[kid[p + d:p + d + k]] for p, d in ((p, rng.randint(20, 180)) for p in pos)] + [[] for _ in novel[:6]]
        sg = ShardedCortexGraph(path, lib=lib)
        links = CortexLinks(link_path, sg.shard, lib=lib)
        for stopper, trav, direction, max_len, wl in (("DestinationStopper", [0], 1, 400, True), ("ExplorationStopper", [0], 0, 150, True),
                                                     ("NovelContinuationStopper", [0], 0, 200, True), ("NahrStopper", [0], 0, 200, False)):
            with_roi = stopper.startswith(("Novel", "Nahr"))
            oe = orc.Engine(og, trav, links=[ol] if wl else [], direction=direction, max_length=max_len, stopper=stopper,
                            rois=oroi if with_roi else None, joining_colors=[1, 2] if with_roi else ())
            it0 = oe.kmers_traversed()
            e = ShardedTraversalEngine(sg, trav, links=[links] if wl else (), direction=direction, max_branch_length=max_len, stopping_rule=stopper,
                                       rows_per_owner=256, check_every=4, rois=rois if with_roi else None, joining_colors=[1, 2] if with_roi else ())
            got = e.dfs_batch(sources, sinks)
            for s_, sk, gi in zip(sources, sinks, got):
                r = oe.dfs(s_, sk)
                assert (gi is None) == r.is_null, (stopper, s_)
                if gi is not None:
                    assert [(km, rec >= 0, ci, ix) for km, rec, ci, ix in gi.vertex_tuples()] == [(km, rec >= 0, ci, ix) for km, rec, ci, ix in r.vertices()]
                    assert gi.edge_tuples() == r.edges() and gi.walk_contig(s_, trav[0]) == r.walk(s_, trav[0])
                r.free()
            assert e.dfs_kmers_traversed == oe.kmers_traversed() - it0 and e.rounds > 0
            e.close()
        sg.close()
    finally:
        dist.destroy_process_group()


def test_dfs_step_limit(orc, lib, tmp_path, monkeypatch): pc.case_dfs_step_limit(orc, lib, tmp_path, monkeypatch)


def test_close_in_any_order(orc, lib, tmp_path): pc.case_close_in_any_order(orc, lib, tmp_path)


def test_small_visited_tables(orc, lib, tmp_path, monkeypatch):
    """visited tables that start at 64 entries: probe rounds wrap round the end of a table and tables regrow many times"""
    monkeypatch.setenv("LDBG_VT_INITIAL", "64")
    pc.case_long_walks(orc, lib, tmp_path)
    pc.case_dense_cycles(orc, lib, tmp_path, 2)
    pc.case_dfs_dense(orc, lib, tmp_path, 3)
    pc.case_dfs_rules(orc, lib, tmp_path, 31, 2, True)


def test_concurrent_engines(orc, lib, tmp_path): pc.case_concurrent_engines(orc, lib, tmp_path)


def test_lowercase_queries(orc, lib, tmp_path): pc.case_lowercase_queries(orc, lib, tmp_path)
