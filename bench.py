#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

metric   : k-mers traversed / s (whole job), contigs / s alongside
workload : configs[2] — synthetic P. falciparum-scale (23,332,839 bp) 3-colour k=47 LdBG with child links,
           link-guided contig walks (ContigStopper, BOTH, OR: the `Partition` configuration,
           Partition.java:85-94) from 50,000 seed k-mers (de novo mutation k-mers padded with random
           child k-mers).  One "step" = one walk_batch over all seeds of the rank; the graph, the links
           and the seeds (2.35 MB of ASCII, an (n, k) uint8 device array) are resident in HBM before the
           timed region and the results stay there; `host_seeds` is the same step with the seeds handed
           over as a host buffer, `with_contigs_fetched` with every contig brought back as well.
           --in-flight N (default 2): the K steps are dealt out to N engines on the one graph, each with its
           own HIP stream and host thread (a launch is bound by its longest strands; another batch fills
           the compute units its early finishers leave); `single_batch` = the same steps one at a time.
N > 1    : one process per GPU (torchrun); every rank holds a replica of the graph (it fits: ~2 GB) and
           walks its own 50,000 seeds — independent units, no data-path collective, weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME_LEN = 23332839          # AssembleReads.wdl:530
K = 47                         # AssembleReads.wdl:140
N_SEEDS = 50000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def workload_files(args, rank, n_seeds=None):
    """generate (or reuse) the synthetic inputs; returns (prefix, stats)"""
    from tools import synth
    n_seeds = n_seeds or args.seeds
    tag = "c3_L%d_k%d_s%d_r%d" % (args.genome_len, args.k, n_seeds, rank)
    d = os.environ.get("LDBG_BENCH_DIR", "/tmp/ldbg_bench")
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, tag)
    meta = prefix + ".json"
    if os.path.exists(meta) and os.path.exists(prefix + ".ctx"):
        return prefix, json.load(open(meta))
    t = time.time()
    # every rank builds the same graph (same seed); only the seed-list RNG stream differs per rank
    st = synth.generate(prefix, args.genome_len, args.k, colours=3, with_links=True, seed=0xC0FFEE03, n_chrom=14,
                        n_repeat_families=args.repeat_families, repeat_copies=4, repeat_len=(50, 300),
                        n_seeds=n_seeds, threads=min(16, os.cpu_count() or 1))
    st["gen_seconds"] = round(time.time() - t, 1)
    json.dump(st, open(meta, "w"))
    return prefix, st


def cpu_baseline(prefix, args, gpu_contigs, seeds_ascii, n_batch):
    """oracle ("port" of the reference algorithm: ASCII 3-point search, LRU, link store) on a bounded sample"""
    import numpy as np
    from oracle import pyoracle as orc
    g = orc.Graph(prefix + ".ctx", use_cache=True, tuned=False)
    l = orc.Links(prefix + ".ctp.gz")
    e = orc.Engine(g, [0], links=[l], stopper="ContigStopper", max_length=args.max_len)
    budget = args.cpu_seconds
    t0 = time.time()
    n = 0
    mismatches = 0
    while n < len(seeds_ascii) and time.time() - t0 < budget:        # (seeds_ascii is a random sample of the batch, drawn by the caller)
        contig, _ = e.walk(seeds_ascii[n].tobytes().decode())
        if contig != gpu_contigs[n]:
            mismatches += 1
        n += 1
    dt = time.time() - t0
    trav = e.kmers_traversed()
    return {
        "value": trav / dt, "unit": "k-mers traversed/s", "cores": 1, "kind": "port",
        "sample": "%d seeds drawn at random (seed 20261004) from the %d of rank 0 (%d k-mers traversed in %.1f s), oracle in faithful mode "
                  "(ASCII 3-point binary search + 1M-entry LRU, CortexGraph.java:272-317)" % (n, n_batch, trav, dt),
        "contigs_per_s": n / dt,
    }, n, mismatches


def cpu_baseline_tuned(prefix, args, seeds_ascii, seconds=8.0):
    """the same algorithm with the CPU-side tuning one would do first — packed binary search without the
    ASCII decode / LRU, one engine per host thread over disjoint seeds.  Reported beside the faithful port, not instead."""
    import threading
    from oracle import pyoracle as orc
    n_threads = min(16, os.cpu_count() or 1)
    done = [0] * n_threads
    trav = [0] * n_threads
    ready = threading.Barrier(n_threads + 1)
    t_end = [0.0]

    def work(t):
        g = orc.Graph(prefix + ".ctx", use_cache=False, tuned=True)
        l = orc.Links(prefix + ".ctp.gz")
        e = orc.Engine(g, [0], links=[l], stopper="ContigStopper", max_length=args.max_len)
        ready.wait()                                     # loading is not part of the measurement
        ready.wait()
        i = len(seeds_ascii) - 1 - t                     # from the far end of the seed list, away from the faithful sample
        while i >= 0 and time.time() < t_end[0]:
            e.walk(seeds_ascii[i].tobytes().decode())
            done[t] += 1
            i -= n_threads
        trav[t] = e.kmers_traversed()

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    ready.wait()
    t0 = time.time()
    t_end[0] = t0 + seconds
    ready.wait()
    for th in ths:
        th.join()
    dt = time.time() - t0
    return {"value": sum(trav) / dt, "unit": "k-mers traversed/s", "cores": n_threads, "kind": "port",
            "sample": "%d seeds over %d threads (%d k-mers traversed in %.1f s), oracle in tuned mode (packed binary search, no LRU), "
                      "one engine per thread" % (sum(done), n_threads, sum(trav), dt)}


def bench_c2(args, ca, rank, local_rank, world, dist):
    """configs[1]: synthetic 10 Mb 1-colour k=31 graph, batches of random-access lookups (50 % present, random
    orientation; 50 % random absent), queries resident in HBM.  --sharded: the table is hash-partitioned over the
    ranks and every batch is routed with all-to-all exchanges (corticall_amd/distributed.py)."""
    import ctypes as C
    import numpy as np
    import torch
    from tools import synth
    from corticall_amd import CortexGraph
    from corticall_amd.distributed import ShardedCortexGraph, pack_kmers
    k, L = 31, 10_000_000
    d = os.environ.get("LDBG_BENCH_DIR", "/tmp/ldbg_bench")
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "c2_L%d_k%d" % (L, k))
    if dist is not None and rank != 0:
        dist.barrier()                  # rank 0 writes the files
    if not os.path.exists(prefix + ".ctx"):
        synth.generate(prefix, L, k, colours=1, with_links=False, seed=0xC0FFEE01, n_chrom=1, n_repeat_families=0, repeat_copies=0,
                       n_indels=0, n_dnm=0, n_tandem=0, n_seeds=100000, threads=min(16, os.cpu_count() or 1))
    if dist is not None and rank == 0:
        dist.barrier()
    present = np.fromfile(prefix + ".seeds", dtype=np.uint8).reshape(-1, k)
    rng = np.random.default_rng(0xC0FFEE02 + rank)
    n = args.lookups
    q = np.empty((n, k), dtype=np.uint8)
    half = n // 2
    q[:half] = present[rng.integers(0, len(present), half)]
    q[half:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n - half, k))]
    q = q[rng.permutation(n)]
    dev = torch.device("cuda", local_rank)
    words = torch.from_numpy(pack_kmers(q, k).view(np.int64)).to(dev)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if args.sharded:
        g = ShardedCortexGraph(prefix + ".ctx", device=local_rank)
        N, W, Cc = g.getNumRecords(), g.W, g.C
        step = lambda: g.find_packed_dev(words)
    else:
        g = CortexGraph(prefix + ".ctx", device=local_rank)
        N, W, Cc = g.getNumRecords(), g.getKmerBits(), g.getNumColors()
        idx = torch.empty(n, dtype=torch.int64, device=dev)
        cov = torch.empty((n, Cc), dtype=torch.int32, device=dev)
        edges = torch.empty((n, Cc), dtype=torch.uint8, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())
        step = lambda: g._lib.check(g._d.ldbg_graph_find_dev(g._h, P(words), C.c_int64(n), P(idx), P(cov), P(edges), None))
    for _ in range(args.warmup):
        step()
    ca.profile_reset()
    sync()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.time() - t0
    find_ms, launches = ca.profile_get("find")
    max_dt = dt
    if dist is not None:
        m = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        max_dt = m[0].item()
    if rank == 0:
        b_find = math.ceil(math.log2(N)) * 8 * W + 5 * Cc
        avg_ms = find_ms / max(1, launches)
        per_launch = n if not args.sharded else n       # every rank's shard answers about n lookups per step
        achieved = per_launch * b_find / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        found = int((idx >= 0).sum().item()) if not args.sharded else int(g.find_batch(q)[0].sum())
        cpu = parity = None
        if not args.no_cpu_baseline:
            # the reference's findRecord (ASCII 3-point binary search + 1 M-entry LRU, CortexGraph.java:272-317) restated in oracle/, one core,
            # on a bounded sample of the same queries; its answers check the device's
            from oracle import pyoracle as orc
            og = orc.Graph(prefix + ".ctx", use_cache=True, tuned=False)
            m = min(n, 200000)
            t2 = time.time()
            done = 0
            ref = np.empty(0, dtype=np.int64)
            while done < m and time.time() - t2 < args.cpu_seconds:
                part = og.find_batch(q[done:done + 20000])
                ref = np.concatenate([ref, part])
                done += len(part)
            dtc = time.time() - t2
            cpu = {"value": done / dtc, "unit": "lookups/s", "cores": 1, "kind": "port",
                   "sample": "the first %d of the step's %d queries in %.1f s, oracle in faithful mode (ASCII 3-point search + LRU)" % (done, n, dtc)}
            if not args.sharded:
                dev_idx = idx[:done].cpu().numpy()
                parity = "%d/%d sampled lookups identical to the oracle's (record index, -1 = absent)" % (int((dev_idx == ref).sum()), done)
        print(json.dumps({
            "metric": "random-access lookups/sec (configs[1]; not the headline metric)", "value": world * n * args.steps / max_dt, "unit": "lookups/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": max_dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "library": ca.default_lib().dll.ldbg_version().decode(),
            "config": {"workload": "configs[1]: synthetic 10 Mb 1-colour k=31 graph, %d lookups per step and GPU (50%% present)%s"
                                   % (n, ", table hash-sharded over the ranks, all-to-all routed" if args.sharded else ""),
                       "records": N, "found": found},
            "roofline": {"bound": "hbm", "kernel": "k_find<%d>" % W, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_lookup": b_find,
                         "avg_launch_ms": avg_ms, "launches": launches},
            "cpu_baseline": cpu, "parity": parity,
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_c3_sharded(args, ca, prefix, st, seeds, rank, local_rank, world, dist):
    """configs[2]'s workload (and configs[4]'s path) over the HASH-SHARDED table: link-guided ContigStopper walks; every rank holds
    1/world of the records, keeps a local image of the rows it is sent and runs the walk kernel on it
    (corticall_amd/distributed.py::ShardedTraversalEngine, csrc/image.h).  The image starts empty in every step."""
    import numpy as np
    import torch
    from corticall_amd import CortexLinks
    from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine
    t0 = time.time()
    sg = ShardedCortexGraph(prefix + ".ctx", device=local_rank)
    t_shard = time.time() - t0
    sg.build_neighbour_index()
    t_load = time.time() - t0
    links = [] if args.no_links else [CortexLinks(prefix + ".ctp.gz", sg.shard)]
    eng = ShardedTraversalEngine(sg, [0], links=links, max_branch_length=args.max_len, rows_per_owner=args.rows_per_owner,
                                 check_every=args.check_every, chain_depth=args.chain_depth)
    mine = [s.tobytes().decode() for s in seeds[:args.sharded_seeds]]

    def sync():
        torch.cuda.synchronize()
        dist.barrier()

    for _ in range(args.warmup):
        eng.walk_batch(mine)
    sync()
    t1 = time.time()
    traversed = rounds = 0
    for _ in range(args.steps):
        contigs = eng.walk_batch(mine)
        traversed += eng.kmers_traversed
        rounds += eng.rounds
    sync()
    dt = time.time() - t1
    t = torch.tensor([float(traversed), float(len(mine) * args.steps)], dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    m = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    if rank == 0:
        N, W, C = sg.getNumRecords(), sg.W, sg.C
        M = max(2, links[0].numKmersWithLinks) if links else 2
        b_find = math.ceil(math.log2(N)) * 8 * W + 5 * C
        b_link = math.ceil(math.log2(M)) * 8 * W if links else 0
        per_round_ms = m[0].item() / max(1, rounds) * 1e3
        achieved = t[0].item() * (b_find + b_link) / m[0].item() / 1e9 / max(1, world)
        out = {
            "metric": "k-mers traversed/sec (whole node) + contigs/sec, k=47 3-color LdBG", "value": t[0].item() / m[0].item(),
            "unit": "k-mers traversed/s", "contigs_per_s": t[1].item() / m[0].item(), "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": m[0].item() / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic", "library": ca.default_lib().dll.ldbg_version().decode(),
            "config": {"workload": "configs[2] over the HASH-SHARDED table (configs[4]'s path): synthetic %.1f Mb 3-colour k=%d LdBG%s, table split over %d rank(s) by "
                                   "the mixed minimizer of the canonical k-mer, link-guided ContigStopper walks BOTH/OR from %d seeds per GPU, maxLength %d; rows fetched from their "
                                   "owners on demand into a local image (empty at the start of every step; an owner sends the rows around the one asked for along with it), "
                                   "RCCL all-to-all per bulk-synchronous round"
                                   % (args.genome_len / 1e6, args.k, "" if args.no_links else " with child links (replicated on every rank)", world, len(mine), args.max_len),
                       "records": N, "records_per_rank": N // max(1, world), "rounds_per_step": rounds // max(1, args.steps), "ms_per_round": per_round_ms,
                       "rows_per_owner_and_round": args.rows_per_owner, "row_slots_per_request": args.chain_depth, "image_rows_used": eng.image_rows_used,
                       "kmers_traversed_per_step": traversed // max(1, args.steps), "multi_gpu": "hash-sharded table, rows exchanged, walks stay on the rank of their seed",
                       "load_seconds": round(t_load, 2), "shard_cut_seconds": round(t_shard, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_walk<%d> on the image, one launch per round" % W, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_kmer": b_find + b_link,
                         "note": "per GPU; the bulk-synchronous rounds (launch + two all-to-alls each), not HBM, bound this regime: see ms_per_round"},
        }
        if not args.no_cpu_baseline:
            from oracle import pyoracle as orc
            og = orc.Graph(prefix + ".ctx", use_cache=True, tuned=False)
            oe = orc.Engine(og, [0], links=[] if args.no_links else [orc.Links(prefix + ".ctp.gz")], stopper="ContigStopper", max_length=args.max_len)
            pick = np.random.default_rng(20261004).permutation(len(mine))
            t2 = time.time()
            i = mism = 0
            while i < len(mine) and time.time() - t2 < args.cpu_seconds:
                mism += 0 if oe.walk(mine[pick[i]])[0] == contigs[pick[i]] else 1
                i += 1
            dtc = time.time() - t2
            out["cpu_baseline"] = {"value": oe.kmers_traversed() / dtc, "unit": "k-mers traversed/s", "cores": 1, "kind": "port",
                                   "sample": "%d seeds drawn at random from rank 0's (%d k-mers traversed in %.1f s), oracle in faithful mode" % (i, oe.kmers_traversed(), dtc)}
            out["parity"] = "%d/%d sampled contigs bit-exact vs oracle" % (i - mism, i)
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def bench_c4_sharded(args, ca, g, links, walk_eng, seeds, st, prefix, rank, local_rank, world, dist, sync):
    """configs[3] as BASELINE names it: DestinationStopper dfs over the HASH-SHARDED table.  Sinks are drawn as in bench_c4 (from the
    seeds' own contigs, computed on a replica of the graph: a benchmark convenience); the searches then run over local images of the
    sharded table, rows fetched from their owners per bulk-synchronous round (corticall_amd/distributed.py)."""
    import numpy as np
    import torch
    from corticall_amd import CortexLinks, DestinationStopper
    from corticall_amd.distributed import ShardedCortexGraph, ShardedTraversalEngine
    k = args.k
    n = min(len(seeds), args.sharded_seeds)
    seeds = seeds[:n]
    arena, offs, wl = walk_eng.walk_batch_arrays(seeds)
    rng = np.random.default_rng(0xC0FFEE05 + rank)
    sinks = []
    for i in range(n):
        c = arena[offs[i]:offs[i + 1]]
        p = c.tobytes().find(seeds[i].tobytes()) if len(c) >= k else -1
        if p < 0:
            sinks.append([seeds[(i + 1) % n].tobytes().decode()])
            continue
        q = min(len(c) - k, p + int(rng.integers(200, 2001)))
        sinks.append([c[q:q + k].tobytes().decode()])
    sources = [s.tobytes().decode() for s in seeds]
    t0 = time.time()
    sg = ShardedCortexGraph(prefix + ".ctx", device=local_rank)
    sg.build_neighbour_index()
    t_load = time.time() - t0
    slinks = [CortexLinks(prefix + ".ctp.gz", sg.shard)]
    eng = ShardedTraversalEngine(sg, [0], links=slinks, direction=1, max_branch_length=args.max_len, stopping_rule=DestinationStopper,
                                 rows_per_owner=args.rows_per_owner, check_every=args.check_every, chain_depth=args.chain_depth)
    for _ in range(args.warmup):
        eng.dfs_batch(sources, sinks)
    sync()
    t1 = time.time()
    traversed = rounds = 0
    for _ in range(args.steps):
        graphs = eng.dfs_batch(sources, sinks)
        traversed += eng.dfs_kmers_traversed
        rounds += eng.rounds
    sync()
    dt = time.time() - t1
    t = torch.tensor([float(traversed), float(n * args.steps)], dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    m = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    if rank == 0:
        N, W, C = sg.getNumRecords(), sg.W, sg.C
        b = math.ceil(math.log2(N)) * 8 * W + 5 * C + math.ceil(math.log2(max(2, slinks[0].numKmersWithLinks))) * 8 * W
        achieved = t[0].item() * b / m[0].item() / 1e9 / max(1, world)
        out = {
            "metric": "k-mers traversed/sec (whole node) + contigs/sec, k=47 3-color LdBG", "value": t[0].item() / m[0].item(),
            "unit": "k-mers traversed/s", "contigs_per_s": t[1].item() / m[0].item(), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m[0].item() / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic", "library": ca.default_lib().dll.ldbg_version().decode(),
            "config": {"workload": "configs[3]: same %.1f Mb 3-colour k=%d LdBG with child links, dfs with DestinationStopper (FORWARD) from %d seeds per GPU to the child "
                                   "k-mer 200-2000 bp downstream, table HASH-SHARDED over %d rank(s), rows fetched on demand into local images, RCCL all-to-all per round"
                                   % (args.genome_len / 1e6, k, n, world),
                       "records": N, "rounds_per_step": rounds // max(1, args.steps), "ms_per_round": m[0].item() / max(1, rounds) * 1e3,
                       "sinks_reached": sum(1 for x in graphs if x is not None), "kmers_traversed_per_step": traversed // max(1, args.steps),
                       "multi_gpu": "hash-sharded table, rows exchanged, searches stay on the rank of their source", "load_seconds": round(t_load, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_dfs<%d> on the image, one launch per round" % W, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_kmer": b,
                         "note": "per GPU; the bulk-synchronous rounds, not HBM, bound this regime: see ms_per_round"},
        }
        if not args.no_cpu_baseline:
            from oracle import pyoracle as orc
            og = orc.Graph(prefix + ".ctx", use_cache=True, tuned=False)
            oe = orc.Engine(og, [0], links=[orc.Links(prefix + ".ctp.gz")], stopper="DestinationStopper", max_length=args.max_len, direction=orc.FORWARD)
            pick = np.random.default_rng(20261004).permutation(n)
            t2 = time.time()
            i = mism = 0
            while i < n and time.time() - t2 < args.cpu_seconds:
                j = int(pick[i])
                r = oe.dfs(sources[j], sinks[j])
                gi = graphs[j]
                same = (gi is None) == r.is_null and (r.is_null or ([(a_, b_ >= 0, c_, d_) for a_, b_, c_, d_ in gi.vertex_tuples()] == [(a_, b_ >= 0, c_, d_) for a_, b_, c_, d_ in r.vertices()]
                                                                  and gi.edge_tuples() == r.edges()))
                mism += 0 if same else 1
                r.free()
                i += 1
            dtc = time.time() - t2
            out["cpu_baseline"] = {"value": oe.kmers_traversed() / dtc, "unit": "k-mers traversed/s", "cores": 1, "kind": "port",
                                   "sample": "%d searches drawn at random (%d k-mers traversed in %.1f s), oracle in faithful mode" % (i, oe.kmers_traversed(), dtc)}
            out["parity"] = "%d/%d sampled dfs graphs bit-exact vs oracle (vertices and edges in insertion order)" % (i - mism, i)
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def bench_c4(args, ca, g, links, walk_eng, seeds, st, prefix, rank, world, dist, sync, t_load):
    """configs[3]: dfs with DestinationStopper (the gap-closing configuration of Call.java:759-779) from every seed
    towards the child k-mer 200-2000 bp downstream on the seed's own link-guided contig."""
    import numpy as np
    import torch
    from corticall_amd import FORWARD, OR, DestinationStopper, TraversalEngineFactory
    k = args.k
    arena, offs, wl = walk_eng.walk_batch_arrays(seeds)
    rng = np.random.default_rng(0xC0FFEE05)
    sink = np.empty_like(seeds)
    for i in range(len(seeds)):
        c = arena[offs[i]:offs[i + 1]]
        sd = seeds[i].tobytes()
        p = c.tobytes().find(sd) if len(c) >= k else -1
        if p < 0:                      # seed not on its own contig (empty walk): an unrelated sink, the search fails
            sink[i] = seeds[(i + 1) % len(seeds)]
            continue
        d = int(rng.integers(200, 2001))
        q = min(len(c) - k, p + d)
        sink[i] = c[q:q + k]
    sink_off = np.arange(len(seeds) + 1, dtype=np.int64)
    sink_buf = np.ascontiguousarray(sink).reshape(-1)
    src = np.ascontiguousarray(seeds).reshape(-1)
    eng = (TraversalEngineFactory().traversalColors(g.getColorForSampleName("child")).traversalDirection(FORWARD)
           .combinationOperator(OR).stoppingRule(args.stopper).maxBranchLength(args.max_len).graph(g).links(links).make())
    n = len(seeds)
    # --in-flight N: N engines on the one graph (own HIP stream and host thread each) take the calls in turn, as for the walks
    engines = [eng]
    for _ in range(max(1, args.in_flight) - 1):
        engines.append(TraversalEngineFactory().traversalColors(g.getColorForSampleName("child")).traversalDirection(FORWARD)
                       .combinationOperator(OR).stoppingRule(args.stopper).maxBranchLength(args.max_len).graph(g).links(links).make())
    for _ in range(max(1, args.warmup)):
        for e2 in engines:
            e2.dfs_batch_arrays(src, n, sink_buf, sink_off)

    def run_calls(engs, n_calls):
        import threading
        trav, last = [0] * len(engs), [None] * len(engs)

        def work(i):
            for _ in range(i, n_calls, len(engs)):
                last[i] = engs[i].dfs_batch_arrays(src, n, sink_buf, sink_off)
                trav[i] += engs[i].dfs_kmers_traversed
        if len(engs) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        return sum(trav), next(x for x in last if x is not None)

    sync()
    t0 = time.time()
    run_calls(engines[:1], args.steps)
    sync()
    dt_single = time.time() - t0
    ca.profile_reset()
    sync()
    t0 = time.time()
    found = 0
    traversed, b = run_calls(engines, args.steps)
    sync()
    dt = time.time() - t0
    dfs_ms, launches = ca.profile_get("dfs")
    found = sum(1 for i in range(n) if b.graph(i) is not None)
    # the same call followed by reading a graph WITH its k-mers: results of one branch per direction stay packed until then, and the
    # k-mers and coverages of all vertices of the batch are gathered from the device on first use
    sync()
    t1 = time.time()
    b2 = eng.dfs_batch_arrays(src, n, sink_buf, sink_off)
    first_graph = next((b2.graph(i) for i in range(n) if b2.graph(i) is not None), None)
    if first_graph is not None:
        first_graph._fetch()
    sync()
    dt_fetched = time.time() - t1
    del b2, first_graph
    tot_trav, tot_seeds, max_dt = traversed, n * args.steps, dt
    if dist is not None:
        t = torch.tensor([float(traversed), float(n * args.steps)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        m = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        tot_trav, tot_seeds, max_dt = int(t[0].item()), int(t[1].item()), m[0].item()
    if rank == 0:
        N, W, C = g.getNumRecords(), g.getKmerBits(), g.getNumColors()
        M = max(2, links.numKmersWithLinks)
        b_find = math.ceil(math.log2(N)) * 8 * W + 5 * C
        b_link = math.ceil(math.log2(M)) * 8 * W
        avg_ms = dfs_ms / max(1, launches)
        achieved = (traversed / max(1, launches)) * (b_find + b_link) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "k-mers traversed/sec (whole node) + contigs/sec, k=47 3-color LdBG",
            "value": tot_trav / max_dt, "unit": "k-mers traversed/s", "contigs_per_s": tot_seeds / max_dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": max_dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "library": ca.default_lib().dll.ldbg_version().decode(),
            "config": {
                "workload": ("configs[3]: same %.1f Mb 3-colour k=%d LdBG with child links; dfs with %s, FORWARD, from %d seeds per GPU "
                             "to the child k-mer 200-2000 bp downstream on the seed's contig; timed: the C-ABI call (kernel, log expansion and download, "
                             "graph assembly on the host: results of one branch per direction are kept as packed vertex entries, the others "
                             "as vertex and edge lists)") % (args.genome_len / 1e6, k, args.stopper, n),
                "records": N, "seeds_per_gpu": n, "kmers_traversed_per_step": traversed // args.steps, "sinks_reached": found,
                "multi_gpu": "replicated graph, seeds partitioned, no data-path collective" if world > 1 else "single GPU",
                "load_seconds": round(t_load, 2),
            },
            "roofline": {"bound": "hbm", "kernel": "k_dfs<%d>" % W, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_kmer": b_find + b_link,
                         "avg_launch_ms": avg_ms, "launches": launches},
            "single_batch": {"ms_per_step": dt_single / args.steps * 1e3, "contigs_per_s": n * args.steps / dt_single if dt_single > 0 else None,
                             "note": "the same calls one at a time on one engine"},
            "batches_in_flight": len(engines),
            "with_graphs_fetched": {"ms_per_step": dt_fetched * 1e3, "value": (traversed / args.steps) / dt_fetched,
                                    "note": "one call + unpacking every graph of the batch + k-mers and coverages of all their vertices gathered from the device"},
        }
        if not args.no_cpu_baseline:
            from oracle import pyoracle as orc
            og = orc.Graph(prefix + ".ctx", use_cache=True, tuned=False)
            ol = orc.Links(prefix + ".ctp.gz")
            oe = orc.Engine(og, [0], links=[ol], stopper=args.stopper, max_length=args.max_len, direction=orc.FORWARD)
            pick = np.random.default_rng(20261004).permutation(n)
            t1 = time.time()
            i = mism = 0
            while i < n and time.time() - t1 < args.cpu_seconds:
                j = int(pick[i])
                r = oe.dfs(seeds[j].tobytes().decode(), [sink[j].tobytes().decode()])
                gi = b.graph(j)
                same = (gi is None) == r.is_null and (r.is_null or (gi.vertex_tuples() == r.vertices() and gi.edge_tuples() == r.edges()))
                mism += 0 if same else 1
                r.free()
                i += 1
            dtc = time.time() - t1
            out["cpu_baseline"] = {"value": oe.kmers_traversed() / dtc, "unit": "k-mers traversed/s", "cores": 1, "kind": "port",
                                   "sample": "%d searches drawn at random (seed 20261004) from the batch (%d k-mers traversed in %.1f s), oracle in faithful mode" % (i, oe.kmers_traversed(), dtc),
                                   "contigs_per_s": i / dtc}
            out["parity"] = "%d/%d sampled dfs graphs bit-exact vs oracle (vertices and edges in insertion order)" % (i - mism, i)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n):
    """python bench.py --gpus N without torchrun: N child processes, one per GPU, RCCL rendezvous on 127.0.0.1"""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = max(rc, abs(p_.wait()))
    sys.exit(rc)


# ---- byte model of the walk kernel (DESIGN.md §4 "What a step must move"): HBM bytes each kind of step REQUIRES, whatever the caches do
WALK_STEP_BYTES = {
    # run step: the two far fringe rows (2 x 64) + their run-index entries (2 x 8) + uo/ubase at the two positions (2 x 5) + three table
    # probe rounds of four slots (y, z, the piece entry: 3 x 32) + five entries written (cv, t, piece, y, z: 5 x 8) + path: t, RUN head,
    # payload, y (4 x 8)
    "run": 2 * 64 + 2 * 8 + 2 * 5 + 3 * 32 + 5 * 8 + 4 * 8,
    # lean step: the next row (64) + its run-index entry (8) + one probe round (32) + the seen mark and the visit count (2 x 8) + one path entry (8)
    "lean": 64 + 8 + 32 + 2 * 8 + 8,
    # general step: what a lean step moves, for the vertex it steps onto (its junction work is counted per link-store element and per choice)
    "general": 64 + 8 + 32 + 2 * 8 + 8,
    # a link-store element: its junction record (20) + the share of the record-range word of its k-mer (8); the store itself lives in LDS
    "add": 20 + 8,
    # a junction choice: the chosen child's row, run-index entry and probe round; junction bases are cached 8 per element (1 byte per 8 positions)
    "choice": 64 + 8 + 32 + 1,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--genome-len", type=int, default=GENOME_LEN)
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--seeds", type=int, default=N_SEEDS)
    ap.add_argument("--max-len", type=int, default=75000)
    ap.add_argument("--repeat-families", type=int, default=4000)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-timed", action="store_true", help="c3: skip the side measurements (single_batch, host_seeds, with_contigs_fetched): for runs under a profiler")
    ap.add_argument("--lookups", type=int, default=100000, help="c2: lookups per step (configs[1] says 100k)")
    ap.add_argument("--sharded", action="store_true", help="hash-shard the table over the ranks: c2 routes lookups with all-to-all, c3 walks over local images of the table")
    ap.add_argument("--sharded-seeds", type=int, default=50000, help="--sharded: seeds per GPU and step")
    ap.add_argument("--in-flight", type=int, default=2, help="c3: engines (own HIP stream + host thread each) that take the steps in turn; 1 = one batch at a time")
    ap.add_argument("--rows-per-owner", type=int, default=65536, help="--sharded: rows one rank may ask of one owner per round")
    ap.add_argument("--check-every", type=int, default=16, help="--sharded: rounds between two looks at the 'anyone still walking' count")
    ap.add_argument("--chain-depth", type=int, default=256, help="--sharded: row slots per request (the row asked for + rows around it its owner holds too); "
                    "the loop iterations a wavefront runs before it ends its round follow it (LDBG_IMG_YIELD, INTEGRATION.md 5) unless the environment sets them: "
                    "profiles/r03_sharded_round_sweep2.log")
    ap.add_argument("--workload", choices=["c3", "c4", "c2"], default="c3",
                    help="c3 (default, the metric's configuration): link-guided contig walks; c4: DestinationStopper dfs to a sink 200-2000 bp downstream")
    ap.add_argument("--stopper", default="DestinationStopper", help="c4: the stopping rule of the searches (a rule that never fails, like ExplorationStopper, returns every branch it explored: use a small --max-len with it)")
    ap.add_argument("--use-seeds", type=int, default=0, help="experiment: walk only the first N seeds")
    ap.add_argument("--no-links", action="store_true", help="experiment: walk without the link annotations")
    ap.add_argument("--no-strict", action="store_true", help="experiment: CanonicalKmer.isFlipped by comparison (not Java-exact, Q6)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for N GPUs but not started by a launcher: start one fresh child process per GPU (nothing in THIS process has touched
        # the GPU yet), rank 0 prints the line; never run one rank and call it N
        return spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher and the flag disagree" % (args.gpus, world))
    import numpy as np
    import torch
    dist = None
    if world > 1 or args.sharded or os.environ.get("LDBG_FORCE_DIST"):      # LDBG_FORCE_DIST: rehearse the N > 1 code path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import corticall_amd as ca
    from corticall_amd import BOTH, OR, ContigStopper, CortexGraph, CortexLinks, TraversalEngineFactory

    if args.workload == "c2":
        return bench_c2(args, ca, rank, local_rank, world, dist)

    # same graph on every rank: rank 0 generates the files (once), the others wait for them
    if dist is not None and rank != 0:
        dist.barrier()
    # weak scaling: world x 50,000 DISTINCT seeds are drawn from the one graph and dealt out, rank r walks seeds r, r + world, ...
    # (the de novo k-mers are a fixed set of ~21,600, so with N ranks each rank's share holds 1/N of them and more random child k-mers)
    prefix, st = workload_files(args, 0, n_seeds=args.seeds * max(1, world))
    if dist is not None and rank == 0:
        dist.barrier()
    seeds = np.fromfile(prefix + ".seeds", dtype=np.uint8).reshape(-1, args.k)
    if world > 1:
        seeds = np.ascontiguousarray(seeds[rank::world])
    if args.use_seeds:
        seeds = seeds[np.random.default_rng(7).permutation(len(seeds))[:args.use_seeds]]

    if args.sharded:
        os.environ.setdefault("LDBG_IMG_YIELD", str(max(32, args.chain_depth)))
    if args.workload == "c3" and args.sharded:
        return bench_c3_sharded(args, ca, prefix, st, seeds, rank, local_rank, world, dist)

    t_load = time.time()
    g = CortexGraph(prefix + ".ctx", device=local_rank)
    links = CortexLinks(prefix + ".ctp.gz", g)
    eng = (TraversalEngineFactory().traversalColors(g.getColorForSampleName("child")).traversalDirection(BOTH)
           .combinationOperator(OR).stoppingRule(ContigStopper).maxBranchLength(args.max_len).graph(g).strictJavaFlip(not args.no_strict))
    if not args.no_links:
        eng.links(links)
    eng = eng.make()
    t_load = time.time() - t_load

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if args.workload == "c4" and args.sharded:
        return bench_c4_sharded(args, ca, g, links, eng, seeds, st, prefix, rank, local_rank, world, dist, sync)
    if args.workload == "c4":
        return bench_c4(args, ca, g, links, eng, seeds, st, prefix, rank, world, dist, sync, t_load)

    # the FIRST batch of an engine builds its run index (the records in unitig order): a caller that runs one batch per engine pays it
    # the step's input resident in HBM before the timed region (the contract's `value`): the seeds as an (n, k) uint8 device array, used
    # where they are by ldbg_engine_walk_batch_run_device.  The same steps with the seeds handed over as a host buffer are timed below
    # (`host_seeds`), and with every contig brought back as well (`with_contigs_fetched`).
    d_seeds = torch.from_numpy(np.ascontiguousarray(seeds)).to("cuda:%d" % local_rank)
    sync()
    first_batch_ms = None
    if args.warmup >= 1:
        t_first = time.time()
        eng.walk_batch_arrays(d_seeds, fetch=False)
        sync()
        first_batch_ms = (time.time() - t_first) * 1e3
    run_index_ms, _ = ca.profile_get("run_index")
    # --in-flight N: N engines on the one graph, each with its own HIP stream and host thread, take the steps in turn.  A walk launch lasts
    # as long as its longest strands and most of its wavefronts are done long before: the next batch of another engine fills
    # the compute units they leave.  Every step is still one whole walk_batch over all the seeds of the rank.
    engines = [eng]
    for _ in range(max(1, args.in_flight) - 1):
        f2 = (TraversalEngineFactory().traversalColors(g.getColorForSampleName("child")).traversalDirection(BOTH).combinationOperator(OR)
              .stoppingRule(ContigStopper).maxBranchLength(args.max_len).graph(g).strictJavaFlip(not args.no_strict))
        if not args.no_links:
            f2.links(links)
        engines.append(f2.make())
    for e2 in engines[1:]:
        e2.walk_batch_arrays(d_seeds, fetch=False)          # (each engine builds its run index and pools with its first batch)
    for _ in range(max(0, args.warmup - 1)):
        for e2 in engines:
            e2.walk_batch_arrays(d_seeds, fetch=False)

    def run_steps(engs, n_steps):
        """n_steps walk batches, dealt out to the engines in turn; one host thread per engine (ctypes drops the GIL during the call)"""
        import threading
        trav = [0] * len(engs)

        def work(i):
            for _ in range(i, n_steps, len(engs)):
                engs[i].walk_batch_arrays(d_seeds, fetch=False)      # results stay in HBM (contigs, offsets, vertex lists)
                trav[i] += engs[i].kmers_traversed
        if len(engs) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        return sum(trav)

    # one batch at a time first (the latency of a step, the kernel durations without a neighbour): reported as `single_batch`
    # (--only-timed: profiling runs — the kernel trace of the process is then the timed region's launches and little else)
    side = 0 if args.only_timed else args.steps
    ca.profile_reset()
    sync()
    t0 = time.time()
    run_steps(engines[:1], side)
    sync()
    dt_single = time.time() - t0
    single_walk_ms, single_launches = ca.profile_get("walk")
    single_contig_ms, _ = ca.profile_get("contig")
    # the timed region of `value`
    ca.profile_reset()
    sync()
    t0 = time.time()
    traversed = run_steps(engines, args.steps)
    sync()
    dt = time.time() - t0
    walk_ms, walk_launches = ca.profile_get("walk")
    contig_ms, _ = ca.profile_get("contig")
    kinds = {nm: ca.profile_get("walk_" + nm)[0] / max(1, walk_launches) for nm in
             ("steps_run", "run_vertices", "steps_lean", "steps_general", "link_adds", "choices", "wave_iterations", "wave_general",
              "busiest_general", "busiest_iterations", "wavefronts")}
    # the same steps with the seeds handed over in host memory (n x k ASCII bytes of an ordinary numpy array)
    sync()
    t1 = time.time()
    for _ in range(side):
        eng.walk_batch_arrays(seeds, fetch=False)
    sync()
    dt_host_seeds = time.time() - t1
    # the same steps with every contig downloaded to the caller (what the reference's walk() hands over): reported beside `value`
    if side:
        eng.walk_batch_arrays(seeds, fetch=True, pinned=True)        # (the page-locked arena is allocated once, like every other buffer of the engine)
    sync()
    t1 = time.time()
    fetched_bytes = 0
    for _ in range(side):
        arena, _, _ = eng.walk_batch_arrays(seeds, fetch=True, pinned=True)
        fetched_bytes += len(arena)
    sync()
    dt_fetch = time.time() - t1
    t1 = time.time()
    for _ in range(min(3, side)):
        eng.walk_batch_arrays(seeds, fetch=True)               # into a fresh pageable array: staged through page-locked buffers by the library
    sync()
    dt_fetch_pageable = (time.time() - t1) / max(1, min(3, side))

    tot_trav, tot_seeds, max_dt = traversed, len(seeds) * args.steps, dt
    if dist is not None:
        t = torch.tensor([float(traversed), float(len(seeds) * args.steps)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        m = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        tot_trav, tot_seeds, max_dt = int(t[0].item()), int(t[1].item()), m[0].item()

    if rank == 0:
        N, W, C = g.getNumRecords(), g.getKmerBits(), g.getNumColors()
        # algorithmic bytes per k-mer traversed: one findRecord of the successor
        # = ceil(log2 N) keys of 8W bytes + the record's 5C payload bytes, plus one link-table lookup
        M = max(2, links.numKmersWithLinks)
        b_find = math.ceil(math.log2(N)) * 8 * W + 5 * C
        b_link = math.ceil(math.log2(M)) * 8 * W
        per_launch_units = traversed / max(1, walk_launches)
        avg_ms = walk_ms / max(1, walk_launches)
        achieved = per_launch_units * (b_find + b_link) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        if run_index_ms == 0.0:
            run_index_ms, _ = ca.profile_get("run_index")          # (--warmup 0: the build fell into the timed region)
        # HBM bytes per launch from the PMC passes of tools/profile_walk.sh (FETCH_SIZE, WRITE_SIZE): reported only when they were taken on
        # this very build of the library (the digest in ldbg_version), else null
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r03_walk_traffic.json")
        if os.path.exists(tf):
            tj = json.load(open(tf))
            if tj.get("library") == ca.default_lib().dll.ldbg_version().decode():
                traffic = tj.get("hbm_bytes_per_launch")
        measured = traffic / (avg_ms * 1e-3) / 1e9 if traffic and avg_ms > 0 else None
        # the model: bytes the steps of THIS launch required (step-kind counters of the kernel x WALK_STEP_BYTES)
        B = WALK_STEP_BYTES
        model_bytes = (kinds["steps_run"] * B["run"] + kinds["steps_lean"] * B["lean"] + kinds["steps_general"] * B["general"]
                       + kinds["link_adds"] * B["add"] + kinds["choices"] * B["choice"])
        model_gbs = model_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        contig_bytes = 2.0 * eng.last_total_bytes               # k_contigs_rle: one byte read (ubase) and one written per contig base
        contig_ms_step = contig_ms / max(1, args.steps)
        out = {
            "metric": "k-mers traversed/sec (whole node) + contigs/sec, k=47 3-color LdBG",
            "value": tot_trav / max_dt, "unit": "k-mers traversed/s", "contigs_per_s": tot_seeds / max_dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": max_dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "library": ca.default_lib().dll.ldbg_version().decode(),
            "config": {
                "workload": "configs[2]: synthetic %.1f Mb 3-colour k=%d LdBG with child links (read 250 bp / stride 8), "
                            "link-guided ContigStopper walks BOTH/OR from %d seeds per GPU (%d de novo), maxLength %d"
                            % (args.genome_len / 1e6, args.k, len(seeds), st["n_novel_seeds"], args.max_len),
                "records": N, "record_bytes": 8 * W + 5 * C, "link_kmers": links.numKmersWithLinks, "links": links.numLinks,
                "seeds_per_gpu": len(seeds), "kmers_traversed_per_step": traversed // args.steps,
                "multi_gpu": "replicated graph, seeds partitioned, no data-path collective" if world > 1 else "single GPU",
                "load_seconds": round(t_load, 2), "run_index_build_ms": run_index_ms, "first_batch_ms": first_batch_ms,
                "inputs": "graph, links and seeds resident in HBM before the timed region (ldbg_engine_walk_batch_run_device); results stay in HBM",
                "batches_in_flight": len(engines),
                "schedule": ("%d engines on the one graph (own HIP stream and host thread each) take the %d steps in turn; every step is one whole "
                             "walk_batch over all the seeds" % (len(engines), args.steps)) if len(engines) > 1 else "one batch at a time",
            },
            "single_batch": {"ms_per_step": dt_single / args.steps * 1e3, "value": traversed / dt_single if dt_single > 0 else None,
                             "k_walk_ms": single_walk_ms / max(1, single_launches), "k_contigs_rle_ms": single_contig_ms / max(1, args.steps),
                             "note": "the same steps one batch at a time on one engine: the latency of a step and the kernel durations without a neighbour"},
            "roofline": {
                "bound": "hbm", "kernel": "k_walk<%d>" % W, "achieved": model_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": model_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": avg_ms, "launches": walk_launches,
                "aggregate": {"achieved": model_bytes * walk_launches / dt / 1e9 if dt > 0 else None,
                              "frac": model_bytes * walk_launches / dt / 1e9 / HBM_PEAK_GBS if dt > 0 else None,
                              "note": "the model bytes of ALL launches of the timed region / its wall time: with engines in flight launches overlap, and "
                                      "each waits for compute units inside its own HIP events (`achieved` / `frac` above are per launch, as measured)"},
                "model": {
                    "bytes_per_launch": model_bytes, "bytes_per_step_kind": B,
                    "per_launch": {"run_steps": kinds["steps_run"], "vertices_crossed_by_run_steps": kinds["run_vertices"], "lean_steps": kinds["steps_lean"],
                                   "general_steps": kinds["steps_general"], "link_store_elements_added": kinds["link_adds"], "junction_choices": kinds["choices"]},
                    "note": "achieved = sum over step kinds of (steps counted by the kernel in this run x HBM bytes that kind of step must move, "
                            "DESIGN.md 4) / launch time: a bandwidth fraction of the kernel that runs (run index: an unbranched stretch is ONE step)",
                },
                "latency_bound": {
                    "wavefronts": kinds["wavefronts"], "loop_iterations_avg": kinds["wave_iterations"] / max(1.0, kinds["wavefronts"]),
                    "with_general_part_avg": kinds["wave_general"] / max(1.0, kinds["wavefronts"]),
                    "busiest_wavefront_iterations": kinds["busiest_iterations"], "busiest_wavefront_with_general_part": kinds["busiest_general"],
                    "us_per_iteration_of_the_busiest_wavefront": avg_ms * 1e3 / kinds["busiest_iterations"] if kinds["busiest_iterations"] else None,
                    "note": "the launch lasts as long as its busiest wavefront: dependent steps x their latency, not bytes, bound it",
                },
                "frac_reference_algorithm": achieved / HBM_PEAK_GBS,
                "reference_algorithm_note": "k-mers traversed x the REFERENCE's per-k-mer search bytes (SURVEY 8d: %d B) / launch time / peak: what a "
                                            "k-mer-by-k-mer binary-search walk would have to sustain to match this launch; above 1 because the run index "
                                            "removes those searches — a speed-up figure, NOT a bandwidth fraction" % (b_find + b_link),
                "measured_gbs": measured, "measured_frac": measured / HBM_PEAK_GBS if measured else None,
                "contig_kernel": {"kernel": "k_contigs_rle<%d>" % W, "ms_per_step": contig_ms_step, "bytes_per_step": contig_bytes,
                                  "achieved": contig_bytes / (contig_ms_step * 1e-3) / 1e9 if contig_ms_step > 0 else None,
                                  "frac": contig_bytes / (contig_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS if contig_ms_step > 0 else None},
                "whole_step_floor": {"bytes": contig_bytes, "ms_at_peak": contig_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
                                     "frac": contig_bytes / (HBM_PEAK_GBS * 1e9) / (max_dt / args.steps) if max_dt > 0 else None,
                                     "note": "a step must at least read one base byte and write one contig byte per k-mer"},
            },
            "host_seeds": {"value": tot_trav / max_dt * dt / dt_host_seeds if dt_host_seeds > 0 else None, "ms_per_step": dt_host_seeds / args.steps * 1e3,
                           "note": "rank 0's steps again, one batch at a time, with the seeds handed over as a host buffer (pageable numpy array, %d bytes per step over "
                                   "PCIe) instead of resident in HBM; results stay in HBM as for `value`" % seeds.nbytes},
            "with_contigs_fetched": {"value": tot_trav / max_dt * dt / dt_fetch if dt_fetch > 0 else None, "ms_per_step": dt_fetch / args.steps * 1e3,
                                     "bytes_per_step": fetched_bytes // max(1, args.steps),
                                     "note": "rank 0's steps again, one batch at a time, host seeds in and all contigs downloaded into the engine's page-locked arena (ldbg_host_alloc): "
                                             "what a host that hands over strings and reads strings sees",
                                     "pageable_ms_per_step": dt_fetch_pageable * 1e3,
                                     "pageable_note": "the same into a fresh pageable array (staged through the library's page-locked buffers)"},
        }
        if args.only_timed:
            for key in ("single_batch", "host_seeds", "with_contigs_fetched"):
                out[key] = {"skipped": "--only-timed"}
        if not args.no_cpu_baseline:
            pick = np.random.default_rng(20261004).choice(len(seeds), min(2000, len(seeds)), replace=False)
            contigs, _ = eng.walk_batch(seeds[pick])
            base, n_cmp, mism = cpu_baseline(prefix, args, contigs, seeds[pick], len(seeds))
            out["cpu_baseline"] = base
            out["cpu_baseline_tuned_all_cores"] = cpu_baseline_tuned(prefix, args, seeds)
            out["parity"] = "%d/%d sampled contigs bit-exact vs oracle" % (n_cmp - mism, n_cmp)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
