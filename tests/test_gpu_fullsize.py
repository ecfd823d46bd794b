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
    #     tests/test_oracle_golden.py, but fast enough for long walks)
    og = orc.Graph(workload + ".ctx", tuned=True)
    ol = orc.Links(workload + ".ctp.gz")
    oe = orc.Engine(og, [0], links=[ol], stopper="ContigStopper")
    t0, checked = time.time(), 0
    for i in sample:
        if time.time() - t0 > 60:
            break
        exp, nv = oe.walk(seeds[i].tobytes().decode())
        assert raw[offs[i]:offs[i + 1]].decode() == exp and wl[i] == nv
        checked += 1
    assert checked >= 20
    g.close()
