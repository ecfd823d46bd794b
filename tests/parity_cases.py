"""Parity cases shared by the CPU host-simulation run (tests/test_hostsim_parity.py, `-m "not gpu"`)
and the real-hardware run (tests/test_gpu_parity.py, `-m gpu`).  Every case drives the product API
(corticall_amd over the C ABI) and compares bit-exactly with the CPU oracle on the same inputs.
Cases named test_ref_* restate the reference's own tests
(T/ = public/java/tests/uk/ac/ox/well/cortexjdk/)."""
import os
import random

import numpy as np
import pytest

import corticall_amd as ca
from corticall_amd._native import JavaNullPointerException
from corticall_amd import (AND, BOTH, FORWARD, OR, REVERSE, ContigStopper, CortexGraph, CortexLinks,
                           TraversalEngineFactory, TraversalUtils)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_seq(rng, n, gc=0.5):
    return "".join(rng.choice("GC") if rng.random() < gc else rng.choice("AT") for _ in range(n))


def mutate(rng, s, snv=0.01, indel=0.002):
    out = []
    i = 0
    while i < len(s):
        r = rng.random()
        if r < snv:
            out.append(rng.choice([b for b in "ACGT" if b != s[i]]))
        elif r < snv + indel:
            if rng.random() < 0.5:
                out.append(s[i] + rand_seq(rng, rng.randint(1, 4)))
            # else: deletion
        else:
            out.append(s[i])
        i += 1
    return "".join(out)


def genome_with_repeats(rng, n, n_rep=6, rep_len=(8, 60), copies=(2, 4), gc=0.5):
    s = rand_seq(rng, n, gc)
    for _ in range(n_rep):
        L = rng.randint(*rep_len)
        rep = rand_seq(rng, L, gc)
        for _ in range(rng.randint(*copies)):
            p = rng.randint(0, len(s))
            s = s[:p] + rep + s[p:]
    # a tandem repeat (forces cycles)
    unit = rand_seq(rng, rng.randint(3, 12), gc)
    p = rng.randint(0, len(s))
    s = s[:p] + unit * rng.randint(3, 6) + s[p:]
    return s


class Case:
    """a small multi-colour graph (+ optional links) built with the oracle's fixture tools"""

    def __init__(self, orc, tmp, lib, haps, k, link_samples=(), reads=None, name="g"):
        self.orc, self.lib, self.k = orc, lib, k
        self.path = str(tmp / (name + ".ctx"))
        orc.build_graph(self.path, haps, k)
        self.og = orc.Graph(self.path, tuned=True)
        self.g = CortexGraph(self.path, lib=lib)
        self.haps = dict(haps)
        self.olinks, self.links = {}, {}
        for s in link_samples:
            lp = str(tmp / (name + "." + s + ".ctp.gz"))
            orc.build_links(self.og, lp, s, (reads or self.haps)[s])
            self.olinks[s] = orc.Links(lp)
            self.links[s] = CortexLinks(lp, self.g)

    def all_kmers(self):
        return [self.og.record_string(i).split()[0] for i in range(self.og.N)]

    def engines(self, trav, links=(), recruit=(), op=OR, direction=BOTH, max_len=75000):
        oe = self.orc.Engine(self.og, trav, links=[self.olinks[s] for s in links], recruitment_colors=recruit,
                             op_and=(op == AND), direction=direction, max_length=max_len, stopper="ContigStopper")
        f = (TraversalEngineFactory(lib=self.lib).traversalColors(*trav).graph(self.g).combinationOperator(op)
             .traversalDirection(direction).maxBranchLength(max_len).stoppingRule(ContigStopper))
        if recruit:
            f.recruitmentColors(*recruit)
        if links:
            f.links(*[self.links[s] for s in links])
        return oe, f.make()


class _HostSeedsAsDevice:
    """the host simulation's "device memory" is host memory: an (n, k) uint8 array dressed as the device tensor walk_batch_arrays accepts"""
    is_cuda = True

    def __init__(self, a):
        self.a = np.ascontiguousarray(a, dtype=np.uint8)
        self.shape = self.a.shape

    def data_ptr(self): return self.a.ctypes.data
    def dim(self): return self.a.ndim
    def element_size(self): return 1
    def is_contiguous(self): return True


def device_seeds(lib, seeds, k):
    """the seeds as ldbg_engine_walk_batch_run_device takes them: an (n, k) uint8 array in the memory of the library's device"""
    a = np.frombuffer("".join(seeds).encode(), dtype=np.uint8).reshape(len(seeds), k)
    if lib.is_hostsim:
        return _HostSeedsAsDevice(a)
    return _HipSeeds(a)


class _HipSeeds:
    """(n, k) uint8 seeds in device memory through the HIP runtime itself (hipMalloc + a synchronous hipMemcpy): no torch needed — and
    importing torch AFTER libldbg has initialised the runtime finds no GPU, which __graft_entry__.smoke() would do"""
    is_cuda = True
    _hip = None

    def __init__(self, a):
        import ctypes as C
        if _HipSeeds._hip is None:
            _HipSeeds._hip = C.CDLL("libamdhip64.so.7")      # by soname: the instance libldbg.so (and torch, if imported) already runs on
        hip = _HipSeeds._hip
        a = np.ascontiguousarray(a, dtype=np.uint8)
        self.shape = a.shape
        self._n = a.ndim
        self._p = C.c_void_p()
        assert hip.hipMalloc(C.byref(self._p), C.c_size_t(max(1, a.nbytes))) == 0
        assert hip.hipMemcpy(self._p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.c_int(1)) == 0      # hipMemcpyHostToDevice

    def __del__(self):
        if _HipSeeds._hip is not None and self._p:
            _HipSeeds._hip.hipFree(self._p)

    def data_ptr(self): return self._p.value
    def dim(self): return self._n
    def element_size(self): return 1
    def is_contiguous(self): return True


def compare_walks(case, seeds, **cfg):
    oe, e = case.engines(**cfg)
    seeds = list(seeds)
    got, wl = e.walk_batch(seeds)
    if seeds:       # the same batch with the seeds already in device memory: the same contigs
        got_d, wl_d = e.walk_batch(device_seeds(case.lib, seeds, case.k))
        assert got_d == got and (wl_d == wl).all()
    km = np.frombuffer("".join(seeds).encode(), dtype=np.uint8).reshape(len(seeds), case.k)
    arena, offs, nv = oe.walk_batch(km)
    raw = arena.tobytes()
    exp = [raw[offs[i]:offs[i + 1]].decode() for i in range(len(seeds))]
    for i, s in enumerate(seeds):
        assert got[i] == exp[i], (s, cfg, got[i], exp[i])
    assert (wl == nv).all()
    assert e.kmers_traversed == oe.kmers_traversed(), (e.kmers_traversed, oe.kmers_traversed())
    return exp


# ------------------------------------------------------------------ graph / records / find
def case_fixture_graph(orc, lib, tmp):
    g = CortexGraph(os.path.join(GOLDEN, "two_short_contigs.ctx"), lib=lib)
    rows = [l.split() for l in open(os.path.join(GOLDEN, "two_short_contigs.expected.txt"))]
    assert (g.getKmerSize(), g.getKmerBits(), g.getNumColors(), g.getNumRecords(), g.getVersion()) == (31, 1, 2, 66, 6)
    assert g.getSampleName(0) == "one" and g.getSampleName(1) == "two"            # CortexGraphTest.java:140-145
    assert g.getColorForSampleName("two") == 1 and g.getColorForSampleName("nope") == -1
    recs = [cr.toString() for cr in g]                                             # :187-198
    assert recs == [" ".join(r) for r in rows]
    for i in range(10, -1, -1):                                                    # :256-265
        assert g.getRecord(i).toString() == " ".join(rows[i])
    assert g.getRecord(66) is None
    for i, r in enumerate(rows):                                                   # :311-320
        cr = g.findRecord(r[0])
        assert cr.toString() == " ".join(r) and cr.index == i
        assert g.findRecord(orc.revcomp(r[0])).index == i
    assert g.findRecord("NTTTTGGGGTATTTGCAGTATTTGGAATAAA") is None                 # :323-331
    assert g.findRecord("A" * 31) is None
    w, c, e = g.records(0, 66)
    for i, r in enumerate(rows):
        assert [int(x) for x in w[i]] == orc.encode_kmer(r[0])
    # BASELINE configs[0]: full-graph iterate (above) + one walk — from record 0's k-mer in colour 0, and from every other record
    # in both colours, against the oracle on the same file
    og = orc.Graph(os.path.join(GOLDEN, "two_short_contigs.ctx"))
    for colour in (0, 1):
        oe = orc.Engine(og, [colour], stopper="ContigStopper")
        e = TraversalEngineFactory(lib=lib).traversalColors(colour).graph(g).stoppingRule(ContigStopper).make()
        seeds = [r[0] for r in rows] + [orc.revcomp(r[0]) for r in rows[:8]]
        got, wl = e.walk_batch(seeds)
        for sd, c, n in zip(seeds, got, wl):
            exp, nv = oe.walk(sd)
            assert c == exp and n == nv, (colour, sd, c, exp)
        if colour == 0:
            assert TraversalUtils.toContig(e.walk(rows[0][0])) == got[0] and len(got[0]) >= 31
    g.close()


def case_random_find(orc, lib, tmp, k, ncol, n=3000, seed=1):
    rng = random.Random(seed * 1000 + k)
    base = genome_with_repeats(rng, n)
    haps = [("s%d" % c, [base if c == 0 else mutate(rng, base)]) for c in range(ncol)]
    cs = Case(orc, tmp, lib, haps, k, name="f%d_%d" % (k, ncol))
    kmers = cs.all_kmers()
    q = []
    for _ in range(2000):
        r = rng.random()
        if r < 0.4:
            s = rng.choice(kmers)
            q.append(s if rng.random() < 0.5 else orc.revcomp(s))
        elif r < 0.8:
            q.append(rand_seq(rng, k))
        elif r < 0.9:
            s = list(rng.choice(kmers)); s[rng.randrange(k)] = rng.choice("ACGT"); q.append("".join(s))
        else:
            s = list(rng.choice(kmers)); s[rng.randrange(k)] = "N"; q.append("".join(s))
    idx, cov, edges = cs.g.find_batch(q)
    km = np.frombuffer("".join(q).encode(), dtype=np.uint8).reshape(len(q), k)
    exp = cs.og.find_batch(km, tuned=False)
    assert (idx == exp).all()
    w, c, e = cs.g.records(0, cs.g.getNumRecords())
    for i in range(len(q)):
        if idx[i] >= 0:
            assert (cov[i] == c[idx[i]]).all() and (edges[i] == e[idx[i]]).all()
        else:
            assert not cov[i].any() and not edges[i].any()
    # iteration == oracle records
    for i in range(0, cs.og.N, max(1, cs.og.N // 50)):
        ow, oc, oe_ = cs.og.get_record(i)
        assert [int(x) for x in w[i]] == ow and list(c[i]) == oc and list(e[i]) == oe_


def case_all_bits_kmers(orc, lib, tmp, k):
    """k = 32, 64, 96: every bit pattern of the packed words is a k-mer, so "not a k-mer" (a string with an N, quirk Q4) must travel
    beside the words.  Poly-T stretches make the all-ones words a real k-mer of the graph."""
    rng = random.Random(k)
    hap = rand_seq(rng, 150) + "T" * (k + 40) + rand_seq(rng, 150) + "A" * (k + 3) + rand_seq(rng, 80)
    cs = Case(orc, tmp, lib, [("a", [hap])], k, link_samples=["a"], name="ones%d" % k)
    polyt = "T" * k
    seeds = [polyt, "A" * k, hap[150 - 9:150 - 9 + k], hap[150 + 5:150 + 5 + k], "T" * 32 + hap[150 + k + 8:150 + k + 8 + k - 32] if k > 32 else polyt,
             "N" + "T" * (k - 1), "T" * (k - 1) + "N", "N" * k, hap[:k], orc.revcomp(hap[200 + k:200 + 2 * k])]
    seeds = [s for s in seeds if len(s) == k]
    idx, cov, edges = cs.g.find_batch(seeds)
    for s_, i in zip(seeds, idx):
        exp = cs.og.find(s_)[0]
        assert exp == int(i), (k, s_, exp, int(i))
    assert idx[0] >= 0 and idx[5] == -1 and idx[6] == -1 and idx[7] == -1
    assert cs.g.findRecord("N" + "T" * (k - 1)) is None and cs.g.findRecord(polyt) is not None
    compare_walks(cs, seeds, trav=[0])
    compare_walks(cs, seeds, trav=[0], links=["a"], max_len=300)
    for stopper in ("ContigStopper", "DestinationStopper", "ExplorationStopper"):
        compare_dfs(cs, seeds[:8], sinks=[[polyt], ["N" * k], [hap[160:160 + k]], ["N" + "T" * (k - 1)], [polyt], [], [polyt, "N" * k], ["A" * k]],
                    trav=[0], stopper=stopper, links=["a"], max_len=200)
    oe, e = cs.engines(trav=[0], links=["a"])
    for sd in ("N" + "T" * (k - 1), polyt):
        e.seek(sd)
        oe.seek(sd)
        assert e.hasNext() == oe.has_next() and e.hasPrevious() == oe.has_previous()


def case_q1_tiny(orc, lib, tmp):
    p = str(tmp / "tiny.ctx")
    orc.build_graph(p, [("s", ["ACGTT"])], 4)
    g = CortexGraph(p, lib=lib)
    assert g.getNumRecords() == 2
    assert g.findRecord("ACGT") is None and g.findRecord("AACG") is None      # Q1
    assert g.getRecord(0).getKmerAsString() == "AACG"


def case_unsorted_rejected(orc, lib, tmp):
    src = open(os.path.join(GOLDEN, "two_short_contigs.ctx"), "rb").read()
    off, rs = 148, 18
    recs = [src[off + i * rs: off + (i + 1) * rs] for i in range(66)]
    recs[10], recs[40] = recs[40], recs[10]
    p = str(tmp / "unsorted.ctx")
    open(p, "wb").write(src[:off] + b"".join(recs))
    try:
        CortexGraph(p, lib=lib)
        raise AssertionError("unsorted graph accepted")
    except ca.CortexJDKException as ex:
        assert "Records are not sorted" in str(ex)
    open(p, "wb").write(b"NOTCTX" + src[6:])
    try:
        CortexGraph(p, lib=lib)
        raise AssertionError("bad magic accepted")
    except ca.CortexJDKException as ex:
        assert "does not appear to be a Cortex graph" in str(ex)
    open(p, "wb").write(b"cortex" + src[6:])          # the reference compares the magic word ignoring case (CortexGraph.java:96)
    CortexGraph(p, lib=lib).close()


def case_record_count_guard(orc, lib, tmp, monkeypatch):
    """a table beyond what the 31-bit record numbers of the device structures can hold is refused at open (the limit itself is
    2^31 - 2 records; LDBG_MAX_RECORDS lowers it so that a 66-record file can stand in for a 2-billion-record one)"""
    fx = os.path.join(GOLDEN, "two_short_contigs.ctx")
    monkeypatch.setenv("LDBG_MAX_RECORDS", "65")
    try:
        CortexGraph(fx, lib=lib)
        raise AssertionError("a table over the record limit was opened")
    except ca.LdbgError as ex:
        assert ex.status == 4 and "holds 66 records" in str(ex) and "at most 65" in str(ex)
    monkeypatch.setenv("LDBG_MAX_RECORDS", "66")
    CortexGraph(fx, lib=lib).close()
    monkeypatch.delenv("LDBG_MAX_RECORDS")
    CortexGraph(fx, lib=lib).close()


def case_rejected_open_frees_device_memory(lib, tmp):
    """(GPU) opening an unsorted file again and again leaves the free device memory where it was"""
    import torch
    src = open(os.path.join(GOLDEN, "two_short_contigs.ctx"), "rb").read()
    off, rs = 148, 18
    recs = [src[off + i * rs: off + (i + 1) * rs] for i in range(66)] * 1      # small file; what leaks is per open (buffers + a stream)
    big = recs * 40000                                                           # 2.6 M records, unsorted by construction: ~50 MB of rows
    p = str(tmp / "unsorted_big.ctx")
    open(p, "wb").write(src[:off] + b"".join(big))
    free0 = None
    for it in range(6):
        try:
            CortexGraph(p, lib=lib)
            raise AssertionError("unsorted graph accepted")
        except ca.CortexJDKException:
            pass
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        if it == 1:
            free0 = free
    assert free0 - free < 32 << 20, (free0, free)


# ------------------------------------------------------------------ reference tests through the product API
def test_ref_short_contig_reconstruction(orc, lib, tmp):     # T/utils/traversal/TraversalEngineTest.java:98-122
    cs = Case(orc, tmp, lib, [("mom", ["AGTTCTGATCTGGGCTATATGCT"]), ("dad", ["AGTTCGAATCTGGGCTATATGCT"]),
                              ("kid", ["AGTTCTGATCTGGGCTATGGCTA"])], 5)
    exp = {"mom": "AGTTCTGATCTGGGCTATATGCT", "dad": "TTCGAATCTGGGCTATATGCT", "kid": "AGTTCTGATCTGGGCTATGGCT"}
    for c in range(3):
        e = TraversalEngineFactory(lib=lib).traversalColors(c).graph(cs.g).stoppingRule(ContigStopper).make()
        assert TraversalUtils.toContig(e.walk("CTGGG")) == exp[cs.g.getSampleName(c)]


def test_ref_recruitment(orc, lib, tmp):                     # TraversalEngineTest.java:125-157
    h = "AGTTCTGATCTGGGCTATATGCT"
    cs = Case(orc, tmp, lib, [("mom", [h]), ("dad", [h]), ("kid", ["AGTTCTG", "ATGGCTA"])], 5)
    g = cs.g
    f = (TraversalEngineFactory(lib=lib).traversalColors(g.getColorForSampleName("kid")).combinationOperator(AND)
         .traversalDirection(BOTH).connectAllNeighbors(False).stoppingRule(ContigStopper).graph(g))
    er = f.recruitmentColors(g.getColorsForSampleNames(["mom", "dad"])).make()
    assert TraversalUtils.toContig(er.walk("GTTCT")) == h
    er = f.recruitmentColors().make()
    assert TraversalUtils.toContig(er.walk("GTTCT")) == "AGTTCTG"


FIG1, FIG1_READ = "ACTGATTTCGATGCGATGCGATGCCACGGTGG", "TTTCGATGCGATGCGATGCCACG"


def test_ref_cycles_without_and_with_links(orc, lib, tmp):   # TraversalEngineTest.java:210-250
    cs = Case(orc, tmp, lib, [("test", [FIG1])], 5, link_samples=["test"], reads={"test": [FIG1_READ]})
    e = TraversalEngineFactory(lib=lib).traversalColors(0).stoppingRule(ContigStopper).graph(cs.g).make()
    assert TraversalUtils.toContig(e.walk("ACTGA")) == "ACTGATTTCGATGC"
    l = cs.links["test"]
    assert (l.version, l.numColors, l.kmerSize, l.numKmersInGraph, l.numKmersWithLinks, l.numLinks) == (4, 1, 5, 21, 4, 6)   # CortexLinksTest.java:32-51
    assert l.get("ATCGC")[1] == [(j[0] == "F", len(j[1]), [1], j[1]) for j in dict(cs.olinks["test"].records())["ATCGC"]]
    assert l.containsKey("GCGAT") and not l.containsKey("AAAAA")
    e = TraversalEngineFactory(lib=lib).traversalColors(0).stoppingRule(ContigStopper).graph(cs.g).links(l).make()
    w = e.walk("ACTGA")
    assert TraversalUtils.toContig(w) == FIG1
    assert max(v.getCopyIndex() for v in w) >= 1      # the cycle is traversed through copies of its vertices


def test_ref_iterate_fwd_rev(orc, lib, tmp):                  # TraversalEngineTest.java:253-358
    hap = "AGTTCGAATCTGGGCTATATGCT"
    cs = Case(orc, tmp, lib, [("mom", [hap])], 7)
    e = TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).make()
    sk = "AGTTCGA"; sb = sk
    e.seek(sk)
    while e.hasNext():
        sb += e.next().getKmerAsString()[-1]
    assert sb == hap
    sk = "ATATGCT"; sb = sk
    e.seek(sk)
    while e.hasPrevious():
        sb = e.previous().getKmerAsString()[0] + sb
    assert sb == hap
    cs = Case(orc, tmp, lib, [("kid", ["AGTTCGAATCTGGGCTATATGCT", "AGTTCGAATCTGAGCTATATGCT"])], 7, name="fork")
    e = TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).make()
    sb = "AGTTCGA"
    e.seek(sb)
    while e.hasNext():
        sb += e.next().getKmerAsString()[-1]
    assert sb == "AGTTCGAATCTG"
    sb = "ATATGCT"
    e.seek(sb)
    while e.hasPrevious():
        sb = e.previous().getKmerAsString()[0] + sb
    assert sb == "GCTATATGCT"
    try:
        e.previous()
        raise AssertionError("previous() past the fork must throw")
    except ca.NoSuchElementException:
        pass


def test_ref_go_forward_and_backward(orc, lib, tmp):          # TraversalEngineTest.java:361-386
    hap, k = "AGTTCGAATCTGAGCTATATGCT", 7
    cs = Case(orc, tmp, lib, [("kid", [hap])], k)
    e = TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).make()
    n = 0
    for i in range(1, len(hap) - k):
        sk = hap[i:i + k]
        e.seek(sk)
        if e.hasPrevious() and e.hasNext():
            e.next()
            assert e.previous().getKmerAsString() == sk
            n += 1
    assert n > 5


def test_ref_link_guided_walk(orc, lib, tmp):                 # T/utils/traversal/TraversalUtilsTest.java:19-47, 56-84
    kid = ["TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"]
    for mom in (["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC"], ["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC", kid[0]]):
        haps = {"mom": mom, "kid": kid}
        order = orc.java_string_hashmap_order(["mom", "kid"])
        cs = Case(orc, tmp, lib, [(s, haps[s]) for s in order], 7, link_samples=["kid"], name="tu%d" % len(mom))
        e = (TraversalEngineFactory(lib=lib).traversalColors(cs.g.getColorForSampleName("kid")).traversalDirection(BOTH)
             .combinationOperator(OR).stoppingRule(ContigStopper).graph(cs.g).links(cs.links["kid"]).make())
        assert TraversalUtils.toContig(e.walk("TGAGATT")) == kid[0]


# ------------------------------------------------------------------ differential: random graphs vs oracle
def case_random_walks(orc, lib, tmp, k, seed, with_links, n=1500):
    rng = random.Random(seed * 7919 + k)
    base = genome_with_repeats(rng, n, n_rep=8, rep_len=(k // 2 + 1, 4 * k), copies=(2, 3))
    kid = mutate(rng, base, snv=0.01, indel=0.003)
    dad = mutate(rng, base, snv=0.02, indel=0.003)
    haps = [("kid", [kid]), ("mom", [base]), ("dad", [dad, mutate(rng, dad)])]
    reads = None
    if with_links:
        rl = max(3 * k, 60)
        reads = {"kid": [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), max(1, rl // 4))] + [kid[-rl:]],
                 "mom": [base[i:i + rl] for i in range(0, max(1, len(base) - rl + 1), max(1, rl // 3))]}
    cs = Case(orc, tmp, lib, haps, k, link_samples=(["kid", "mom"] if with_links else []), reads=reads,
              name="w%d_%d_%d" % (k, seed, int(with_links)))
    kmers = cs.all_kmers()
    seeds = rng.sample(kmers, min(120, len(kmers)))
    seeds = [s if rng.random() < 0.5 else orc.revcomp(s) for s in seeds]
    seeds += [rand_seq(rng, k), "N" * k, kid[:k], kid[-k:]]
    L = ["kid"] if with_links else []
    # with links a walk keeps circling a tandem repeat until maxLength (the reference does too), so the
    # differential runs cap maxLength to keep the oracle's share of the test in seconds
    ML = 400 if with_links else 75000
    compare_walks(cs, seeds, trav=[0], links=L, max_len=ML)
    compare_walks(cs, seeds[:40], trav=[0], links=L, op=AND, direction=FORWARD, max_len=ML)
    compare_walks(cs, seeds[:40], trav=[0], links=L, direction=REVERSE, max_len=ML)
    compare_walks(cs, seeds[:40], trav=[1], links=(["mom"] if with_links else []), max_len=ML)
    compare_walks(cs, seeds[:40], trav=[0, 2], links=L, max_len=ML)
    compare_walks(cs, seeds[:40], trav=[0], recruit=[1, 2], links=L, op=AND, max_len=ML)
    compare_walks(cs, seeds[:40], trav=[0], links=L, max_len=7)
    if with_links:
        compare_walks(cs, seeds[:40], trav=[0], links=["kid", "mom"], max_len=ML)   # only kid's links belong to the traversal sample
        compare_walks(cs, seeds[:40], trav=[2], links=["kid"], max_len=ML)          # cursor driven, no usable links


def case_lowercase_queries(orc, lib, tmp):
    """Encoding a k-mer takes either case (CortexRecord.encodeBinaryKmer -> charToBinaryNucleotide, CortexRecord.java:347-360); LOOKING ONE UP
    compares bytes with the records' upper-case k-mers (CortexGraph.findRecord, :272-317): "acgt…" has no record, like a string with an N.
    findRecord, walk, dfs sources and sinks, neighbours, assemble and the cursor agree with the oracle on lower- and mixed-case strings —
    through the host-seed and the device-seed entry of the walks alike."""
    import ctypes as C
    rng = random.Random(77)
    k = 31
    base = genome_with_repeats(rng, 1500, n_rep=4, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
    kid = mutate(rng, base, snv=0.01, indel=0.0)
    rl = 3 * k
    reads = {"kid": [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), rl // 4)] + [kid[-rl:]]}
    cs = Case(orc, tmp, lib, [("kid", [kid]), ("mom", [base])], k, link_samples=["kid"], reads=reads, name="lower")
    kmers = cs.all_kmers()
    up = [kmers[5], kmers[200], kid[300:300 + k], orc.revcomp(kid[700:700 + k])]
    qs = []
    for q in up:
        qs += [q, q.lower(), q[:7].lower() + q[7:], q[:-1] + q[-1].lower()]
    # findRecord
    for q in qs:
        cr = cs.g.findRecord(q)
        assert (cr.index if cr is not None else -1) == cs.og.find(q)[0], q
        assert (cr is None) == (q != q.upper())
    # encodeBinaryKmer: either case, the same words
    for q in up:
        w1, w2 = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
        lib.check(lib.dll.ldbg_kmer_encode(q.encode(), k, w1))
        lib.check(lib.dll.ldbg_kmer_encode(q.lower().encode(), k, w2))
        assert list(w1)[:1] == list(w2)[:1] == orc.encode_kmer(q) == orc.encode_kmer(q.lower())
    # walks (compare_walks runs the host-seed and the device-seed entry), with and without links
    compare_walks(cs, qs, trav=[0], links=["kid"], max_len=500)
    compare_walks(cs, qs, trav=[0])
    # dfs: sources and sinks
    oe = orc.Engine(cs.og, [0], links=[cs.olinks["kid"]], stopper="DestinationStopper", direction=orc.FORWARD, max_length=300)
    e = (TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).stoppingRule("DestinationStopper").traversalDirection(FORWARD).maxBranchLength(300)
         .links(cs.links["kid"]).make())
    src = kid[100:100 + k]
    snk = kid[100 + 2 * k:100 + 3 * k]
    for source, sinks in ((src, [snk]), (src, [snk.lower()]), (src.lower(), [snk]), (src[:3].lower() + src[3:], [snk, snk.lower()])):
        r = oe.dfs(source, sinks)
        g = e.dfs(source, *sinks)
        assert (g is None) == r.is_null, (source, sinks)
        if g is not None:
            assert g.vertex_tuples() == r.vertices() and g.edge_tuples() == r.edges(), (source, sinks)
        r.free()
    # neighbours, assemble, cursor
    oc = orc.Engine(cs.og, [0], links=[cs.olinks["kid"]], stopper="ContigStopper")
    ec = TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).stoppingRule(ContigStopper).links(cs.links["kid"]).make()
    for q in qs:
        assert [(v.getKmerAsString(), v.getCortexRecord().index if v.getCortexRecord() is not None else -1) for v in ec.getNextVertices(q)] == oc.next_vertices(q), q
        assert [(v.getKmerAsString(), v.getCortexRecord().index if v.getCortexRecord() is not None else -1) for v in ec.assemble(q)] == oc.assemble(q, 75000), q
        ec.seek(q); oc.seek(q)
        assert ec.hasNext() == oc.has_next(), q


def case_concurrent_engines(orc, lib, tmp):
    """two engines on ONE graph, each driven by its own host thread (every engine has its own HIP stream: csrc/walk.cpp, Engine::Engine):
    walk batches and dfs batches running side by side give what they give one after the other"""
    import threading
    rng = random.Random(4711)
    k = 31
    base = genome_with_repeats(rng, 6000, n_rep=10, rep_len=(k // 2 + 1, 4 * k), copies=(2, 3))
    kid = mutate(rng, base, snv=0.01, indel=0.003)
    rl = 3 * k
    reads = {"kid": [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), max(1, rl // 4))] + [kid[-rl:]]}
    cs = Case(orc, tmp, lib, [("kid", [kid]), ("mom", [base])], k, link_samples=["kid"], reads=reads, name="conc")
    kmers = cs.all_kmers()
    batches = [rng.sample(kmers, 400) for _ in range(2)]
    engines = [cs.engines(trav=[0], links=["kid"], max_len=2000)[1] for _ in range(2)]
    expect = [engines[0].walk_batch(b) for b in batches]
    out = [[None] * 6, [None] * 6]

    def work(i):
        for r in range(6):
            out[i][r] = engines[i].walk_batch(batches[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        for r in range(6):
            got, wl = out[i][r]
            assert got == expect[i][0] and (wl == expect[i][1]).all(), (i, r)
    oe = cs.engines(trav=[0], links=["kid"], max_len=2000)[0]
    for s, c in list(zip(batches[0], expect[0][0]))[:40]:
        assert c == oe.walk(s)[0]
    # the same through the host mirror's EnginePool: five batches dealt out to two engines, results in the order of the batches
    pool = (TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).maxBranchLength(2000).stoppingRule(ContigStopper)
            .links(cs.links["kid"]).make_pool(2))
    more = [batches[0], batches[1], batches[1][:50], batches[0][:7], batches[1]]
    res = pool.walk_batches(more)
    assert len(res) == 5
    for b, (got, wl) in zip(more, res):
        ref_c, ref_wl = engines[0].walk_batch(b)
        assert got == ref_c and (wl == ref_wl).all()
    pool.close()


def case_dense_cycles(orc, lib, tmp, seed):
    """tiny k on a low-complexity genome: junctions and cycles everywhere"""
    rng = random.Random(seed)
    k = rng.choice([4, 5, 6])
    g1 = "".join(rng.choice("ACGT") for _ in range(rng.randint(40, 160)))
    g2 = mutate(rng, g1, snv=0.05)
    reads = {"a": [g1[i:i + 5 * k] for i in range(0, len(g1), k)], "b": [g2]}
    cs = Case(orc, tmp, lib, [("a", [g1]), ("b", [g2])], k, link_samples=["a", "b"], reads=reads, name="d%d" % seed)
    seeds = cs.all_kmers()
    seeds = seeds + [orc.revcomp(s) for s in seeds]
    compare_walks(cs, seeds, trav=[0])
    compare_walks(cs, seeds, trav=[0], links=["a"], max_len=150)
    compare_walks(cs, seeds, trav=[1], links=["b"], recruit=[0], max_len=150)
    compare_walks(cs, seeds, trav=[0, 1], links=["a", "b"], max_len=12)


def compare_walk_vertices(case, seeds, **cfg):
    """vertex by vertex: (k-mer, record, copyIndex, index) of every walk against the vertex set of the oracle's dfs graph"""
    oe, e = case.engines(**cfg)
    seeds = list(seeds)
    e.walk_batch_arrays(seeds, fetch=False)
    for i, s in enumerate(seeds):
        got = e.walk_vertices(i)
        r = oe.dfs(s)
        exp = [] if r.is_null else r.vertices()
        gt = sorted((v.getKmerAsString(), v.getCortexRecord() is not None, v.getCopyIndex(), v.getIndex()) for v in got)
        if got:
            et = sorted((km, rec >= 0, ci, ix) for km, rec, ci, ix in exp)
            assert gt == et, (s, cfg, len(gt), len(et), [x for x in gt if x not in et][:4], [x for x in et if x not in gt][:4])
        r.free()


def case_run_steps(orc, lib, tmp, seed):
    """long unbranched stretches crossed several times: tandem arrays with long units (a link-guided walk goes round them, every
    revolution crossing the same stretches with the next copyIndex), seeds in the middle of a stretch that the walk comes
    back to, inverted repeats (the same stretch in both orientations), maxLength falling inside a stretch.  Exercises the run
    steps and the repeat detection of the walk kernel (csrc/runstep.h) against the k-mer-by-k-mer oracle."""
    rng = random.Random(1000 + seed)
    k = rng.choice([9, 11, 15, 21])
    parts = []
    for _ in range(3):
        parts.append(rand_seq(rng, rng.randint(40, 160)))
        unit = rand_seq(rng, rng.randint(k + 3, 3 * k + 20))
        parts.append(unit * rng.randint(2, 9))
        parts.append(rand_seq(rng, rng.randint(30, 90)))
        inv = rand_seq(rng, rng.randint(k + 5, 3 * k))
        parts.append(inv + rand_seq(rng, rng.randint(5, 40)) + orc.revcomp(inv))
    rep = rand_seq(rng, rng.randint(2 * k, 4 * k))
    g1 = "".join(parts) + rep + rand_seq(rng, 50) + rep + rand_seq(rng, 60)
    g2 = mutate(rng, g1, snv=0.01, indel=0.0)
    rl = rng.choice([4 * k, 6 * k, 12 * k])
    reads = {"a": [g1[i:i + rl] for i in range(0, max(1, len(g1) - rl + 1), max(1, k // 2))] + [g1[-rl:]]}
    cs = Case(orc, tmp, lib, [("a", [g1]), ("b", [g2])], k, link_samples=["a"], reads=reads, name="rs%d" % seed)
    kmers = cs.all_kmers()
    seeds = rng.sample(kmers, min(100, len(kmers)))
    seeds = [s if rng.random() < 0.5 else orc.revcomp(s) for s in seeds] + [g1[:k], g1[-k:], orc.revcomp(g1[100:100 + k])]
    compare_walks(cs, seeds[:6], trav=[0], links=["a"], max_len=75000)
    for ml in (3000, 997):
        compare_walks(cs, seeds, trav=[0], links=["a"], max_len=ml)
    compare_walks(cs, seeds, trav=[0])
    compare_walks(cs, seeds[:60], trav=[0, 1], links=["a"], max_len=2000)
    compare_walks(cs, seeds[:60], trav=[0], recruit=[1], links=["a"], max_len=1500, direction=FORWARD)
    compare_walks(cs, seeds[:60], trav=[1], links=["a"], max_len=1500)           # cursor driven, no usable links
    for ml in rng.sample(range(20, 400), 6):
        compare_walks(cs, seeds[:40], trav=[0], links=["a"], max_len=ml)
        compare_walks(cs, seeds[:40], trav=[0], max_len=ml)
    compare_walk_vertices(cs, seeds[:50], trav=[0], links=["a"], max_len=1200)
    compare_walk_vertices(cs, seeds[:30], trav=[0], max_len=300)


def case_long_walks(orc, lib, tmp):
    """walks far longer than the initial per-strand visited table (4096 entries) and than one path block (1024):
    exercises table regrowth, block chaining, and the maxLength cut"""
    rng = random.Random(99)
    g1 = rand_seq(rng, 9000)
    cs = Case(orc, tmp, lib, [("a", [g1])], 21, link_samples=["a"], reads={"a": [g1[:300]]}, name="long")
    seeds = [g1[i:i + 21] for i in (0, 1500, 4400, 8979)] + [orc.revcomp(g1[3000:3021])]
    exp = compare_walks(cs, seeds, trav=[0])
    assert max(len(c) for c in exp) == 9000
    compare_walks(cs, seeds, trav=[0], links=["a"])
    compare_walks(cs, seeds, trav=[0], max_len=5000)
    compare_walks(cs, seeds, trav=[0], links=["a"], max_len=3000)


def case_hash_collision(orc, lib, tmp):
    """Q6: k-mers whose Arrays.hashCode equals that of their reverse complement make CanonicalKmer.isFlipped()
    lie; the walk then takes its neighbours from the wrong orientation.  tests/golden/hash_collisions.txt holds
    such k-mers (found by tests/golden/make_hash_collisions.py); the walks through them must still match the oracle."""
    rng = random.Random(5)
    for x in open(os.path.join(GOLDEN, "hash_collisions.txt")).read().split():
        k = len(x)
        assert orc.jhash_bytes(x) == orc.jhash_bytes(orc.revcomp(x)) and x != orc.revcomp(x)
        flank = lambda n: rand_seq(rng, n)
        xo = x if rng.random() < 0.5 else orc.revcomp(x)
        h1 = flank(3 * k) + xo + flank(3 * k)
        h2 = flank(2 * k) + h1[2 * k: 5 * k + 5] + flank(2 * k)        # shares the colliding k-mer, forks around it
        h3 = flank(k) + orc.revcomp(xo) + flank(k)
        cs = Case(orc, tmp, lib, [("a", [h1, h2, h3]), ("b", [h1])], k, link_samples=["a"],
                  reads={"a": [h1, h2, h3]}, name="coll%d_%s" % (k, x[:6]))
        # the strict reference behaviour: the flipped orientation of x is reported as NOT flipped
        assert not orc.is_flipped(orc.revcomp(orc.canonical(x)))
        seeds = [x, orc.revcomp(x)] + [h1[i:i + k] for i in range(2 * k, 4 * k + 1, 3)]
        seeds += [orc.revcomp(s) for s in seeds]
        for links in ([], ["a"]):
            compare_walks(cs, seeds, trav=[0], links=links, max_len=300)
            # with recruitment colours the wrong-orientation neighbours of a Q6 vertex can be absent from the
            # graph, which is a NullPointerException in the reference (Q14): compare seed by seed, errors included
            oe, e = cs.engines(trav=[1], links=links, recruit=[0], max_len=300)
            n_npe = 0
            for sd in seeds:
                try:
                    exp = oe.walk(sd)[0]
                except orc.OracleError as ex:
                    assert "NullPointerException" in str(ex)
                    exp = None
                try:
                    got = e.walk_batch([sd])[0][0]
                except ca.JavaNullPointerException:
                    got = None
                assert got == exp, (sd, got, exp)
                n_npe += exp is None
        idx, _, _ = cs.g.find_batch([x, orc.revcomp(x)])
        assert idx[0] == idx[1] >= 0
        # dfs graphs through the colliding k-mer (children order, sinks, Q6 vertices in the log)
        for links in ([], ["a"]):
            for stopper in ("ExplorationStopper", "DestinationStopper", "VisualizationStopper"):
                compare_dfs(cs, seeds, sinks=[[h1[4 * k:5 * k], orc.revcomp(x)] for _ in seeds], trav=[0], stopper=stopper, links=links, max_len=300)
        # cursor through the colliding k-mer, both ways, against the oracle's cursor
        oe, e = cs.engines(trav=[0], links=["a"])
        for start, fwd in ((h1[k:2 * k], True), (h1[5 * k:6 * k], False)):
            oe.seek(start); e.seek(start)
            for _ in range(6 * k):
                a, b = (oe.has_next(), e.hasNext()) if fwd else (oe.has_previous(), e.hasPrevious())
                assert a == b
                if not a:
                    break
                ov = oe.next() if fwd else oe.previous()
                v = e.next() if fwd else e.previous()
                assert ov[0] == v.getKmerAsString() and (ov[1] >= 0) == (v.getCortexRecord() is not None)


# ------------------------------------------------------------------ dfs with stopping rules vs oracle
def dfs_engines(case, trav, stopper, links=(), rois=None, join=(), recruit=(), secondary=(), op=OR, direction=BOTH, max_len=75000):
    """rois: (oracle graph, product graph) or None"""
    oe = case.orc.Engine(case.og, trav, links=[case.olinks[s] for s in links], rois=rois[0] if rois else None,
                         joining_colors=join, recruitment_colors=recruit, secondary_colors=secondary, op_and=(op == AND), direction=direction,
                         max_length=max_len, stopper=stopper)
    f = (TraversalEngineFactory(lib=case.lib).traversalColors(*trav).graph(case.g).combinationOperator(op)
         .traversalDirection(direction).maxBranchLength(max_len).stoppingRule(stopper))
    if join:
        f.joiningColors(*join)
    if recruit:
        f.recruitmentColors(*recruit)
    if secondary:
        f.secondaryColors(*secondary)
    if rois:
        f.rois(rois[1])
    if links:
        f.links(*[case.links[s] for s in links])
    return oe, f.make()


def _oracle_dfs(case, oe, seed, sinks, color):
    try:
        r = oe.dfs(seed, sinks)
    except case.orc.OracleError as ex:
        return ("error", "NullPointerException" if "NullPointerException" in str(ex) else "CortexJDKException")
    if r.is_null:
        r.free()
        return None
    try:
        contig = r.walk(seed, color)
    except case.orc.OracleError:
        contig = "<error>"
    out = (r.vertices(), r.edges(), contig)
    r.free()
    return out


def _product_dfs_one(e, g, seed, color):
    if g is None:
        return None
    try:
        contig = g.walk_contig(seed, color)
    except ca.JavaNullPointerException:
        contig = "<error>"
    return (g.vertex_tuples(), g.edge_tuples(), contig)


def compare_dfs(case, seeds, sinks=None, **cfg):
    """every seed's dfs graph (vertices and edges in insertion order), toWalk/toContig of it, and the errors the
    reference raises, against the oracle"""
    oe, e = dfs_engines(case, **cfg)
    color = cfg["trav"][0]
    seeds = list(seeds)
    sinks = sinks if sinks is not None else [[] for _ in seeds]
    it0 = oe.kmers_traversed()
    exp = [_oracle_dfs(case, oe, s, sk, color) for s, sk in zip(seeds, sinks)]
    exp_iters = oe.kmers_traversed() - it0
    n_err = sum(1 for x in exp if isinstance(x, tuple) and x[0] == "error")
    if n_err == 0:
        got = [_product_dfs_one(e, g, s, color) for g, s in zip(e.dfs_batch(seeds, sinks), seeds)]
        assert e.dfs_kmers_traversed == exp_iters, (e.dfs_kmers_traversed, exp_iters)
    else:       # an exception aborts a batch call, like the reference's loop would: compare seed by seed
        got = []
        for s, sk in zip(seeds, sinks):
            try:
                got.append(_product_dfs_one(e, e.dfs_batch([s], [sk])[0], s, color))
            except ca.JavaNullPointerException:
                got.append(("error", "NullPointerException"))
            except ca.CortexJDKException:
                got.append(("error", "CortexJDKException"))
    for i, s in enumerate(seeds):
        assert got[i] == exp[i], (s, sinks[i], cfg, got[i], exp[i])
    return exp


def case_dfs_rules(orc, lib, tmp, k, seed, with_links):
    """every stopping rule on a three-colour graph with repeats, bubbles and a ROI graph of child-only k-mers"""
    rng = random.Random(seed * 104729 + k)
    base = genome_with_repeats(rng, 700, n_rep=5, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
    kid = mutate(rng, base, snv=0.02, indel=0.004)
    dad = mutate(rng, base, snv=0.02, indel=0.003)
    haps = [("kid", [kid]), ("mom", [base]), ("dad", [dad])]
    reads = None
    if with_links:
        rl = max(3 * k, 40)
        reads = {"kid": [kid[i:i + rl] for i in range(0, max(1, len(kid) - rl + 1), max(1, rl // 4))] + [kid[-rl:]]}
    cs = Case(orc, tmp, lib, haps, k, link_samples=(["kid"] if with_links else []), reads=reads, name="dfs%d_%d_%d" % (k, seed, int(with_links)))
    # ROI graph: the k-mers only the child has
    parents = set()
    for h in (base, dad):
        parents |= {orc.canonical(h[i:i + k]) for i in range(len(h) - k + 1)}
    novel = [kid[i:i + k] for i in range(len(kid) - k + 1) if orc.canonical(kid[i:i + k]) not in parents]
    roi_path = str(tmp / "rois.ctx")
    orc.build_graph(roi_path, [("kid", novel or [kid[:k]])], k)
    rois = (orc.Graph(roi_path, tuned=True), CortexGraph(roi_path, lib=lib))
    kmers = cs.all_kmers()
    seeds = rng.sample(kmers, min(40, len(kmers)))
    seeds = [s if rng.random() < 0.5 else orc.revcomp(s) for s in seeds] + novel[:10] + [rand_seq(rng, k), kid[:k], kid[-k:]]
    L = ["kid"] if with_links else []
    ML = 300
    # sinks: a child k-mer 5..150 bases downstream of the seed (gap closing), or unrelated ones
    pos = {kid[i:i + k]: i for i in range(len(kid) - k + 1)}
    sinks = []
    for s in seeds:
        i = pos.get(s, pos.get(orc.revcomp(s)))
        if i is None:
            sinks.append([rand_seq(rng, k)])
            continue
        j = min(len(kid) - k, i + rng.randint(5, 150))
        t = kid[j:j + k]
        sinks.append([t if s in pos else orc.revcomp(kid[max(0, i - rng.randint(5, 150)):][:k])] + ([rand_seq(rng, k)] if rng.random() < 0.3 else []))
    for stopper in ca.traversal.STOPPING_RULES:
        if stopper == "CycleCollapsingContigStopper" and k < 15:
            continue        # forks at every junction until everything is visited: exponential in the reference as well
        cfg = dict(trav=[0], stopper=stopper, links=L, max_len=ML, join=[1, 2], rois=rois)
        if stopper == "NovelKmerLimitedContigStopper":
            # never fails and only succeeds after a novel k-mer: from anywhere else the reference forks without end
            nov = set(novel) | {orc.revcomp(x) for x in novel}
            sel = [i for i, s in enumerate(seeds) if s in nov]
            compare_dfs(cs, [seeds[i] for i in sel], sinks=[sinks[i] for i in sel], **cfg)
            continue
        compare_dfs(cs, seeds, sinks=sinks, **cfg)
    # variations on the rules the reference's commands use most
    compare_dfs(cs, seeds, sinks=sinks, trav=[0], stopper="DestinationStopper", links=L, max_len=ML, direction=FORWARD)
    compare_dfs(cs, seeds, sinks=sinks, trav=[0], stopper="DestinationStopper", links=L, max_len=40, direction=REVERSE, op=AND)
    compare_dfs(cs, seeds, trav=[0, 2], stopper="ExplorationStopper", links=L, max_len=ML)
    compare_dfs(cs, seeds, trav=[1], stopper="ExplorationStopper", max_len=ML, recruit=[0])
    if k >= 15:
        compare_dfs(cs, seeds, trav=[0], stopper="CycleCollapsingContigStopper", links=L, max_len=ML, op=AND)
    compare_dfs(cs, seeds, trav=[0], stopper="NovelPartitionStopper", links=L, max_len=ML)              # no ROI graph: the reference throws
    compare_dfs(cs, seeds, trav=[0], stopper="BubbleOpeningStopper", links=L, max_len=ML, join=[1])     # no ROI graph: NullPointerException
    compare_dfs(cs, seeds, trav=[0], stopper="ContigStopper", links=L, max_len=ML)
    if k % 2:       # addSecondaryColors (the visualiser's configuration): the other colours' edges at every vertex
        compare_dfs(cs, seeds, trav=[0], stopper="ExplorationStopper", links=L, max_len=ML, secondary=[1, 2])
        compare_dfs(cs, seeds, trav=[0, 2], stopper="ContigStopper", links=L, max_len=ML, secondary=[0, 1], op=AND)
    for g in rois:
        g.close()


def case_dfs_run_steps(orc, lib, tmp, seed):
    """searches through long unbranched stretches (csrc/dfs.cpp: dfs_run_step): sinks in the middle of a stretch, on its fringes and
    nowhere; maxLength, the rules' size limits (VisualizationStopper 500, DestinationStopper's junction limit at graph sizes 2231 /
    5108) and failing branches ending inside a stretch with the search going on in a sibling; stretches entered again by a child
    branch after the parent went through them (tandem arrays with links); both directions; with and without the cursor."""
    rng = random.Random(7000 + seed)
    k = rng.choice([9, 11, 15, 21])
    parts = []
    for _ in range(4):
        parts.append(rand_seq(rng, rng.randint(300, 1500)))
        unit = rand_seq(rng, rng.randint(k + 3, 3 * k + 20))
        parts.append(unit * rng.randint(2, 5))
        parts.append(rand_seq(rng, rng.randint(100, 900)))
        inv = rand_seq(rng, rng.randint(k + 5, 3 * k))
        parts.append(inv + rand_seq(rng, rng.randint(5, 40)) + orc.revcomp(inv))
    rep = rand_seq(rng, rng.randint(2 * k, 4 * k))
    g1 = "".join(parts) + rep + rand_seq(rng, 250) + rep + rand_seq(rng, 400)
    g2 = mutate(rng, g1, snv=0.004, indel=0.0)
    rl = rng.choice([4 * k, 8 * k])
    reads = {"a": [g1[i:i + rl] for i in range(0, max(1, len(g1) - rl + 1), max(1, k // 2))] + [g1[-rl:]]}
    cs = Case(orc, tmp, lib, [("a", [g1]), ("b", [g2])], k, link_samples=["a"], reads=reads, name="drs%d" % seed)
    pos = {}
    for i in range(len(g1) - k + 1):
        pos.setdefault(g1[i:i + k], i)
    starts = rng.sample(range(len(g1) - k), 50)
    seeds, sinks = [], []
    for i in starts:
        s = g1[i:i + k]
        d = rng.choice([rng.randint(1, 12), rng.randint(20, 400), rng.randint(400, 3000)])
        j = min(len(g1) - k, i + d)
        sk = [g1[j:j + k]]
        r = rng.random()
        if r < 0.2:
            sk = [rand_seq(rng, k)]                          # unreachable: the search fails at its limits
        elif r < 0.4:
            sk.append(g1[max(0, i - d):max(0, i - d) + k])   # one ahead, one behind
        if rng.random() < 0.3:
            s, sk = orc.revcomp(s), [orc.revcomp(x) for x in sk]
        seeds.append(s)
        sinks.append(sk)
    compare_dfs(cs, seeds[:4], sinks=sinks[:4], trav=[0], stopper="DestinationStopper", links=["a"], max_len=30000)
    for ml in (5000, 1200):
        compare_dfs(cs, seeds, sinks=sinks, trav=[0], stopper="DestinationStopper", links=["a"], max_len=ml)
    compare_dfs(cs, seeds, sinks=sinks, trav=[0], stopper="DestinationStopper", links=["a"], max_len=4000, direction=FORWARD)
    compare_dfs(cs, seeds, sinks=sinks, trav=[0], stopper="DestinationStopper", max_len=6000)
    compare_dfs(cs, seeds[:30], sinks=sinks[:30], trav=[0, 1], stopper="DestinationStopper", max_len=3000, direction=FORWARD)
    compare_dfs(cs, seeds[:30], sinks=sinks[:30], trav=[0, 1], stopper="DestinationStopper", links=["a"], max_len=3000)
    for stopper in ("ContigStopper", "ExplorationStopper", "VisualizationStopper", "BubbleClosingStopper", "GapClosingStopper"):
        compare_dfs(cs, seeds[:25], sinks=sinks[:25], trav=[0, 1], stopper=stopper, max_len=rng.choice([700, 2500]))
    for stopper in ("GapClosingStopper", "BubbleClosingStopper"):
        compare_dfs(cs, seeds[:25], sinks=sinks[:25], trav=[0], stopper=stopper, max_len=20000)
    for stopper in ("ContigStopper", "ExplorationStopper", "VisualizationStopper"):
        compare_dfs(cs, seeds[:25], sinks=sinks[:25], trav=[0], stopper=stopper, links=["a"], max_len=rng.choice([333, 2500, 20000]))


def case_factory_validation(orc, lib, tmp):
    """TraversalEngineFactory.make() :54-88 — the configuration errors and their messages"""
    rng = random.Random(12)
    cs = Case(orc, tmp, lib, [("a", [rand_seq(rng, 80)]), ("b", [rand_seq(rng, 80)])], 11, name="cfg")
    F = lambda: TraversalEngineFactory(lib=lib).graph(cs.g)

    def message(f):
        with pytest.raises(ca.CortexJDKException) as ex:
            f.make()
        return str(ex.value)
    assert "Traversal color(s) must be specified." in message(F())
    assert "Traversal colors must be between 0 and 2 (provided 2)" in message(F().traversalColors(2))
    assert "Joining colors must be between 0 and 2 (provided 5)" in message(F().traversalColors(0).joiningColors(5))
    assert "Joining colors must be between 0 and 2 (provided -1)" in message(F().traversalColors(0).joiningColors(-1))
    assert "Recruitment colors must be between 0 and 2 (provided 2)" in message(F().traversalColors(0).recruitmentColors(2))
    assert "Secondary colors must be between 0 and 2 (provided 7)" in message(F().traversalColors(1).secondaryColors(7))
    assert "Must provide stopping rule for graph traversal" in message(F().traversalColors(0).stoppingRule(None))
    with pytest.raises(ca.JavaNullPointerException):
        TraversalEngineFactory(lib=lib).traversalColors(0).make()
    assert "Traversal color(s) must be specified." in message(TraversalEngineFactory(lib=lib))
    e = F().traversalColors(0, 1).joiningColors(1).recruitmentColors(1).make()      # a valid one
    e.close()


def case_dfs_packed_results(orc, lib, tmp):
    """results that are one branch per direction stay packed until their k-mers are asked for: sizes and the vertex / edge lists read
    from the packed form (ldbg_dfs_result_get without k-mer words) are those of the unpacked graphs"""
    import ctypes as C
    rng = random.Random(31)
    g1 = rand_seq(rng, 900)
    cs = Case(orc, tmp, lib, [("a", [g1])], 15, link_samples=["a"], name="packed")
    seeds = [g1[i:i + 15] for i in (0, 100, 417, 885)] + [orc.revcomp(g1[300:315]), rand_seq(rng, 15)]
    sinks = [[g1[i:i + 15]] for i in (60, 300, 500, 880)] + [[orc.revcomp(g1[250:265])], [g1[:15]]]
    for direction in (BOTH, FORWARD):
        e = (TraversalEngineFactory(lib=lib).traversalColors(0).traversalDirection(direction).combinationOperator(OR).stoppingRule("DestinationStopper")
             .graph(cs.g).links(cs.links["a"]).make())
        src = np.frombuffer("".join(seeds).encode(), dtype=np.uint8)
        sink_buf = np.frombuffer("".join(x[0] for x in sinks).encode(), dtype=np.uint8)
        off = np.arange(len(seeds) + 1, dtype=np.int64)
        batch = e.dfs_batch_arrays(src, len(seeds), sink_buf, off)
        raw = []
        for i in range(len(seeds)):
            isnull, nv, ne = C.c_int(), C.c_int64(), C.c_int64()
            e._lib.check(e._d.ldbg_dfs_result_sizes(batch.h, C.c_int64(i), C.byref(isnull), C.byref(nv), C.byref(ne)))
            if isnull.value:
                raw.append(None)
                continue
            rec, cp, ix = np.zeros(max(1, nv.value), np.int64), np.zeros(max(1, nv.value), np.int32), np.zeros(max(1, nv.value), np.int32)
            es, et, ec = (np.zeros(max(1, ne.value), np.int32) for _ in range(3))
            P = lambda a: a.ctypes.data_as(C.c_void_p)
            e._lib.check(e._d.ldbg_dfs_result_get(batch.h, C.c_int64(i), None, P(rec), P(cp), P(ix), P(es), P(et), P(ec)))
            raw.append((nv.value, ne.value, list(rec[:nv.value]), list(cp[:nv.value]), list(ix[:nv.value]), list(zip(es[:ne.value], et[:ne.value], ec[:ne.value]))))
        n_graphs = 0
        for i in range(len(seeds)):
            gi = batch.graph(i)
            assert (gi is None) == (raw[i] is None)
            if gi is None:
                continue
            vt, et_ = gi.vertex_tuples(), gi.edge_tuples()          # (asks for the k-mers: the batch is unpacked here)
            assert raw[i][0] == len(vt) and raw[i][1] == len(et_)
            assert raw[i][2] == [v[1] for v in vt] and raw[i][3] == [v[2] for v in vt] and raw[i][4] == [v[3] for v in vt]
            assert [tuple(int(x) for x in t) for t in raw[i][5]] == et_
            n_graphs += 1
        assert n_graphs >= 4
        e.close()


def case_dfs_dense(orc, lib, tmp, seed):
    """tiny k: junctions everywhere, deep recursion, many failing branches (visited-set undo, log truncation)"""
    rng = random.Random(1000 + seed)
    k = rng.choice([4, 5, 6])
    g1 = "".join(rng.choice("ACGT") for _ in range(rng.randint(40, 120)))
    g2 = mutate(rng, g1, snv=0.06)
    reads = {"a": [g1[i:i + 5 * k] for i in range(0, len(g1), k)]}
    cs = Case(orc, tmp, lib, [("a", [g1]), ("b", [g2])], k, link_samples=["a"], reads=reads, name="dd%d" % seed)
    seeds = cs.all_kmers()
    seeds = seeds[:30] + [orc.revcomp(s) for s in seeds[:30]]
    sinks = [[rng.choice(seeds), rng.choice(seeds)] for _ in seeds]
    # only rules that bound their own recursion: the others fork without end on a graph this dense (in the reference too),
    # and with links a walk circles a cycle for ever unless the rule looks at reachedMaxBranchLength / branchSize
    for stopper in ("DestinationStopper", "ExplorationStopper", "GapClosingStopper", "VisualizationStopper", "BubbleClosingStopper",
                    "PairedReadClosingStopper"):
        compare_dfs(cs, seeds, sinks=sinks, trav=[0], stopper=stopper, max_len=60, join=[1])
    for stopper in ("DestinationStopper", "ExplorationStopper", "VisualizationStopper", "PairedReadClosingStopper"):
        compare_dfs(cs, seeds, sinks=sinks, trav=[0], stopper=stopper, links=["a"], max_len=60, join=[1])
    compare_dfs(cs, seeds, sinks=sinks, trav=[0, 1], stopper="ExplorationStopper", max_len=60)


def test_ref_fill_gaps(orc, lib, tmp):                            # TraversalUtilsTest.java:19-97 (V13, both cases)
    from corticall_amd import traversal_utils as tu
    cases = [({"mom": ["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC"], "kid": ["TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"]},
              [("TGGCTAG", "kid", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"), ("TGGCTAG", "mom", "TGGCTAGGTCATTATGATATTAAAATGCTAGCGC")]),
             ({"mom": ["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"], "kid": ["TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"]},
              [("TGAGATT", "kid", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"), ("TGATATT", "mom", "TGGCTAGGTCATTATGATATTAAAATGCTAGCGC"),
               ("TGAGATT", "mom", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC")])]
    for ci, (haps, expected) in enumerate(cases):
        # (a java.util.HashMap of the two sample names iterates "kid" before "mom": TempGraphAssembler gives kid colour 0)
        order = orc.java_string_hashmap_order(["mom", "kid"])
        cs = Case(orc, tmp, lib, [(s_, haps[s_]) for s_ in order], 7, link_samples=["mom", "kid"], name="v13_%d" % ci)
        colors = list(range(cs.g.getNumColors()))
        kid = cs.g.getColorForSampleName("kid")
        e = (TraversalEngineFactory(lib=lib).traversalColors(kid).traversalDirection(BOTH).combinationOperator(OR).stoppingRule("ContigStopper")
             .graph(cs.g).links(cs.links["kid"]).make())
        w = e.walk("TGAGATT")
        assert tu.toContig(w) == haps["kid"][0]
        gapped = tu.toGraph(w, colors)
        # the five strings the reference's test asserts: reproduced when a search result is joined on (k-mer, record, copyIndex)
        filled = tu.fillGaps(gapped, cs.g, [cs.links["mom"], cs.links["kid"]], colors)
        for seed, sample, hap in expected:
            assert tu.toContig(tu.toWalk(filled, seed, cs.g.getColorForSampleName(sample))) == hap, (ci, seed, sample)
        assert filled.vertexSet()[:len(gapped.vertexSet())] == gapped.vertexSet()      # what was there stays there, in its place
        # the literal reading (CortexVertex.equals includes `index`): the same searches, the filled stretch joined at its far end only
        lit = tu.fillGaps(gapped, cs.g, [cs.links["mom"], cs.links["kid"]], colors, relabel=False)
        assert {v.getKmerAsString() for v in lit.vertexSet()} == {v.getKmerAsString() for v in filled.vertexSet()}
        assert len(lit.vertexSet()) > len(filled.vertexSet())                         # (at least) the search's own source vertex, index 0
        if ci == 0:
            mom_walk = tu.toContig(tu.toWalk(lit, "TGATATT", cs.g.getColorForSampleName("mom")))
            assert mom_walk == haps["mom"][0][haps["mom"][0].index("ATTATGA"):]       # starts at the source, runs to the end of the haplotype
        e.close()


def case_fill_gaps_random(orc, lib, tmp, seed):
    """fillGaps on a child walk over a three-colour graph: every parental allele next to a child-only stretch is recovered — after the
    fill, a walk in a parent's colour from a shared k-mer spells that parent's haplotype across the gap"""
    from corticall_amd import traversal_utils as tu
    rng = random.Random(4400 + seed)
    k = rng.choice([15, 17, 21])
    mom = rand_seq(rng, 400)
    kid = list(mom)
    sites = sorted(rng.sample(range(3 * k, len(mom) - 3 * k, 4 * k), 3))
    for p in sites:
        kid[p] = rng.choice([b for b in "ACGT" if b != kid[p]])
    kid = "".join(kid)
    cs = Case(orc, tmp, lib, [("mom", [mom]), ("kid", [kid])], k, link_samples=["mom", "kid"], name="fg%d" % seed)
    colors = [0, 1]
    kc, mc = cs.g.getColorForSampleName("kid"), cs.g.getColorForSampleName("mom")
    e = (TraversalEngineFactory(lib=lib).traversalColors(kc).traversalDirection(BOTH).combinationOperator(OR).stoppingRule("ContigStopper")
         .graph(cs.g).links(cs.links["kid"]).make())
    p0 = sites[1]
    w = e.walk(kid[p0 - 2:p0 - 2 + k])
    assert tu.toContig(w) == kid
    filled = tu.fillGaps(tu.toGraph(w, colors), cs.g, [cs.links["mom"], cs.links["kid"]], colors, relabel=True)
    assert tu.toContig(tu.toWalk(filled, kid[:k], kc)) == kid
    assert tu.toContig(tu.toWalk(filled, mom[:k], mc)) == mom
    e.close()


def test_ref_multiple_traversal_colors(orc, lib, tmp):           # TraversalEngineTest.java:160-208 (incl. PathFinder.getPaths, PathFinder.java:18-83)
    from corticall_amd.traversal_utils import PathFinder, Pseudograph, java_string_set_order
    haps = [("mom", ["AGTTCTGATCTAGGCTATATGCT"]), ("dad", ["AGTTCTGATCTGGGCTATATGCT"]), ("kid", ["AGTTCTG", "ATGGCTA"])]
    cs = Case(orc, tmp, lib, haps, 5, name="v_multi")
    f = TraversalEngineFactory(lib=lib).combinationOperator(AND).traversalDirection(BOTH).stoppingRule(ContigStopper).graph(cs.g)
    samples = []
    for sample in ("kid", "dad"):
        samples.append(sample)
        # g.getColorsForSampleNames(HashSet<String>): colours in the set's iteration order, ADDED to the factory's LinkedHashSet of colours
        e = f.traversalColors([cs.g.getColorForSampleName(x) for x in java_string_set_order(samples)]).make()
        contig, _ = e.walk_batch(["GTTCT"])
        assert contig[0] == dict(haps)[sample][0]
        oe = orc.Engine(cs.og, list(f._trav), op_and=True, stopper="ContigStopper")
        assert oe.walk("GTTCT")[0] == contig[0]
        e.close()
    samples.append("mom")
    e = f.traversalColors([cs.g.getColorForSampleName(x) for x in java_string_set_order(samples)]).stoppingRule("ExplorationStopper").make()
    assert f._trav[0] == cs.g.getColorForSampleName("kid")          # the edges of the dfs graph carry the FIRST traversal colour: the kid's
    d = e.dfs("AGTTC", "ATGCT")
    pg = Pseudograph.fromDfsGraph(d)
    v0 = next(v for v in pg.vertexSet() if v.getKmerAsString() == "AGTTC")      # TraversalUtils.findVertex
    v1 = next(v for v in pg.vertexSet() if v.getKmerAsString() == "ATGCT")
    gps = PathFinder(pg, cs.g.getColorForSampleName("kid")).getPaths(v0, v1)
    assert len(gps) == 2
    contigs = [TraversalUtils.toContig(gp.getVertexList()) for gp in gps]
    assert sorted(contigs) == sorted([haps[0][1][0], haps[1][1][0]])
    assert PathFinder(pg, cs.g.getColorForSampleName("mom")).getPaths(v0, v1) == []      # no edge carries that colour: an empty graph
    # constraint (a canonical k-mer on one path only) with accept / reject
    only_mom = orc.canonical("CTAGG")
    pf = PathFinder(pg, cs.g.getColorForSampleName("kid"))
    acc = [TraversalUtils.toContig(gp.getVertexList()) for gp in pf.getPaths(v0, v1, only_mom, True)]
    rej = [TraversalUtils.toContig(gp.getVertexList()) for gp in pf.getPaths(v0, v1, only_mom, False)]
    assert acc == [haps[0][1][0]] and rej == [haps[1][1][0]]
    assert pf.getPath(v0, v1).getWeight() == float(len(haps[0][1][0]) - 5)
    e.close()


def test_ref_dfs_with_sinks(orc, lib, tmp):                       # TraversalEngineTest.java:389-410
    hap = "GTGTGCTAGGTCTATAGTTATAGGCGCGTCTCCGCAAAAATCGT"
    cs = Case(orc, tmp, lib, [("test", [hap])], 5, link_samples=["test"], name="v12")
    e = (TraversalEngineFactory(lib=lib).traversalColors(0).traversalDirection(BOTH).combinationOperator(OR)
         .graph(cs.g).links(cs.links["test"]).make())
    g = e.dfs(hap[:5], hap[-5:])
    assert g.walk_contig(hap[:5], 0) == hap


def case_big_link_stores(orc, lib, tmp):
    """reads that thread a tandem repeat from every third position: hundreds of live links per walk — the link store
    spills from LDS to HBM, overflows its first capacity (the host retries with a larger one) and the wave-cooperative
    scans of the walk kernel run over several rounds"""
    for seed in range(3):
        rng = random.Random(50 + seed)
        unit = rand_seq(rng, 9)
        g1 = rand_seq(rng, 60) + unit * 6 + rand_seq(rng, 60) + unit * 3 + rand_seq(rng, 40)
        reads = {"a": [g1[i:] for i in range(0, len(g1) - 30, 3)]}
        cs = Case(orc, tmp, lib, [("a", [g1])], 5, link_samples=["a"], reads=reads, name="big%d" % seed)
        seeds = cs.all_kmers()[:40]
        compare_walks(cs, seeds, trav=[0], links=["a"], max_len=400)
        compare_dfs(cs, seeds[:10], trav=[0], stopper="ExplorationStopper", links=["a"], max_len=200)


# ------------------------------------------------------------------ FindTips (the other seed loop around the walks)
class JavaHashMap:
    """java.util.HashMap as far as iteration order goes: an array of buckets (chains in insertion order), index (h ^ h >>> 16) & (n - 1),
    doubled when the size passes 0.75 n — every chain split in two without reordering (HashMap.resize)"""

    def __init__(self):
        self.table, self.size, self.vals = [[] for _ in range(16)], 0, {}

    @staticmethod
    def _spread(h):
        return (h ^ (h >> 16)) & 0xFFFFFFFF

    def put(self, key, h, val):
        if key not in self.vals:
            self.table[self._spread(h) & (len(self.table) - 1)].append((key, h))
            self.size += 1
            if self.size > len(self.table) * 3 // 4:
                new = [[] for _ in range(2 * len(self.table))]
                for chain in self.table:
                    for kh in chain:
                        new[self._spread(kh[1]) & (len(new) - 1)].append(kh)
                self.table = new
        self.vals[key] = val

    def keys(self):
        return [kh[0] for chain in self.table for kh in chain]


def findtips_reference(orc, og, oroi, olinks, parents, k):
    """J/commands/prefilter/FindTips.java:30-137 restated over the oracle engine, one seed at a time -> (numTipChains, tip k-mers)"""
    child = og.color_for_sample_name(oroi.sample_name(0))
    pcols = [og.color_for_sample_name(p) for p in parents]
    oe = orc.Engine(og, [child], links=olinks, rois=oroi, joining_colors=pcols, op_and=True, stopper="ContigStopper")
    used = JavaHashMap()
    for i in range(oroi.N):
        ck = oroi.record_string(i).split()[0]
        used.put(ck, (orc.jhash_bytes(ck)) & 0xFFFFFFFF, False)

    def degree(sk, prev):
        idx = og.find(sk)[0]
        if idx < 0:
            return 0
        edges = og.record_string(idx).split()[1 + og.C + child]
        ins, outs = sum(1 for c in edges[:4] if c != "."), sum(1 for c in edges[4:] if c != ".")
        return (outs if prev else ins) if orc.is_flipped(sk) else (ins if prev else outs)

    tips, chains = set(), 0
    for rr in used.keys():
        if used.vals[rr]:
            continue
        contig, nv = oe.walk(rr)
        if nv == 0:
            continue
        w = [contig[j:j + k] for j in range(len(contig) - k + 1)]
        canon = [orc.canonical(sk) if og.find(sk)[0] >= 0 else None for sk in w]
        left = canon[0] in used.vals and degree(w[0], True) == 0
        right = canon[-1] in used.vals and degree(w[-1], False) == 0
        if left or right:
            chains += 1
        for c in canon:
            if c in used.vals:
                used.vals[c] = True
                if left or right:
                    tips.add(c)
    return chains, tips


def case_findtips(orc, lib, tmp, k, seed, with_links):
    from corticall_amd.partition import FindTips
    rng = random.Random(9100 + seed * 17 + k)
    base = genome_with_repeats(rng, 1800, n_rep=4, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
    kid = list(base)
    for _ in range(8):                                   # de novo mutations: chains of child-only k-mers anchored at both ends
        p = rng.randrange(2 * k, len(kid) - 2 * k)
        kid[p] = rng.choice([b for b in "ACGT" if b != kid[p]])
    kid = "".join(kid)
    # tips: child-only sequence hanging off the genome (an error at the end of a read), in both orientations, and a free-standing one
    extra = [kid[p:p + k + 5] + rand_seq(rng, rng.randint(2, k)) for p in rng.sample(range(100, len(kid) - 200), 4)]
    extra += [orc.revcomp(rand_seq(rng, rng.randint(2, k)) + kid[p:p + k + 5]) for p in rng.sample(range(100, len(kid) - 200), 3)]
    extra.append(rand_seq(rng, 2 * k + 7))
    dad = mutate(rng, base, snv=0.01, indel=0.002)
    reads = {"kid": [kid[i:i + 4 * k] for i in range(0, len(kid) - 4 * k, k)] + [kid[-4 * k:]] + extra} if with_links else None
    cs = Case(orc, tmp, lib, [("kid", [kid] + extra), ("mom", [base]), ("dad", [dad])], k, link_samples=(["kid"] if with_links else []), reads=reads,
              name="tips%d_%d_%d" % (k, seed, int(with_links)))
    parents = set()
    for h in (base, dad):
        parents |= {orc.canonical(h[i:i + k]) for i in range(len(h) - k + 1)}
    novel = [h[i:i + k] for h in [kid] + extra for i in range(len(h) - k + 1) if orc.canonical(h[i:i + k]) not in parents]
    assert len(novel) > 8
    roi_path = str(tmp / "roi.ctx")
    orc.build_graph(roi_path, [("kid", novel)], k)
    oroi, roi = orc.Graph(roi_path, tuned=True), CortexGraph(roi_path, lib=lib)
    exp_chains, exp_tips = findtips_reference(orc, cs.og, oroi, [cs.olinks["kid"]] if with_links else [], ["mom", "dad"], k)
    ft = FindTips(cs.g, roi, ["mom", "dad"], [cs.links["kid"]] if with_links else [])
    out_path = str(tmp / "tips.ctx")
    got_chains, got_n = ft.execute(out_path)
    got_tips = {roi.getRecord(i).getKmerAsString() for i in ft.tips}
    assert (got_chains, got_tips) == (exp_chains, exp_tips), (got_chains, exp_chains, sorted(got_tips ^ exp_tips))
    assert exp_chains >= 1 and 0 < len(exp_tips) <= oroi.N
    # the graph written: the ROI header, the tip records in ROI order
    tg = CortexGraph(out_path, lib=lib)
    assert tg.getNumRecords() == got_n and tg.getKmerSize() == k and tg.getNumColors() == roi.getNumColors() and tg.getSampleName(0) == roi.getSampleName(0)
    for j, i in enumerate(ft.tips):
        a, b = tg.getRecord(j), roi.getRecord(i)
        assert a.getKmerAsString() == b.getKmerAsString() and list(a.getCoverages()) == list(b.getCoverages()) and list(a.getEdges()) == list(b.getEdges())
    tg.close(); roi.close(); oroi.close()


# ------------------------------------------------------------------ façade members beside walk / dfs(source, sinks)
def case_facade(orc, lib, tmp, k, seed, with_links):
    """getNextVertices / getPrevVertices (TraversalEngine.java:147-239: vertices in HashSet iteration order), assemble (:112-145) and
    dfs(Collection<String> sources, Collection<String> sinks) (:37-62: Graphs.addGraph merges in source order) against the oracle"""
    rng = random.Random(7700 + 31 * seed + k)
    base = genome_with_repeats(rng, 1500, n_rep=5, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
    kid = mutate(rng, base, snv=0.015, indel=0.003)
    dad = mutate(rng, base, snv=0.02, indel=0.003)
    cs = Case(orc, tmp, lib, [("kid", [kid]), ("mom", [base]), ("dad", [dad])], k, link_samples=(["kid"] if with_links else []),
              reads={"kid": [kid[i:i + 4 * k] for i in range(0, max(1, len(kid) - 4 * k), k)] + [kid[-4 * k:]]} if with_links else None,
              name="fac%d_%d_%d" % (k, seed, int(with_links)))
    kmers = cs.all_kmers()
    qs = rng.sample(kmers, min(len(kmers), 250))
    qs = [q if rng.random() < 0.5 else orc.revcomp(q) for q in qs] + [rand_seq(rng, k) for _ in range(10)] + ["N" * k, kid[:k], kid[-k:]]
    for trav, recruit in (([0], ()), ([1, 0], ()), ([2], (0, 1))):
        oe = orc.Engine(cs.og, trav, links=[cs.olinks["kid"]] if with_links else [], recruitment_colors=recruit, stopper="ContigStopper")
        f = TraversalEngineFactory(lib=lib).traversalColors(*trav).graph(cs.g).stoppingRule(ContigStopper)
        if recruit:
            f.recruitmentColors(*recruit)
        if with_links:
            f.links(cs.links["kid"])
        e = f.make()
        use = [q for q in qs if not recruit or cs.og.find(q)[0] >= 0]       # (recruitment colours + a k-mer without a record: NullPointerException, below)
        for fwd in (True, False):
            got = e.neighbours_batch(use, fwd)
            for q, vs in zip(use, got):
                exp = oe.next_vertices(q) if fwd else oe.prev_vertices(q)
                have = [(v.getKmerAsString(), v.getCortexRecord().index if v.getCortexRecord() is not None else -1) for v in vs]
                assert have == exp, (trav, recruit, fwd, q, have, exp)
        assert [v.getKmerAsString() for v in e.getNextVertices(use[0])] == [x[0] for x in oe.next_vertices(use[0])]
        assert e.neighbours_batch([], True) == []
        if recruit:
            with pytest.raises(JavaNullPointerException):
                e.getPrevVertices(rand_seq(rng, k))
            with pytest.raises(orc.OracleError):
                oe.prev_vertices(rand_seq(rng, k))
        e.close()
    # assemble: with the cursor's link store when links are bound; maxBranchLength cuts both directions
    for max_len in (75000, 7):
        oe = orc.Engine(cs.og, [0], links=[cs.olinks["kid"]] if with_links else [], stopper="ContigStopper", max_length=max_len)
        f = TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).stoppingRule(ContigStopper).maxBranchLength(max_len)
        if with_links:
            f.links(cs.links["kid"])
        e = f.make()
        for sd in rng.sample(kmers, 12) + [orc.revcomp(kmers[3]), rand_seq(rng, k)]:
            exp = oe.assemble(sd, max_len)
            got = [(v.getKmerAsString(), v.getCortexRecord().index if v.getCortexRecord() is not None else -1) for v in e.assemble(sd)]
            assert got == exp, (max_len, sd, len(got), len(exp))
        e.close()
    # dfs over collections: sources a few hundred bases apart towards common sinks; the merged graph vertex by vertex, edge by edge
    # (ContigStopper at 6,000: on a short cycle the reference's copy-index search (TraversalEngine.java:391-402) is quadratic in the
    # branch length, and so is the oracle's — at 75,000 one seed in a few hundred takes the checker half an hour)
    for stopper, direction, max_len in (("DestinationStopper", FORWARD, 300), ("ExplorationStopper", BOTH, 60), ("ContigStopper", BOTH, 6000)):
        oe = orc.Engine(cs.og, [0], links=[cs.olinks["kid"]] if with_links else [], stopper=stopper, direction=direction, max_length=max_len)
        f = TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).stoppingRule(stopper).traversalDirection(direction).maxBranchLength(max_len)
        if with_links:
            f.links(cs.links["kid"])
        e = f.make()
        for _ in range(4):
            p0 = rng.randrange(0, len(kid) - 8 * k - 200)
            sources = [kid[p0 + d:p0 + d + k] for d in (0, k // 2, 3 * k, 3 * k)] + [rand_seq(rng, k)]      # overlapping searches, a repeated source, one that returns null
            sinks = [kid[p0 + 5 * k + 40:p0 + 6 * k + 40], kid[p0 + 7 * k + 90:p0 + 8 * k + 90]]
            r = oe.dfs_collection(sources, sinks)
            g = e.dfs(sources, sinks)
            assert (g is None) == r.is_null, (stopper, sources)
            if g is not None:
                assert g.vertex_tuples() == r.vertices() and g.edge_tuples() == r.edges(), (stopper, p0, g.nv, r.nv)
            r.free()
        lone = [rand_seq(rng, k)]
        r = oe.dfs_collection(lone, [])
        g = e.dfs(lone, [])
        assert (g is None) == r.is_null and (g is None or (g.vertex_tuples() == r.vertices() and g.edge_tuples() == r.edges()))
        r.free()
        assert e.dfs([], sinks) is None
        e.close()


# ------------------------------------------------------------------ Partition (the seed loop around the walks)
def partition_reference(orc, og, oroi, olinks, k):
    """J/commands/discover/call/Partition.java:57-219 restated over the oracle engine, one seed at a time"""
    color = og.color_for_sample_name(oroi.sample_name(0))
    oe = orc.Engine(og, [color], links=olinks, rois=oroi, stopper="ContigStopper")
    keys = [oroi.record_string(i).split()[0] for i in range(oroi.N)]
    used = {ck: None for ck in keys}
    for ck in keys:
        if used[ck] is not None:
            continue
        contig, _ = oe.walk(ck)
        w = [contig[j:j + k] for j in range(len(contig) - k + 1)] if contig else [ck]
        canon = []
        for sk in w:
            assert og.find(sk)[0] >= 0, "vertex without a record: the reference throws here"
            canon.append(orc.canonical(sk))
        for c in canon:
            if c in used and (used[c] is None or len(w) > len(used[c][0])):
                used[c] = (w, contig if contig else ck)
    contigs = set()
    for ck in keys:
        if used[ck] is not None:
            fw = used[ck][1]
            if fw not in contigs and orc.revcomp(fw) not in contigs:
                contigs.add(fw)
    out = []
    for num, part in enumerate(sorted(contigs)):
        nn = sum(1 for j in range(len(part) - k + 1) if orc.canonical(part[j:j + k]) in used)
        out += [">partition%d len=%d numNovels=%d" % (num, len(part) - k + 1, nn), part]
    return "\n".join(out) + ("\n" if out else "")


def case_partition(orc, lib, tmp, k, seed, with_links):
    from corticall_amd.partition import Partition
    rng = random.Random(7000 + seed * 31 + k)
    base = genome_with_repeats(rng, 1500, n_rep=5, rep_len=(k // 2 + 1, 3 * k), copies=(2, 3))
    kid = list(base)
    for _ in range(6):                                   # de novo mutations: child-only k-mers around them
        p = rng.randrange(k, len(kid) - k)
        kid[p] = rng.choice([b for b in "ACGT" if b != kid[p]])
    kid = "".join(kid)
    dad = mutate(rng, base, snv=0.01, indel=0.002)
    reads = {"kid": [kid[i:i + 4 * k] for i in range(0, len(kid) - 4 * k, k)] + [kid[-4 * k:]]} if with_links else None
    cs = Case(orc, tmp, lib, [("kid", [kid]), ("mom", [base]), ("dad", [dad])], k, link_samples=(["kid"] if with_links else []), reads=reads,
              name="part%d_%d_%d" % (k, seed, int(with_links)))
    parents = set()
    for h in (base, dad):
        parents |= {orc.canonical(h[i:i + k]) for i in range(len(h) - k + 1)}
    novel = [kid[i:i + k] for i in range(len(kid) - k + 1) if orc.canonical(kid[i:i + k]) not in parents]
    assert len(novel) > 3
    roi_path = str(tmp / "roi.ctx")
    orc.build_graph(roi_path, [("kid", novel)], k)
    oroi, roi = orc.Graph(roi_path, tuned=True), CortexGraph(roi_path, lib=lib)
    exp = partition_reference(orc, cs.og, oroi, [cs.olinks["kid"]] if with_links else [], k)
    got = Partition(cs.g, roi, [cs.links["kid"]] if with_links else []).execute()
    assert got == exp, (got, exp)
    assert exp.count(">partition") >= 1
    roi.close(); oroi.close()


# ------------------------------------------------------------------ link file formats (L4): .ctp.gz versions 2, 3 and 4
def _write_ctp(path, version, k, records, two_colours=False):
    """records: [(kmer, [(F|R, num_kmers, junctions, [cov per colour])...])] as McCortex / IndexLinks would write them"""
    import gzip
    import json
    ncol = 2 if two_colours else 1
    cols = [{"colour": c, "sample": "s%d" % c, "total_sequence": 1000 + c, "cleaned_tips": bool(c)} for c in range(ncol)]
    nlinks = sum(len(js) for _, js in records)
    if version == 2:
        hdr = {"format_version" if two_colours else "formatVersion": 2, "ncols": ncol, "kmer_size": k, "num_kmers_in_graph": 77,
               "num_kmers_with_paths": len(records), "num_paths": nlinks, "path_bytes": 99, "colours": cols}
    else:
        hdr = {"file_format": "ctp", "formatVersion": version,
               "graph": {"num_colours": ncol, "kmer_size": k, "num_kmers_in_graph": 77, "colours": cols},
               "paths": {"num_kmers_with_paths": len(records), "num_paths": nlinks, "path_bytes": 99}}
    text = json.dumps(hdr, indent=2).replace("{\n", "{\n", 1)
    lines = ["{"] + text.split("\n")[1:-1] + ["}", "", "# comment line", "# kmer num_links", ""]
    for kmer, js in records:
        lines.append("%s %d" % (kmer, len(js)))
        for orient, nk, junc, cov in js:
            covs = ",".join(str(c) for c in cov)
            if version == 4:
                lines.append("%s %d %s %s" % (orient, len(junc), covs, junc))
            else:
                lines.append("%s %d %d %s %s" % (orient, nk, len(junc), covs, junc))
    with gzip.open(path, "wt") as f:
        f.write("\n".join(lines) + "\n")


def _check_link_index(bgz_path, k, recs, two, source, orc):
    """the LNKIDX file as CortexLinksRandomAccess reads it (CortexLinksRandomAccess.java:33-89): big-endian header between two magic words,
    then per k-mer (in k-mer string order) the binary k-mer, the BGZF virtual offset and the text length; every record is fetched here
    through its virtual offset with nothing but zlib (block at offset >> 16, byte offset & 0xFFFF inside its uncompressed data)"""
    import struct
    import zlib
    raw = open(bgz_path + ".idx", "rb").read()
    assert raw[:6] == b"LNKIDX"
    ncol, kk, nkg, nkl, lb = struct.unpack(">iiqqq", raw[6:38])
    assert (ncol, kk, nkg, nkl, lb) == (2 if two else 1, k, 77, len(recs), 99)
    p = 38
    (sl,) = struct.unpack(">i", raw[p:p + 4]); p += 4
    assert raw[p:p + sl].decode() == source; p += sl
    for c in range(ncol):
        (n,) = struct.unpack(">i", raw[p:p + 4]); p += 4
        assert raw[p:p + n].decode() == "s%d" % c; p += n
    assert raw[p:p + 6] == b"LNKIDX"; p += 6
    W = (k + 31) // 32
    bgz = open(bgz_path, "rb").read()
    assert bgz[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")      # the BGZF end-of-file marker

    def block(addr):
        assert bgz[addr:addr + 4] == b"\x1f\x8b\x08\x04" and bgz[addr + 12:addr + 14] == b"BC"
        bsize = struct.unpack("<H", bgz[addr + 16:addr + 18])[0] + 1
        return zlib.decompress(bgz[addr + 18:addr + bsize - 8], -15), bsize
    by_kmer = {km: js for km, js in recs}
    seen = []
    for i in range(nkl):
        words = struct.unpack("<%dQ" % W, raw[p:p + 8 * W]); p += 8 * W
        voff, ln = struct.unpack(">qi", raw[p:p + 12]); p += 12
        addr, off = voff >> 16, voff & 0xFFFF
        text = b""
        while len(text) < ln:
            data, bsize = block(addr)
            text += data[off:off + ln - len(text)]
            addr, off = addr + bsize, 0
        lines = text.decode().split("\n")
        km, n = lines[0].split()
        assert list(words) == orc.encode_kmer(orc.canonical(km)) and len(lines) == 1 + int(n)
        want = {(o, len(j), ",".join(str(x) for x in cov), j) for o, _, j, cov in by_kmer[km]}
        assert {tuple(l.split()) for l in lines[1:]} == {(o, str(n_), c_, j) for o, n_, c_, j in want}
        seen.append(km)
    assert p == len(raw) and seen == sorted(by_kmer)                    # TreeMap<CortexByteKmer, ...> order


def case_link_formats(orc, lib, tmp):
    """the three header dialects (CortexLinksIterable.java:69-123) and record layouts (:172-226): header fields, record
    lookup in either orientation, junction records in the reference's HashSet order — product vs oracle, then a walk"""
    rng = random.Random(77)
    k = 7
    g1 = rand_seq(rng, 140)
    cs = Case(orc, tmp, lib, [("s0", [g1]), ("s1", [mutate(rng, g1, snv=0.03)])], k, name="lf")
    kmers = cs.all_kmers()
    for version in (2, 3, 4):
        for two in (False, True):
            recs = []
            for km in rng.sample(kmers, 12):
                js = []
                for _ in range(rng.randint(1, 4)):
                    junc = rand_seq(rng, rng.randint(1, 9))
                    js.append((rng.choice("FR"), rng.randint(2, 30), junc, [rng.randint(1, 9) for _ in range(2 if two else 1)]))
                recs.append((km if rng.random() < 0.5 else orc.revcomp(km), js))
            recs.sort(key=lambda r: orc.canonical(r[0]))
            p = str(tmp / ("v%d_%d.ctp.gz" % (version, int(two))))
            _write_ctp(p, version, k, recs, two)
            ol, l = orc.Links(p), CortexLinks(p, cs.g)
            assert (l.version, l.numColors, l.kmerSize, l.numKmersInGraph, l.numKmersWithLinks, l.numLinks) == (version, 2 if two else 1, k, 77, len(recs), sum(len(j) for _, j in recs))
            assert l.getSampleNameForColor(0) == "s0"
            exp = dict(ol.records())
            for km, js in recs:
                cov_of = {(o, j): c for o, _, j, c in js}
                for q in (km, orc.revcomp(km)):
                    found, got = l.get(q)
                    assert found
                    # junction records in the reference's HashSet iteration order, with their coverages
                    assert [(x[0], x[3]) for x in got] == [(j[0] == "F", j[1]) for j in exp[km]], (version, km, got, exp[km])
                    for x in got:
                        assert x[1] == len(x[3]) and x[2] == cov_of[("F" if x[0] else "R", x[3])]
            assert not l.containsKey("A" * k) or "A" * k in {orc.canonical(r[0]) for r in recs}
            # and the links drive a walk identically
            cs.olinks["s0"], cs.links["s0"] = ol, l
            compare_walks(cs, kmers[:25], trav=[0], links=["s0"], max_len=60)
            # IndexLinks (IndexLinks.java:62-135) on this file: a BGZF copy + the big-endian LNKIDX index; CortexLinks then picks the
            # random-access back-end (CortexLinksRandomAccess), whose records hash differently (quirk Q11) -> possibly another order of
            # the junction records, same content.  The index is checked byte by byte against the reference's reader's view of it.
            pb = str(tmp / ("v%d_%d.ctp.bgz" % (version, int(two))))
            assert CortexLinks.index(p, pb, "src %d" % version, lib=lib) == len(recs)
            _check_link_index(pb, k, recs, two, "src %d" % version, orc)
            oli, li = orc.Links(pb), CortexLinks(pb, cs.g)
            assert li.getSource() == "src %d" % version and l.getSource() == "unknown"
            assert (li.numColors, li.kmerSize, li.numKmersInGraph, li.numKmersWithLinks, li.numLinks) == (2 if two else 1, k, 77, len(recs), sum(len(j) for _, j in recs))
            assert li.getSampleNameForColor(0) == "s0"
            expi = dict(oli.records())
            for km, js in recs:
                for q in (km, orc.revcomp(km)):
                    found, got = li.get(q)
                    assert found and [(x[0], x[3]) for x in got] == [(j[0] == "F", j[1]) for j in expi[km]], (version, km, got, expi[km])
                    assert sorted((x[0], x[3]) for x in got) == sorted((j[0] == "F", j[1]) for j in exp[km])
            cs.olinks["s0"], cs.links["s0"] = oli, li
            compare_walks(cs, kmers[:25], trav=[0], links=["s0"], max_len=60)
            li.close()
            l.close()           # gives the graph's flag bit back: more than 6 link sets pass through this graph


def case_sort(orc, lib, tmp):
    """Sort.java:20-49: Arrays.sort with CortexRecord.compareTo (k-mer strings) is a stable merge sort — shuffled files, with
    duplicate k-mers kept in input order, must come back byte for byte as Python's stable sort on the k-mer strings gives"""
    from corticall_amd.distributed import ctx_header
    from corticall_amd.partition import Sort, unpack_kmers
    rng = random.Random(31)
    for k, ncol, n_bp in ((5, 1, 60), (31, 2, 3000), (47, 3, 5000), (65, 1, 2000)):
        src = str(tmp / ("sorted%d.ctx" % k))
        orc.build_graph(src, [("s%d" % c, [rand_seq(rng, n_bp)]) for c in range(ncol)], k)
        raw = np.fromfile(src, dtype=np.uint8)
        h = ctx_header(raw)
        rec = 8 * h["W"] + 5 * h["C"]
        body = raw[h["data_offset"]:].reshape(-1, rec)
        dup = body[rng.sample(range(len(body)), min(7, len(body)))].copy()
        dup[:, -1] ^= 0x5A                                   # same k-mers, different payload: stability is visible
        shuffled = np.concatenate([body, dup])[np.random.default_rng(k).permutation(len(body) + len(dup))]
        unsorted = str(tmp / ("unsorted%d.ctx" % k))
        np.concatenate([raw[:h["data_offset"]], shuffled.reshape(-1)]).tofile(unsorted)
        out = str(tmp / ("resorted%d.ctx" % k))
        assert Sort(unsorted, out, lib=lib).execute() == len(shuffled)
        words = np.ascontiguousarray(shuffled[:, :8 * h["W"]]).view("<u8").reshape(-1, h["W"])
        kmers = [x.tobytes().decode() for x in unpack_kmers(words, k)]
        order = sorted(range(len(kmers)), key=lambda i: kmers[i])          # stable, like Arrays.sort on objects
        expected = np.concatenate([raw[:h["data_offset"]], shuffled[order].reshape(-1)])
        assert (np.fromfile(out, dtype=np.uint8) == expected).all()
        if k != 5:      # and without the duplicates the result loads as a graph again
            clean = str(tmp / ("clean%d.ctx" % k))
            np.concatenate([raw[:h["data_offset"]], body[np.random.default_rng(1).permutation(len(body))].reshape(-1)]).tofile(clean)
            Sort(clean, out, lib=lib).execute()
            assert (np.fromfile(out, dtype=np.uint8) == raw).all()
            CortexGraph(out, lib=lib).close()


def java_read_header(raw):
    """CortexGraph.loadCortexGraph (:66-168): the values the reader keeps — total_sequence read big-endian (quirk Q16),
    names cut at their first NUL, the error rate skipped -> (k, W, [colour dicts])"""
    import struct
    ver, k, W, C = struct.unpack_from("<IIII", raw, 6)
    p = 22
    mrl = struct.unpack_from("<%dI" % C, raw, p); p += 4 * C
    tot = struct.unpack_from(">%dQ" % C, raw, p); p += 8 * C
    names = []
    for _ in range(C):
        (ln,) = struct.unpack_from("<I", raw, p); p += 4
        names.append(bytes(raw[p:p + ln]).split(b"\0")[0]); p += ln
    p += 16 * C
    cols = []
    for c in range(C):
        fl = bytes(raw[p:p + 4]); t1, t2, ln = struct.unpack_from("<III", raw, p + 4); p += 16
        cols.append(dict(mrl=mrl[c], tot=tot[c], name=names[c], flags=bytes(1 if b else 0 for b in fl), t1=t1, t2=t2,
                         cleaned=bytes(raw[p:p + ln]).split(b"\0")[0])); p += ln
    return k, W, cols


def java_write_header(k, W, cols):
    """CortexGraphWriter.initialize (:40-113)"""
    import struct
    C = len(cols)
    o = b"CORTEX" + struct.pack("<IIII", 6, k, W, C) + struct.pack("<%dI" % C, *[c["mrl"] for c in cols])
    o += struct.pack("<%dQ" % C, *[c["tot"] for c in cols])
    for c in cols:
        o += struct.pack("<I", len(c["name"])) + c["name"]
    o += bytes([0, 0xd8, 0xa3, 0x70, 0x3d, 0x0a, 0xd7, 0xa3, 0xf8, 0x3f, 0, 0, 0, 0, 0, 0]) * C
    for c in cols:
        o += c["flags"] + struct.pack("<III", c["t1"], c["t2"], len(c["cleaned"])) + c["cleaned"]
    return o + b"CORTEX"


def java_rewritten_header(raw):
    """what the reference's reader + writer pair makes of a header: values, not bytes"""
    return java_write_header(*java_read_header(raw))


def case_sort_rewrites_header(orc, lib, tmp):
    """a McCortex-like header (total_sequence set, another error rate, a name padded with NULs) comes out of Sort the way the
    reference's reader + writer pair would leave it"""
    import struct
    from corticall_amd.distributed import ctx_header
    from corticall_amd.partition import Sort
    rng = random.Random(41)
    src = str(tmp / "plain.ctx")
    orc.build_graph(src, [("a", [rand_seq(rng, 400)]), ("b", [rand_seq(rng, 300)])], 21)
    raw = np.fromfile(src, dtype=np.uint8)
    h = ctx_header(raw)
    body = raw[h["data_offset"]:]
    hdr = (b"CORTEX" + struct.pack("<IIII", 6, 21, 1, 2) + struct.pack("<II", 100, 76) + struct.pack("<QQ", 123456789, 2 ** 40 + 7)
           + struct.pack("<I", 6) + b"mom\0\0\0" + struct.pack("<I", 3) + b"kid" + bytes(range(1, 33))
           + bytes([1, 0, 2, 0]) + struct.pack("<III", 5, 0, 7) + b"ref.ctx" + bytes([0, 1, 0, 0]) + struct.pack("<III", 0, 3, 0) + b"CORTEX")
    rec = 8 + 10
    recs = body.reshape(-1, rec)
    shuffled = recs[np.random.default_rng(3).permutation(len(recs))]
    unsorted = str(tmp / "mccortex_like.ctx")
    np.concatenate([np.frombuffer(hdr, dtype=np.uint8), shuffled.reshape(-1)]).tofile(unsorted)
    out = str(tmp / "mccortex_like.sorted.ctx")
    assert Sort(unsorted, out, lib=lib).execute() == len(recs)
    got = np.fromfile(out, dtype=np.uint8).tobytes()
    exp_hdr = java_rewritten_header(np.frombuffer(hdr, dtype=np.uint8))
    assert got[:len(exp_hdr)] == exp_hdr and got[len(exp_hdr):] == body.tobytes()
    assert exp_hdr != hdr                                                   # the rewrite is visible (Q16, error rate, names)
    g = CortexGraph(out, lib=lib)
    assert g.getSampleName(0) == "mom" and g.getNumRecords() == len(recs)
    g.close()


def test_ref_collection(orc, lib, tmp):                          # T/utils/io/graph/collection/CortexCollectionTest.java:16-140 (all six tests)
    from corticall_amd import CortexCollection
    h1 = [("fred", ["ACCGTATGTA"]), ("wilma", ["ACCGCATGTA"]), ("barney", ["ACCGTATATA"])]
    h2 = [("pebbles", ["AACGTATGTA"]), ("bam-bam", ["ACTGCATGTA"])]
    p1, p2 = str(tmp / "cc1.ctx"), str(tmp / "cc2.ctx")
    orc.build_graph(p1, h1, 3)
    orc.build_graph(p2, h2, 3)
    g0, g1 = CortexGraph(p1, lib=lib), CortexGraph(p2, lib=lib)
    cc = CortexCollection(g0, g1)
    assert g0.getNumRecords() > 2 and g1.getNumRecords() > 2
    for c0 in g0:                                                   # testDynamicGraphMerging :38-67
        c1 = g1.findRecord(c0.getKmerAsString())
        cm = cc.findRecord(c0.getKmerAsString())
        assert cm.getKmerAsString() == c0.getKmerAsString()
        for c in range(3):
            assert cm.getCoverage(c) == c0.getCoverage(c) and cm.getEdgesAsString(c) == c0.getEdgesAsString(c)
        for c in range(2):
            if c1 is None:
                assert cm.getCoverage(3 + c) == 0 and cm.getEdgesAsString(3 + c) == "........"
            else:
                assert cm.getCoverage(3 + c) == c1.getCoverage(c) and cm.getEdgesAsString(3 + c) == c1.getEdgesAsString(c)
    single = CortexCollection(p1, lib=lib)                          # testSingleFileCollection :70-84
    for c0 in g0:
        cm = single.findRecord(c0.getKmerAsString())
        assert cm.getKmerAsString() == c0.getKmerAsString()
        for c in range(2):
            assert cm.getCoverage(c) == c0.getCoverage(c) and cm.getEdgesAsString(c) == c0.getEdgesAsString(c)
    single.close()
    kmers = sorted({r.getKmerAsString() for r in g0} | {r.getKmerAsString() for r in g1})     # testIteration :87-116 (TreeSet order)
    seen = 0
    for cr, esk in zip(cc, kmers):
        assert cr.getKmerAsString() == esk
        c0, c1 = g0.findRecord(esk), g1.findRecord(esk)
        if c0 is not None:
            assert cr.getCoverage(0) == c0.getCoverage(0) and cr.getEdgesAsString(0) == c0.getEdgesAsString(0)
        if c1 is not None:
            assert cr.getCoverage(3) == c1.getCoverage(0) and cr.getEdgesAsString(3) == c1.getEdgesAsString(0)
        seen += 1
    assert seen == len(kmers) == sum(1 for _ in cc)
    assert cc.getNumColors() == g0.getNumColors() + g1.getNumColors() == 5                   # testNumColors, testColorNames, testColors
    assert [cc.getSampleName(c) for c in range(5)] == [g0.getSampleName(0), g0.getSampleName(1), g0.getSampleName(2), g1.getSampleName(0), g1.getSampleName(1)]
    assert [cc.getColor(c) for c in range(5)] == [g0.getColor(0), g0.getColor(1), g0.getColor(2), g1.getColor(0), g1.getColor(1)]
    cc.close()


def case_collection(orc, lib, tmp):
    """CortexCollection.java:34-58, 160-188, 218-293: several graphs as one — the iterator's records (head-by-head merge), findRecord
    (one lookup per member, Q1 for a member of two records), the colour bookkeeping, and a traversal engine over the collection
    against the oracle over the joined file"""
    from corticall_amd import CortexCollection
    from corticall_amd.distributed import ctx_header
    from corticall_amd.partition import Join
    rng = random.Random(77)
    k = 21
    base = rand_seq(rng, 1500)
    specs = [[("kid", [mutate(rng, base, snv=0.01)])], [("mom", [base[:900]]), ("dad", [mutate(rng, base[300:], snv=0.02)])], [("kid", [base[100:100 + k + 1]])]]
    paths, parsed = [], []
    for gi, haps in enumerate(specs):
        p = str(tmp / ("c%d.ctx" % gi))
        orc.build_graph(p, haps, k)
        raw = np.fromfile(p, dtype=np.uint8)
        h = ctx_header(raw)
        parsed.append((h, raw[h["data_offset"]:].reshape(-1, 8 * h["W"] + 5 * h["C"])))
        paths.append(p)
    assert len(parsed[2][1]) == 2                      # the member findRecord never answers from (Q1)
    W, Ctot = parsed[0][0]["W"], sum(h["C"] for h, _ in parsed)

    def merged(members):
        out, off = {}, 0
        for mi, (h, recs) in enumerate(parsed):
            if mi in members:
                for r in recs:
                    cov, edges = out.setdefault(r[:8 * W].tobytes(), ([0] * Ctot, [0] * Ctot))
                    for c in range(h["C"]):
                        cov[off + c] = int.from_bytes(r[8 * W + 4 * c:8 * W + 4 * c + 4].tobytes(), "little")
                        edges[off + c] = int(r[8 * W + 4 * h["C"] + c])
            off += h["C"]
        return out

    def key_str(key):
        return orc.decode_kmer([int.from_bytes(key[8 * w:8 * w + 8], "little") for w in range(W)], k)

    it_view, find_view = merged({0, 1, 2}), merged({0, 1})
    members = [CortexGraph(p, lib=lib) for p in paths]
    col = CortexCollection(*members)
    assert col.getNumColors() == Ctot == 4 and col.getKmerSize() == k and col.getNumRecords() == 0 and col.getVersion() == 6
    assert [col.getSampleName(c) for c in range(4)] == ["kid", "mom", "dad", "kid"]
    assert col.getColorForSampleName("mom") == 1 and col.getColorForSampleName("kid") == -1 and col.getColorForSampleName("nobody") == -1
    assert col.getColorsForSampleNames(["kid", "dad"]) == [0, 2, 3]
    assert col.getGraph(2) is members[1] and col.hasColor(3) and not col.hasColor(4)
    for bad in (lambda: col.getFile(), lambda: col.position(), lambda: col.getRecord(0)):
        with pytest.raises(NotImplementedError):
            bad()
    rows = [(r.getKmerAsString(), [int(x) & 0xFFFFFFFF for x in r.getCoverages()], [int(x) for x in r.getEdges()]) for r in col]
    exp_rows = sorted((key_str(key), cov, edges) for key, (cov, edges) in it_view.items())
    assert rows == exp_rows and len(rows) > 1000
    tiny_only = [key_str(key) for key in it_view if key not in find_view]
    for ks, cov, edges in exp_rows[::7] + [r for r in exp_rows if r[0] in tiny_only]:
        for q in (ks, orc.revcomp(ks)):
            r = col.findRecord(q)
            if ks in tiny_only:
                assert r is None
            else:
                assert r is not None and r.getKmerAsString() == ks
    by_str = {key_str(key): v for key, v in find_view.items()}
    idx, cov, edges = col.find_batch([r[0] for r in exp_rows])
    for j, (ks, _, _) in enumerate(exp_rows):
        if ks in by_str:
            assert idx[j] >= 0 and [int(x) & 0xFFFFFFFF for x in cov[j]] == by_str[ks][0] and [int(x) for x in edges[j]] == by_str[ks][1]
        else:
            assert idx[j] < 0
    assert col.findRecord(rand_seq(rng, k)) is None
    # a traversal engine over the collection = the oracle over the file Join writes of the same graphs (no tiny member here)
    joined = str(tmp / "joined.ctx")
    Join(paths[:2], joined, lib=lib).execute()
    og = orc.Graph(joined, tuned=True)
    col2 = CortexCollection(paths[0], paths[1], lib=lib)
    seeds = [r[0] for r in exp_rows[::40]]
    for trav, stopper in (([0], "ContigStopper"), ([1, 2], "ContigStopper")):
        oe = orc.Engine(og, trav, stopper=stopper)
        e = TraversalEngineFactory(lib=lib).traversalColors(*trav).graph(col2).stoppingRule(stopper).make()
        got, _ = e.walk_batch(seeds)
        for s_, c in zip(seeds, got):
            assert c == oe.walk(s_)[0]
        assert max(len(c) for c in got) > 3 * k
        e.close()
    col2.close(); col.close(); og.close()


def case_join(orc, lib, tmp):
    """Join.java:16-60 over CortexCollection (:34-58 colours side by side, :218-293 head-by-head merge of the sorted files):
    the union of the k-mers, zero coverage / no edges where a file lacks the k-mer — byte for byte"""
    from corticall_amd.distributed import ctx_header
    from corticall_amd.partition import Join
    rng = random.Random(58)
    for k in (21, 47):
        base = rand_seq(rng, 1200)
        paths, parsed = [], []
        for gi, ncol in enumerate((1, 2, 1)):
            haps = [("s%d_%d" % (gi, c), [mutate(rng, base[rng.randint(0, 200):rng.randint(600, 1200)], snv=0.02)]) for c in range(ncol)]
            p = str(tmp / ("j%d_%d.ctx" % (k, gi)))
            orc.build_graph(p, haps, k)
            raw = np.fromfile(p, dtype=np.uint8)
            h = ctx_header(raw)
            rec = 8 * h["W"] + 5 * h["C"]
            parsed.append((raw, h, raw[h["data_offset"]:].reshape(-1, rec)))
            paths.append(p)
        W = parsed[0][1]["W"]
        Ctot = sum(h["C"] for _, h, _ in parsed)
        cols, merged, off = [], {}, 0
        for raw, h, recs in parsed:
            cols += java_read_header(raw)[2]
            for r in recs:
                key = r[:8 * W].tobytes()
                cov, edges = merged.setdefault(key, (bytearray(4 * Ctot), bytearray(Ctot)))
                cov[4 * off:4 * (off + h["C"])] = r[8 * W:8 * W + 4 * h["C"]].tobytes()
                edges[off:off + h["C"]] = r[8 * W + 4 * h["C"]:].tobytes()
            off += h["C"]
        def kmer_order(key):          # file order = k-mer order: words most significant first
            return tuple(int.from_bytes(key[8 * w:8 * w + 8], "little") for w in range(W))
        body = b"".join(key + bytes(merged[key][0]) + bytes(merged[key][1]) for key in sorted(merged, key=kmer_order))
        expected = java_write_header(k, W, cols) + body
        out = str(tmp / ("joined%d.ctx" % k))
        assert Join(paths, out, lib=lib).execute() == len(merged)
        assert np.fromfile(out, dtype=np.uint8).tobytes() == expected
        g = CortexGraph(out, lib=lib)
        assert g.getNumColors() == Ctot and g.getNumRecords() == len(merged) and g.getSampleName(1) == "s1_0"
        g.close()
    other = str(tmp / "otherk.ctx")
    orc.build_graph(other, [("x", [rand_seq(rng, 100)])], 31)
    try:
        Join([paths[0], other], str(tmp / "bad.ctx"), lib=lib).execute()
        assert False
    except ca.CortexJDKException as ex:
        assert "Graph kmer sizes are not equal" in str(ex)


def case_dfs_step_limit(orc, lib, tmp, monkeypatch):
    """a rule that ignores maxLength on a cycle with links never returns in the reference; the device path gives up after its step
    limit instead of spinning"""
    rng = random.Random(3)
    unit = rand_seq(rng, 9)
    g1 = rand_seq(rng, 40) + unit * 8 + rand_seq(rng, 40)
    cs = Case(orc, tmp, lib, [("a", [g1])], 5, link_samples=["a"], reads={"a": [g1[i:] for i in range(0, len(g1) - 30, 3)]}, name="spin")
    monkeypatch.setenv("LDBG_DFS_ITER_LIMIT", "5000")
    f = (TraversalEngineFactory(lib=lib).traversalColors(0).graph(cs.g).links(cs.links["a"]).stoppingRule("GapClosingStopper"))
    e = f.make()
    with pytest.raises(ca.LdbgError) as ex:
        e.dfs_batch(cs.all_kmers()[:20]).graphs()
    assert "step limit" in str(ex.value)


def case_close_in_any_order(orc, lib, tmp):
    """handles are closed by a garbage collector in any order: a graph may go before the link sets bound to it"""
    rng = random.Random(8)
    g1 = rand_seq(rng, 300)
    cs = Case(orc, tmp, lib, [("a", [g1])], 21, link_samples=["a"], name="order")
    links = cs.links["a"]
    again = CortexLinks(links.path, cs.g, lib=lib)
    again.close()                 # the usual order
    cs.g.close()
    links.close()                 # after its graph: nothing left to give back
    links.close()
