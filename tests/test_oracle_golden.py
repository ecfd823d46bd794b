"""Pins the CPU oracle (oracle/) against the reference's own known-answer tests,
SURVEY.md §8c vectors V1..V15.  Each test cites the reference test it restates
(T/ = public/java/tests/uk/ac/ox/well/cortexjdk/)."""
import math
import os
import random

import numpy as np
import pytest


def _table(golden_dir):
    rows = []
    for line in open(os.path.join(golden_dir, "two_short_contigs.expected.txt")):
        k, c0, c1, e0, e1 = line.split()
        rows.append((k, int(c0), int(c1), e0, e1))
    return rows


@pytest.fixture(scope="module")
def fixture_graph(orc, golden_dir):
    g = orc.Graph(os.path.join(golden_dir, "two_short_contigs.ctx"))
    yield g
    g.close()


# ---- V1: T/utils/kmer/CortexGraphTest.java:140-152, 187-198
def test_v1_header_and_iteration(orc, fixture_graph, golden_dir):
    g = fixture_graph
    assert (g.k, g.W, g.C, g.N, g.data_offset) == (31, 1, 2, 66, 148)
    assert g.sample_name(0) == "one" and g.sample_name(1) == "two"
    rows = _table(golden_dir)
    for i, (k, c0, c1, e0, e1) in enumerate(rows):
        assert g.record_string(i) == f"{k} {c0} {c1} {e0} {e1}"
    assert g.get_record(66) is None          # Q2
    assert [r[0] for r in rows] == sorted(r[0] for r in rows)


# ---- V2: CortexGraphTest.java:256-280, 311-331
def test_v2_get_find_encode(orc, fixture_graph, golden_dir):
    g = fixture_graph
    rows = _table(golden_dir)
    for i in range(10, -1, -1):
        assert g.record_string(i).split()[0] == rows[i][0]
    for i, row in enumerate(rows):
        idx, cov, _ = g.find(row[0])
        assert idx == i and cov == [row[1], row[2]]
        idx_rc, _, _ = g.find(orc.revcomp(row[0]))
        assert idx_rc == i
        assert orc.decode_kmer(orc.encode_kmer(row[0]), 31) == row[0]
        w, _, _ = g.get_record(i)
        assert w == orc.encode_kmer(row[0])
    assert g.find("NTTTTGGGGTATTTGCAGTATTTGGAATAAA")[0] == -1      # Q4
    # tuned lookup returns the same indices
    km = np.array([list(r[0].encode()) for r in rows], dtype=np.uint8)
    assert (g.find_batch(km, tuned=True) == np.arange(66)).all()


def _graph(orc, tmp_path, haps, k, name="g.ctx"):
    p = str(tmp_path / name)
    orc.build_graph(p, haps, k)
    return orc.Graph(p)


def _records(g):
    return [g.record_string(i) for i in range(g.N)]


# ---- V3: T/utils/traversal/TraversalEngineTest.java:49-62
def test_v3_arbitrary_graph_construction(orc, tmp_path):
    g = _graph(orc, tmp_path, [("mom", ["AATA"]), ("dad", ["AATG"])], 3)
    assert g.N == 3
    assert set(_records(g)) == {"AAT 1 1 ....A... ......G.", "ATA 1 0 a....... ........", "ATG 0 1 ........ a......."}


# ---- V4: TraversalEngineTest.java:65-95
V4 = """AGAAC 1 1 1 .c.....T .c.....T .c.....T
AGATC 1 1 1 .c..A... .c..A... .c..A...
AGCAT 1 1 1 ....A... ....A... ....A...
AGCCC 1 1 1 ...tA... ...tA... ...tA...
AGTTC 1 1 1 .......T .......T .......T
ATAGC 1 1 1 ...t.C.. ...t.C.. ...t.C..
ATATA 1 1 1 .c....G. .c....G. .c....G.
ATATG 1 1 1 ...t.C.. ...t.C.. ...t.C..
ATCAG 1 1 1 ..g.A... ..g.A... ..g.A...
ATCTG 1 1 1 ..g...G. ..g...G. ..g...G.
CAGAA 1 1 1 ...t.C.. ...t.C.. ...t.C..
CCAGA 1 1 1 .c.....T .c.....T .c.....T
CCCAG 1 1 1 ..g.A... ..g.A... ..g.A...
CTATA 1 1 1 ..g....T ..g....T ..g....T
GATCA 1 1 1 a.....G. a.....G. a.....G.
GCATA 1 1 1 a......T a......T a......T
GCCCA 1 1 1 a.....G. a.....G. a.....G.
GGCTA 1 1 1 ..g....T ..g....T ..g....T
TCAGA 1 1 1 a...A... a...A... a...A...""".split("\n")


def test_v4_larger_graph_construction(orc, tmp_path):
    h = "AGTTCTGATCTGGGCTATATGCT"
    g = _graph(orc, tmp_path, [("mom", [h]), ("dad", [h]), ("kid", [h])], 5)
    assert g.N == 19
    assert _records(g) == V4          # also pins the sorted order


# ---- V5: TraversalEngineTest.java:98-122
def test_v5_short_contig_reconstruction(orc, tmp_path):
    g = _graph(orc, tmp_path, [("mom", ["AGTTCTGATCTGGGCTATATGCT"]), ("dad", ["AGTTCGAATCTGGGCTATATGCT"]),
                               ("kid", ["AGTTCTGATCTGGGCTATGGCTA"])], 5)
    exp = {"mom": "AGTTCTGATCTGGGCTATATGCT", "dad": "TTCGAATCTGGGCTATATGCT", "kid": "AGTTCTGATCTGGGCTATGGCT"}
    for c in range(3):
        e = orc.Engine(g, [c], stopper="ContigStopper")
        assert e.walk("CTGGG")[0] == exp[g.sample_name(c)]


# ---- V6: TraversalEngineTest.java:125-157
def test_v6_recruitment(orc, tmp_path):
    h = "AGTTCTGATCTGGGCTATATGCT"
    g = _graph(orc, tmp_path, [("mom", [h]), ("dad", [h]), ("kid", ["AGTTCTG", "ATGGCTA"])], 5)
    kid = g.color_for_sample_name("kid")
    rec = [g.color_for_sample_name("mom"), g.color_for_sample_name("dad")]
    e = orc.Engine(g, [kid], op_and=True, recruitment_colors=rec)
    assert e.walk("GTTCT")[0] == h
    e = orc.Engine(g, [kid], op_and=True)
    assert e.walk("GTTCT")[0] == "AGTTCTG"


MCCORTEX_FIG1 = "ACTGATTTCGATGCGATGCGATGCCACGGTGG"
MCCORTEX_READ = "TTTCGATGCGATGCGATGCCACG"


# ---- V7: TraversalEngineTest.java:210-250
def test_v7_cycles_with_and_without_links(orc, tmp_path):
    g = _graph(orc, tmp_path, [("test", [MCCORTEX_FIG1])], 5)
    e = orc.Engine(g, [0])
    assert e.walk("ACTGA")[0] == "ACTGATTTCGATGC"
    lp = str(tmp_path / "l.ctp.gz")
    orc.build_links(g, lp, "test", [MCCORTEX_READ])
    l = orc.Links(lp)
    e = orc.Engine(g, [0], links=[l])
    assert e.walk("ACTGA")[0] == MCCORTEX_FIG1


# ---- V8: T/utils/io/graph/links/CortexLinksTest.java:32-51
def test_v8_links_header_and_counts(orc, tmp_path):
    g = _graph(orc, tmp_path, [("test", [MCCORTEX_FIG1])], 5)
    lp = str(tmp_path / "l.ctp.gz")
    orc.build_links(g, lp, "test", [MCCORTEX_READ])
    l = orc.Links(lp)
    assert (l.version, l.num_colors, l.k) == (4, 1, 5)
    assert (l.num_kmers_in_graph, l.num_kmers_with_links, l.num_links) == (21, 4, 6)
    recs = l.records()
    assert len(recs) == 4 and sum(len(j) for _, j in recs) == 6
    # derived by the survey's scratch restatement (not asserted by the reference)
    got = {k: sorted(j) for k, j in recs}
    assert got == {"ATCGA": [("R", "GGC")], "ATCGC": [("R", "C"), ("R", "GC")], "ATGCC": [("R", "CCA")],
                   "ATGCG": [("R", "A"), ("R", "CA")]}


# ---- V9: TraversalEngineTest.java:253-304
def test_v9_iterate_fwd_rev(orc, tmp_path):
    hap = "AGTTCGAATCTGGGCTATATGCT"
    g = _graph(orc, tmp_path, [("mom", [hap])], 7)
    e = orc.Engine(g, [0])
    s = "AGTTCGA"
    e.seek(s)
    while e.has_next():
        s += e.next()[0][-1]
    assert s == hap
    s = "ATATGCT"
    e.seek(s)
    while e.has_previous():
        s = e.previous()[0][0] + s
    assert s == hap


# ---- V10: TraversalEngineTest.java:307-358
def test_v10_iterate_to_fork(orc, tmp_path):
    g = _graph(orc, tmp_path, [("kid", ["AGTTCGAATCTGGGCTATATGCT", "AGTTCGAATCTGAGCTATATGCT"])], 7)
    e = orc.Engine(g, [0])
    s = "AGTTCGA"
    e.seek(s)
    while e.has_next():
        s += e.next()[0][-1]
    assert s == "AGTTCGAATCTG"
    s = "ATATGCT"
    e.seek(s)
    while e.has_previous():
        s = e.previous()[0][0] + s
    assert s == "GCTATATGCT"


# ---- V11: TraversalEngineTest.java:361-386
def test_v11_forward_and_backward(orc, tmp_path):
    hap, k = "AGTTCGAATCTGAGCTATATGCT", 7
    g = _graph(orc, tmp_path, [("kid", [hap])], k)
    e = orc.Engine(g, [0])
    checked = 0
    for i in range(1, len(hap) - k):
        sk = hap[i:i + k]
        e.seek(sk)
        if e.has_previous() and e.has_next():
            e.next()
            assert e.previous()[0] == sk
            checked += 1
    assert checked > 0


# ---- V12: TraversalEngineTest.java:389-410
def test_v12_dfs_source_to_sink_with_links(orc, tmp_path):
    k, hap = 5, "GTGTGCTAGGTCTATAGTTATAGGCGCGTCTCCGCAAAAATCGT"
    g = _graph(orc, tmp_path, [("mom", [hap])], k)
    lp = str(tmp_path / "l.ctp.gz")
    orc.build_links(g, lp, "mom", [hap])
    e = orc.Engine(g, [0], links=[orc.Links(lp)])
    r = e.dfs(hap[:k], [hap[-k:]])
    assert not r.is_null
    assert r.walk(hap[:k], 0) == hap


# ---- V13 (walk part): T/utils/traversal/TraversalUtilsTest.java:19-47, 56-84
@pytest.mark.parametrize("mom", [["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC"],
                                 ["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"]])
def test_v13_link_guided_walk(orc, tmp_path, mom):
    kid = ["TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"]
    # the reference test uses a java.util.HashMap<String,...>: colour order = HashMap iteration order
    order = orc.java_string_hashmap_order(["mom", "kid"])
    assert order == ["kid", "mom"]
    haps = {"mom": mom, "kid": kid}
    g = _graph(orc, tmp_path, [(s, haps[s]) for s in order], 7)
    lk = str(tmp_path / "kid.ctp.gz")
    orc.build_links(g, lk, "kid", kid)
    e = orc.Engine(g, [g.color_for_sample_name("kid")], links=[orc.Links(lk)], direction=orc.BOTH, stopper="ContigStopper")
    assert e.walk("TGAGATT")[0] == kid[0]


# ---- V13 (gap-filling part): T/utils/traversal/TraversalUtilsTest.java:48-54, 85-97 — toGraph + fillGaps + toWalk over the ORACLE's searches.
# The container logic (corticall_amd/traversal_utils.py: pure Python, no device) is the same the product uses; here every
# DestinationStopper search runs in the CPU oracle, so the five strings the reference asserts are pinned without a GPU.
class _OracleSearches:
    """what fillGaps needs from an engine: dfs_batch(sources, sinks per source) -> Pseudograph / None per source; close()"""

    def __init__(self, orc, og, links, color, direction):
        from corticall_amd.graph import CortexRecord
        self.orc, self.og, self.CortexRecord = orc, og, CortexRecord
        self.e = orc.Engine(og, [color], links=links, direction=direction, stopper="DestinationStopper", max_length=1000)

    def graph_of(self, r):
        from corticall_amd.traversal import CortexVertex
        from corticall_amd.traversal_utils import CortexEdge, Pseudograph
        if r.is_null:
            return None
        g, vs = Pseudograph(), []
        for km, rec, ci, ix in r.vertices():
            cr = None
            if rec >= 0:
                w, cov, ed = self.og.get_record(rec)
                cr = self.CortexRecord(w, cov, ed, self.og.k, rec)
            vs.append(CortexVertex(km, cr, ci, ix))
            g.addVertex(vs[-1])
        for s_, t_, c in r.edges():
            g.addEdge(vs[s_], vs[t_], CortexEdge(vs[s_], vs[t_], c, 1.0))
        return g

    def dfs_batch(self, sources, sinks):
        out = []
        for s_, sk in zip(sources, sinks):
            r = self.e.dfs(s_, sk)
            out.append(self.graph_of(r))
            r.free()
        return out

    def close(self):
        self.e.close()


@pytest.mark.parametrize("mom,expected", [
    (["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC"],
     [("TGGCTAG", "kid", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"), ("TGGCTAG", "mom", "TGGCTAGGTCATTATGATATTAAAATGCTAGCGC")]),
    (["TGGCTAGGTCATTATGATATTAAAATGCTAGCGC", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"],
     [("TGAGATT", "kid", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"), ("TGATATT", "mom", "TGGCTAGGTCATTATGATATTAAAATGCTAGCGC"),
      ("TGAGATT", "mom", "TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC")])])
def test_v13_fill_gaps(orc, tmp_path, mom, expected):
    from corticall_amd import traversal_utils as tu
    kid = ["TGGCTAGGTCATTATGAGATTAAAATGCTAGCGC"]
    order = orc.java_string_hashmap_order(["mom", "kid"])
    haps = {"mom": mom, "kid": kid}
    g = _graph(orc, tmp_path, [(s, haps[s]) for s in order], 7)
    links = {}
    for s in ("mom", "kid"):
        p = str(tmp_path / (s + ".ctp.gz"))
        orc.build_links(g, p, s, haps[s])
        links[s] = orc.Links(p)
    kc, mc = g.color_for_sample_name("kid"), g.color_for_sample_name("mom")
    # e.walk("TGAGATT") = toWalk(dfs(seed)) :108-110
    walker = orc.Engine(g, [kc], links=[links["kid"]], direction=orc.BOTH, stopper="ContigStopper")
    searches = _OracleSearches(orc, g, [], kc, orc.BOTH)
    r = walker.dfs("TGAGATT")
    w = tu.toWalk(searches.graph_of(r), "TGAGATT", kc)
    r.free()
    searches.close()
    assert tu.toContig(w) == kid[0]
    colors = list(range(g.C))
    gapped = tu.toGraph(w, colors)
    factory = lambda c, direction: _OracleSearches(orc, g, [links["mom"], links["kid"]], c, direction)
    filled = tu.fillGaps(gapped, None, None, colors, engine_factory=factory)
    for seed, sample, hap in expected:
        assert tu.toContig(tu.toWalk(filled, seed, g.color_for_sample_name(sample))) == hap, (seed, sample)


# ---- V14: T/utils/sequence/SequenceUtilsTest.java:19-73
def test_v14_sequence_utils(orc):
    for a, b in zip("ACGTN.acgt", "TGCAN.tgca"):
        assert orc.complement_char(a) == b
    assert orc.revcomp("TACTGACTTTTCTCGCTATTCGTATGCATG") == "CATGCATACGAATAGCGAGAAAAGTCAGTA"
    assert orc.revcomp("NACTGACTTTTCTCGCTATTCGTATGCATG") == "CATGCATACGAATAGCGAGAAAAGTCAGTN"
    assert orc.revcomp("NACTGACTTTTCTCGCTATTCGTATGCATg") == "cATGCATACGAATAGCGAGAAAAGTCAGTN"
    rng = random.Random(1)
    for _ in range(2000):
        for k in (21, 31, 41, 51):
            fw = "".join(rng.choice("ACGT") for _ in range(k))
            rc = orc.revcomp(fw)
            assert orc.canonical(fw) == (fw if fw < rc else rc)


# ---- V15: T/utils/kmer/CanonicalKmerTest.java:8-14
def test_v15_hash_collision(orc):
    a, b = "GAACAAAAAAACTTGATAAATGTTTACAAAA", "ACTCTTTTTTAAATGATTATTGCAGATATAT"
    assert orc.canonical(a) == a and orc.canonical(b) == b
    assert orc.jhash_bytes(a) == orc.jhash_bytes(b) and a != b
    # Java: "abc".getBytes() -> Arrays.hashCode == 126145 ; String.hashCode("abc") == 96354
    assert orc.jhash_bytes("abc") == 126145
    assert orc.lib().orc_jhash_string(b"abc") == 96354


# ---- S2: DestinationStopper's junction limit as an exact integer table
def test_destination_limit_table(orc):
    for size in list(range(0, 40000)) + [10 ** 6, 7451332, 7451333, 2 ** 31 - 1]:
        assert orc.destination_junction_limit(size) == 1 + math.ceil(5.0 * math.exp(-0.0001 * size))


# ---- CortexGraphWriter round trip: T/utils/kmer/CortexGraphWriterTest.java:19-78
def test_writer_roundtrip(orc, tmp_path, golden_dir):
    rows = _table(golden_dir)
    # rebuild the fixture's records from haplotypes is not possible (unknown reads); instead
    # round-trip a TempGraphAssembler graph through write -> read -> compare strings
    h = "AGTTCTGATCTGGGCTATATGCT"
    g = _graph(orc, tmp_path, [("a", [h]), ("b", [h[::-1]])], 5)
    recs = _records(g)
    assert recs == sorted(recs)
    assert g.sample_name(0) == "a" and g.sample_name(1) == "b"
    raw = open(g.path, "rb").read()
    assert raw[:6] == b"CORTEX" and raw[g.data_offset - 6:g.data_offset] == b"CORTEX"
    assert len(raw) == g.data_offset + g.N * (8 * g.W + 5 * g.C)
    assert len(rows) == 66


# ---- Q1: N <= 2 never found on a cold cache
def test_q1_tiny_graph(orc, tmp_path):
    g = _graph(orc, tmp_path, [("s", ["ACGTT"])], 4)   # ACGT (palindrome) + CGTT -> AACG
    assert g.N == 2
    g2 = orc.Graph(g.path, use_cache=False)
    assert g2.find(g.record_string(0).split()[0])[0] == -1
    assert g2.find_batch(np.array([list(b"ACGT")], dtype=np.uint8), tuned=True)[0] == -1
