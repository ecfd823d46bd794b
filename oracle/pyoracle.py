"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/liboracle.so (the CPU restatement of Corticall's
LdBG hot path, see oracle/ldbg_oracle.hpp).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package
(corticall_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

STOPPERS = [
    "ContigStopper", "CycleCollapsingContigStopper", "DestinationStopper", "ExplorationStopper",
    "NovelPartitionStopper", "NovelKmerLimitedContigStopper", "NovelContinuationStopper",
    "BubbleClosingStopper", "BubbleOpeningStopper", "ContaminantStopper", "DustStopper",
    "GapClosingStopper", "NahrStopper", "NovelKmerAggregationStopper", "OrphanStopper",
    "PairedReadClosingStopper", "TipBeginningStopper", "TipEndStopper", "VisualizationStopper",
]
BOTH, FORWARD, REVERSE = 0, 1, 2


class OracleError(RuntimeError):
    pass


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_last_error.restype = C.c_char_p
        L.orc_graph_open.restype = C.c_void_p
        L.orc_links_open.restype = C.c_void_p
        L.orc_engine_create.restype = C.c_void_p
        L.orc_engine_dfs.restype = C.c_void_p
        L.orc_engine_kmers_traversed.restype = C.c_uint64
        L.orc_result_num_vertices.restype = C.c_int64
        L.orc_result_num_edges.restype = C.c_int64
        L.orc_links_dump.restype = C.c_int64
        L.orc_jhash_bytes.restype = C.c_int32
        L.orc_jhash_string.restype = C.c_int32
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise OracleError(lib().orc_last_error().decode())
    return rc


def _b(s):
    return s.encode() if isinstance(s, str) else s


def canonical(s):
    out = C.create_string_buffer(len(s) + 1)
    lib().orc_canonical(_b(s), out)
    return out.value.decode()


def revcomp(s):
    out = C.create_string_buffer(len(s) + 1)
    lib().orc_revcomp(_b(s), out)
    return out.value.decode()


def complement_char(c):
    return chr(lib().orc_complement_char(ord(c)))


def jhash_bytes(s):
    return lib().orc_jhash_bytes(_b(s))


def is_flipped(s):
    return bool(lib().orc_is_flipped(_b(s)))


def encode_kmer(s):
    w = (C.c_uint64 * 8)()
    n = _check(lib().orc_encode_kmer(_b(s), w))
    return [int(w[i]) for i in range(n)]


def decode_kmer(words, k):
    w = (C.c_uint64 * len(words))(*words)
    out = C.create_string_buffer(k + 1)
    _check(lib().orc_decode_kmer(w, k, out))
    return out.value.decode()


def destination_junction_limit(size):
    return lib().orc_destination_junction_limit(size)


def java_string_hashmap_order(keys):
    arr = (C.c_char_p * len(keys))(*[_b(k) for k in keys])
    out = (C.c_int * len(keys))()
    lib().orc_java_string_hashmap_order(len(keys), arr, out)
    return [keys[out[i]] for i in range(len(keys))]


class Graph:
    def __init__(self, path, use_cache=True, tuned=False):
        self.path = path
        self.h = lib().orc_graph_open(_b(path), 1 if use_cache else 0)
        if not self.h:
            raise OracleError(lib().orc_last_error().decode())
        k, W, Cc, N, off = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), C.c_int64()
        lib().orc_graph_info(C.c_void_p(self.h), C.byref(k), C.byref(W), C.byref(Cc), C.byref(N), C.byref(off))
        self.k, self.W, self.C, self.N, self.data_offset = k.value, W.value, Cc.value, N.value, off.value
        if tuned:
            self.set_tuned(True)

    def set_tuned(self, t):
        lib().orc_graph_set_tuned(C.c_void_p(self.h), 1 if t else 0)

    def close(self):
        if self.h:
            lib().orc_graph_close(C.c_void_p(self.h))
            self.h = None

    def sample_name(self, c):
        buf = C.create_string_buffer(4096)
        if lib().orc_graph_sample_name(C.c_void_p(self.h), c, buf, 4096) != 0:
            raise OracleError("bad colour")
        return buf.value.decode()

    def color_for_sample_name(self, name):
        return lib().orc_graph_color_for_sample_name(C.c_void_p(self.h), _b(name))

    def get_record(self, i):
        """-> (words, cov, edges) or None"""
        w = (C.c_uint64 * self.W)()
        cov = (C.c_int32 * self.C)()
        ed = (C.c_uint8 * self.C)()
        rc = _check(lib().orc_graph_get_record(C.c_void_p(self.h), C.c_int64(i), w, cov, ed))
        if rc == 0:
            return None
        return list(w), list(cov), list(ed)

    def record_string(self, i):
        buf = C.create_string_buffer(4096)
        rc = _check(lib().orc_graph_record_string(C.c_void_p(self.h), C.c_int64(i), buf, 4096))
        return buf.value.decode() if rc else None

    def find(self, kmer):
        """-> (idx, cov, edges); idx == -1 when the reference returns null"""
        idx = C.c_int64(-1)
        cov = (C.c_int32 * self.C)()
        ed = (C.c_uint8 * self.C)()
        _check(lib().orc_graph_find(C.c_void_p(self.h), _b(kmer), C.byref(idx), cov, ed))
        return idx.value, list(cov), list(ed)

    def find_batch(self, kmers_ascii, tuned=False):
        """kmers_ascii: np.uint8 [n, k] -> np.int64 [n]"""
        a = np.ascontiguousarray(kmers_ascii, dtype=np.uint8)
        n = a.shape[0]
        out = np.empty(n, dtype=np.int64)
        _check(lib().orc_graph_find_batch(C.c_void_p(self.h), a.ctypes.data_as(C.c_char_p), C.c_int64(n),
                                          out.ctypes.data_as(C.POINTER(C.c_int64)), 1 if tuned else 0))
        return out


def build_graph(out_path, haplotypes, k):
    """haplotypes: list of (sample, [hap, ...]) in map-iteration order (TempGraphAssembler)."""
    names = (C.c_char_p * len(haplotypes))(*[_b(n) for n, _ in haplotypes])
    nh = (C.c_int * len(haplotypes))(*[len(h) for _, h in haplotypes])
    flat = [_b(x) for _, h in haplotypes for x in h]
    fl = (C.c_char_p * len(flat))(*flat)
    _check(lib().orc_build_graph(_b(out_path), k, len(haplotypes), names, nh, fl))
    return out_path


def build_links(graph, out_path, sample, reads):
    arr = (C.c_char_p * len(reads))(*[_b(r) for r in reads])
    _check(lib().orc_build_links(C.c_void_p(graph.h), _b(out_path), _b(sample), len(reads), arr))
    return out_path


class Links:
    def __init__(self, path):
        self.path = path
        self.h = lib().orc_links_open(_b(path))
        if not self.h:
            raise OracleError(lib().orc_last_error().decode())
        v, nc, k = C.c_int(), C.c_int(), C.c_int()
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        lib().orc_links_info(C.c_void_p(self.h), C.byref(v), C.byref(nc), C.byref(k), C.byref(a), C.byref(b), C.byref(c))
        self.version, self.num_colors, self.k = v.value, nc.value, k.value
        self.num_kmers_in_graph, self.num_kmers_with_links, self.num_links = a.value, b.value, c.value

    def dump(self):
        n = 1 << 20
        while True:
            buf = C.create_string_buffer(n)
            rc = lib().orc_links_dump(C.c_void_p(self.h), buf, C.c_int64(n))
            if rc >= 0:
                return buf.value.decode()
            n = -rc + 16

    def records(self):
        """-> list of (kmer, [(orient, junctions)])"""
        out = []
        lines = [l for l in self.dump().split("\n") if l]
        i = 0
        while i < len(lines):
            kmer, n = lines[i].split()
            js = []
            for j in range(int(n)):
                f = lines[i + 1 + j].split()
                js.append((f[0], f[-1]))
            out.append((kmer, js))
            i += 1 + int(n)
        return out

    def close(self):
        if self.h:
            lib().orc_links_close(C.c_void_p(self.h))
            self.h = None


def _iarr(xs):
    xs = list(xs)
    return (C.c_int * max(1, len(xs)))(*xs), len(xs)


class DfsResult:
    def __init__(self, engine, h):
        self.engine, self.h = engine, h
        self.is_null = bool(lib().orc_result_is_null(C.c_void_p(h)))
        self.nv = lib().orc_result_num_vertices(C.c_void_p(h))
        self.ne = lib().orc_result_num_edges(C.c_void_p(h))

    def vertices(self):
        """-> list of (kmer, rec, copy_index, index) in insertion order"""
        k = self.engine.graph.k
        km = np.zeros((max(self.nv, 1), k), dtype=np.uint8)
        rec = np.zeros(max(self.nv, 1), dtype=np.int64)
        ci = np.zeros(max(self.nv, 1), dtype=np.int32)
        ix = np.zeros(max(self.nv, 1), dtype=np.int32)
        lib().orc_result_vertices(C.c_void_p(self.h), km.ctypes.data_as(C.c_char_p), rec.ctypes.data_as(C.POINTER(C.c_int64)),
                                  ci.ctypes.data_as(C.POINTER(C.c_int32)), ix.ctypes.data_as(C.POINTER(C.c_int32)))
        return [(km[i].tobytes().decode(), int(rec[i]), int(ci[i]), int(ix[i])) for i in range(self.nv)]

    def edges(self):
        s = np.zeros(max(self.ne, 1), dtype=np.int32)
        t = np.zeros(max(self.ne, 1), dtype=np.int32)
        c = np.zeros(max(self.ne, 1), dtype=np.int32)
        P = C.POINTER(C.c_int32)
        lib().orc_result_edges(C.c_void_p(self.h), s.ctypes.data_as(P), t.ctypes.data_as(P), c.ctypes.data_as(P))
        return [(int(s[i]), int(t[i]), int(c[i])) for i in range(self.ne)]

    def canonical_sets(self):
        """order-free form: (sorted vertex tuples, sorted edge tuples over vertex tuples)"""
        vs = self.vertices()
        es = sorted((vs[s], vs[t], c) for s, t, c in self.edges())
        return sorted(vs), es

    def walk(self, seed, color):
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        ln = C.c_int64()
        _check(lib().orc_result_walk(C.c_void_p(self.engine.h), C.c_void_p(self.h), _b(seed), color, buf, C.c_int64(cap), C.byref(ln)))
        return buf.value.decode()

    def free(self):
        if self.h:
            lib().orc_result_free(C.c_void_p(self.h))
            self.h = None


class Engine:
    """Mirror of TraversalEngineFactory()...make() (J/utils/traversal/TraversalEngineFactory.java)."""

    def __init__(self, graph, traversal_colors, links=(), rois=None, joining_colors=(), recruitment_colors=(),
                 secondary_colors=(), op_and=False, direction=BOTH, connect_all_neighbors=False,
                 max_length=75000, stopper="ContigStopper"):
        self.graph = graph
        self._links = list(links)
        la = (C.c_void_p * max(1, len(self._links)))(*[l.h for l in self._links])
        tr, ntr = _iarr(traversal_colors)
        jo, njo = _iarr(joining_colors)
        re_, nre = _iarr(recruitment_colors)
        se, nse = _iarr(secondary_colors)
        sid = STOPPERS.index(stopper) if isinstance(stopper, str) else int(stopper)
        self.h = lib().orc_engine_create(C.c_void_p(graph.h), C.c_void_p(rois.h) if rois else None, la, len(self._links),
                                         tr, ntr, jo, njo, re_, nre, se, nse, 1 if op_and else 0, direction,
                                         1 if connect_all_neighbors else 0, max_length, sid)
        if not self.h:
            raise OracleError(lib().orc_last_error().decode())

    def seek(self, kmer):
        _check(lib().orc_engine_seek(C.c_void_p(self.h), _b(kmer)))

    def has_next(self):
        return bool(lib().orc_engine_has_next(C.c_void_p(self.h)))

    def has_previous(self):
        return bool(lib().orc_engine_has_previous(C.c_void_p(self.h)))

    def _step(self, fn):
        buf = C.create_string_buffer(self.graph.k + 1)
        rec = C.c_int64()
        _check(fn(C.c_void_p(self.h), buf, C.byref(rec)))
        return buf.value.decode(), rec.value

    def next(self):
        return self._step(lib().orc_engine_next)

    def previous(self):
        return self._step(lib().orc_engine_previous)

    def kmers_traversed(self):
        return int(lib().orc_engine_kmers_traversed(C.c_void_p(self.h)))

    def walk(self, seed):
        """TraversalUtils.toContig(e.walk(seed)) -> (contig, n_vertices)"""
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            ln, nv = C.c_int64(), C.c_int64()
            rc = _check(lib().orc_engine_walk(C.c_void_p(self.h), _b(seed), buf, C.c_int64(cap), C.byref(ln), C.byref(nv)))
            if rc == 0:
                return buf.value.decode(), nv.value
            cap = ln.value + 16

    def walk_batch(self, seeds_ascii, arena_cap=None):
        a = np.ascontiguousarray(seeds_ascii, dtype=np.uint8)
        n = a.shape[0]
        cap = arena_cap or (1 << 24)
        while True:
            arena = np.empty(cap, dtype=np.uint8)
            offs = np.zeros(n + 1, dtype=np.int64)
            nv = np.zeros(max(n, 1), dtype=np.int64)
            rc = _check(lib().orc_engine_walk_batch(C.c_void_p(self.h), a.ctypes.data_as(C.c_char_p), C.c_int64(n),
                                                    arena.ctypes.data_as(C.c_char_p), C.c_int64(cap),
                                                    offs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                    nv.ctypes.data_as(C.POINTER(C.c_int64))))
            if rc == 0:
                return arena[:offs[n]], offs, nv[:n]
            cap *= 4

    def dfs(self, source, sinks=()):
        sk = [_b(s) for s in sinks]
        arr = (C.c_char_p * max(1, len(sk)))(*sk)
        st = C.c_int()
        h = lib().orc_engine_dfs(C.c_void_p(self.h), _b(source), arr, len(sk), C.byref(st))
        _check(st.value)
        return DfsResult(self, h)

    def dfs_collection(self, sources, sinks=()):
        """dfs(Collection<String> sources, Collection<String> sinks) (TraversalEngine.java:37-62)"""
        so = [_b(s) for s in sources]
        sk = [_b(s) for s in sinks]
        a1 = (C.c_char_p * max(1, len(so)))(*so)
        a2 = (C.c_char_p * max(1, len(sk)))(*sk)
        st = C.c_int()
        lib().orc_engine_dfs_collection.restype = C.c_void_p
        h = lib().orc_engine_dfs_collection(C.c_void_p(self.h), a1, len(so), a2, len(sk), C.byref(st))
        _check(st.value)
        return DfsResult(self, h)

    def _adjacent(self, kmer, forward):
        k = self.graph.k
        buf = np.zeros((8, k), dtype=np.uint8)
        rec = np.zeros(8, dtype=np.int64)
        n = C.c_int()
        _check(lib().orc_engine_adjacent(C.c_void_p(self.h), _b(kmer), 1 if forward else 0, buf.ctypes.data_as(C.c_char_p),
                                         rec.ctypes.data_as(C.POINTER(C.c_int64)), 8, C.byref(n)))
        return [(buf[i].tobytes().decode(), int(rec[i])) for i in range(n.value)]

    def next_vertices(self, kmer):
        """getNextVertices (TraversalEngine.java:195-239) -> [(kmer, record index or -1)] in HashSet iteration order"""
        return self._adjacent(kmer, True)

    def prev_vertices(self, kmer):
        return self._adjacent(kmer, False)

    def assemble(self, seed, max_length):
        """assemble(seed) (TraversalEngine.java:112-145) over the cursor -> [(kmer, record index or -1)] in contig order"""
        rec, _, _ = self.graph.find(seed)
        fw, rv = [], []
        self.seek(seed)
        while self.has_next() and len(fw) < max_length:
            fw.append(self.next())
        self.seek(seed)
        while self.has_previous() and len(rv) < max_length:
            rv.insert(0, self.previous())
        return rv + [(seed, rec)] + fw

    def close(self):
        if self.h:
            lib().orc_engine_destroy(C.c_void_p(self.h))
            self.h = None
