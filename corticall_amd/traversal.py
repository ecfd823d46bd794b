"""Host-side mirror of Corticall's traversal API over libldbg.

Mirrors  uk.ac.ox.well.cortexjdk.utils.traversal.TraversalEngineFactory / TraversalEngine /
         TraversalEngineConfiguration / CortexVertex / TraversalUtils.toContig
         uk.ac.ox.well.cortexjdk.utils.io.graph.links.CortexLinks
         uk.ac.ox.well.cortexjdk.utils.stoppingrules.*  (as names; the rules run on the device)
All traversal work happens in HIP kernels behind the C ABI (include/ldbg.h).
"""
import ctypes as C

import numpy as np

from . import _native
from .graph import CortexGraph, CortexRecord, _as_bytes

# J/utils/stoppingrules/*.java, in the order of ldbg_stopper
STOPPING_RULES = [
    "ContigStopper", "CycleCollapsingContigStopper", "DestinationStopper", "ExplorationStopper",
    "NovelPartitionStopper", "NovelKmerLimitedContigStopper", "NovelContinuationStopper",
    "BubbleClosingStopper", "BubbleOpeningStopper", "ContaminantStopper", "DustStopper",
    "GapClosingStopper", "NahrStopper", "NovelKmerAggregationStopper", "OrphanStopper",
    "PairedReadClosingStopper", "TipBeginningStopper", "TipEndStopper", "VisualizationStopper",
]
globals().update({name: name for name in STOPPING_RULES})   # ContigStopper = "ContigStopper", ...

BOTH, FORWARD, REVERSE = 0, 1, 2      # TraversalEngineConfiguration.TraversalDirection
OR, AND = 0, 1                        # TraversalEngineConfiguration.GraphCombinationOperator


class CortexLinks:
    """J/utils/io/graph/links/CortexLinks.java (un-indexed .ctp.gz -> CortexLinksMap)"""

    def __init__(self, path, graph, lib=None):
        self._lib = lib or graph._lib
        self._d = self._lib.dll
        self.path = str(path)
        h = C.c_void_p()
        self._lib.check(self._d.ldbg_links_open(self.path.encode(), graph._h, C.byref(h)))
        self._h = h
        self._graph = graph
        v, nc, k = C.c_int(), C.c_int(), C.c_int()
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._lib.check(self._d.ldbg_links_info(h, C.byref(v), C.byref(nc), C.byref(k), C.byref(a), C.byref(b), C.byref(c)))
        self.version, self.numColors, self.kmerSize = v.value, nc.value, k.value
        self.numKmersInGraph, self.numKmersWithLinks, self.numLinks = a.value, b.value, c.value

    def getFile(self): return self.path
    def size(self): return self.numKmersWithLinks
    def isEmpty(self): return self.numKmersWithLinks == 0
    def getSource(self):
        """ConnectivityAnnotations.getSource(): the LNKIDX header's source of an indexed file, "unknown" otherwise (:29)"""
        buf = C.create_string_buffer(4096)
        self._lib.check(self._d.ldbg_links_source(self._h, buf, 4096))
        return buf.value.decode() or "unknown"

    @staticmethod
    def index(in_path, out_path, source="", lib=None):
        """IndexLinks (J/commands/index/links/IndexLinks.java:62-135): `in_path` (.ctp / .ctp.gz) -> `out_path` (BGZF, conventionally
        .ctp.bgz) + `out_path`.idx; opening `out_path` then reads through the index.  -> number of records"""
        lib = lib or _native.default_lib()
        n = C.c_int64()
        lib.check(lib.dll.ldbg_links_index(str(in_path).encode(), str(out_path).encode(), str(source).encode(), C.byref(n)))
        return n.value

    def getSampleNameForColor(self, c):
        buf = C.create_string_buffer(4096)
        self._lib.check(self._d.ldbg_links_sample_name(self._h, int(c), buf, 4096))
        return buf.value.decode()

    def _get(self, key):
        kb = _as_bytes(key.getKmerAsBytes() if hasattr(key, "getKmerAsBytes") else key)
        found = C.c_int()
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            st = self._d.ldbg_links_get(self._h, kb, C.byref(found), buf, C.c_int64(cap))
            if st == 7:
                cap *= 8
                continue
            self._lib.check(st)
            return bool(found.value), buf.value.decode()

    def containsKey(self, key): return self._get(key)[0]

    def get(self, key):
        """-> (record k-mer, [(isForward, numJunctions, coverages, junctions)]) in the reference's HashSet order, or None"""
        found, text = self._get(key)
        if not found:
            return None
        lines = [l for l in text.split("\n") if l]
        kmer = lines[0].split()[0]
        out = []
        for l in lines[1:]:
            f = l.split()
            out.append((f[0] == "F", int(f[1]), [int(x) for x in f[2].split(",")], f[3]))
        return kmer, out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.check(self._d.ldbg_links_close(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CortexVertex:
    """J/utils/traversal/CortexVertex.java"""

    def __init__(self, kmer, record, copyIndex=0, index=0):
        self._kmer, self._record, self._copyIndex, self._index = kmer, record, copyIndex, index

    def getKmerAsString(self): return self._kmer
    def getCortexRecord(self): return self._record
    def getCanonicalKmer(self): return self._record.getKmerAsString() if self._record is not None else None
    def getCopyIndex(self): return self._copyIndex
    def getIndex(self): return self._index

    def _key(self): return (self._kmer, self._record, self._copyIndex, self._index)
    def __eq__(self, o): return isinstance(o, CortexVertex) and self._key() == o._key()
    def __hash__(self): return hash(self._key())
    def __repr__(self): return "CortexVertex{sk=%s, index=%d, copyIndex=%d}" % (self._kmer, self._index, self._copyIndex)


class _DfsBatch:
    """owner of one ldbg_dfs_result handle; DfsGraph objects keep it alive"""

    def __init__(self, engine, h):
        self.engine, self.h = engine, h

    def graph(self, i):
        e = self.engine
        isnull, nv, ne = C.c_int(), C.c_int64(), C.c_int64()
        e._lib.check(e._d.ldbg_dfs_result_sizes(self.h, C.c_int64(i), C.byref(isnull), C.byref(nv), C.byref(ne)))
        if isnull.value:
            return None
        return DfsGraph(self, i, nv.value, ne.value)

    def __del__(self):
        try:
            if self.h:
                self.engine._d.ldbg_dfs_result_free(self.h)
                self.h = None
        except Exception:
            pass


class DfsGraph:
    """The DirectedWeightedPseudograph<CortexVertex, CortexEdge> dfs() returns (TraversalEngine.java:64-106):
    vertices and edges in insertion order; edge weight is always 1.0."""

    def __init__(self, batch, i, nv, ne):
        self._batch, self._i, self.nv, self.ne = batch, i, nv, ne
        self._raw = None

    def _fetch(self):
        if self._raw is None:
            e = self._batch.engine
            W = e._graph.getKmerBits()
            words = np.zeros((max(1, self.nv), W), dtype=np.uint64)
            rec = np.zeros(max(1, self.nv), dtype=np.int64)
            copy = np.zeros(max(1, self.nv), dtype=np.int32)
            index = np.zeros(max(1, self.nv), dtype=np.int32)
            es, et, ec = (np.zeros(max(1, self.ne), dtype=np.int32) for _ in range(3))
            P = lambda a: a.ctypes.data_as(C.c_void_p)
            e._lib.check(e._d.ldbg_dfs_result_get(self._batch.h, C.c_int64(self._i), P(words), P(rec), P(copy), P(index), P(es), P(et), P(ec)))
            self._raw = (words[:self.nv], rec[:self.nv], copy[:self.nv], index[:self.nv], es[:self.ne], et[:self.ne], ec[:self.ne])
        return self._raw

    def vertex_tuples(self):
        """-> list of (kmer, record index or -1, copyIndex, index) in insertion order"""
        words, rec, copy, index = self._fetch()[:4]
        k = self._batch.engine._graph.getKmerSize()
        return [(CortexRecord(words[j], [0], [0], k).getKmerAsString(), int(rec[j]), int(copy[j]), int(index[j])) for j in range(self.nv)]

    def edge_tuples(self):
        """-> list of (source vertex number, target vertex number, colour) in insertion order"""
        es, et, ec = self._fetch()[4:]
        return [(int(es[j]), int(et[j]), int(ec[j])) for j in range(self.ne)]

    def vertexSet(self):
        g = self._batch.engine._graph
        cache = {}
        out = []
        for kmer, r, ci, ix in self.vertex_tuples():
            if r >= 0 and r not in cache:
                cache[r] = g.getRecord(r)
            out.append(CortexVertex(kmer, cache.get(r), ci, ix))
        return out

    def walk_contig(self, seed, color):
        """TraversalUtils.toContig(TraversalUtils.toWalk(g, seed, color))"""
        e = self._batch.engine
        cap = self.nv + e._graph.getKmerSize() + 8
        buf = C.create_string_buffer(cap)
        ln = C.c_int64()
        e._lib.check(e._d.ldbg_dfs_result_walk(self._batch.h, C.c_int64(self._i), _as_bytes(seed), C.c_int(color), buf, C.c_int64(cap), C.byref(ln)))
        return buf.value.decode()


class TraversalUtils:
    """J/utils/traversal/TraversalUtils.java (the members on the hot path)"""

    @staticmethod
    def toContig(walk):    # :367-381
        s = ""
        for v in walk:
            sk = v.getKmerAsString()
            s = sk if not s else s + sk[-1]
        return s


class TraversalEngineFactory:
    """J/utils/traversal/TraversalEngineFactory.java:12-88 — builder; make() validates like the reference."""

    def __init__(self, lib=None):
        self._lib = lib
        self._trav, self._join, self._recruit, self._secondary = [], set(), set(), set()
        self._op, self._dir, self._connect, self._maxlen = OR, BOTH, False, 75000
        self._stopper = "ContigStopper"
        self._graph = self._rois = None
        self._links = []
        self._strict = True

    def combinationOperator(self, op): self._op = op; return self
    def traversalDirection(self, td): self._dir = td; return self
    def connectAllNeighbors(self, b): self._connect = bool(b); return self
    def maxBranchLength(self, n): self._maxlen = int(n); return self

    @staticmethod
    def _flat(colors):
        out = []
        for c in colors:
            out.extend(c) if isinstance(c, (list, tuple, set)) else out.append(c)
        return [int(c) for c in out]

    def traversalColors(self, *colors):
        if not colors:
            self._trav = []
        for c in self._flat(colors):
            if c not in self._trav:
                self._trav.append(c)
        return self

    def joiningColors(self, *colors):
        self._join = set() if not colors else self._join | set(self._flat(colors)); return self

    def recruitmentColors(self, *colors):
        self._recruit = set() if not colors else self._recruit | set(self._flat(colors)); return self

    def secondaryColors(self, *colors):
        self._secondary = set() if not colors else self._secondary | set(self._flat(colors)); return self

    def stoppingRule(self, rule): self._stopper = rule; return self
    def graph(self, g): self._graph = g; return self
    def rois(self, g): self._rois = g; return self

    def links(self, *links):
        if not links:
            self._links = []
        for l in links:
            if l is None:
                continue
            for x in (l if isinstance(l, (list, tuple, set)) else [l]):
                if x not in self._links:
                    self._links.append(x)
        return self

    def strictJavaFlip(self, b): self._strict = bool(b); return self

    def make(self):       # TraversalEngineFactory.java:54-88: the checks in the reference's order
        if len(self._trav) == 0:
            raise _native.CortexJDKException("Traversal color(s) must be specified.")
        if self._graph is None:
            # (the colour checks dereference getGraph() before the "Must provide graph to traverse." test at the end is reached)
            raise _native.JavaNullPointerException("TraversalEngineFactory.make: no graph (the reference dereferences it while checking the colours)")
        return TraversalEngine(self)

    def make_pool(self, n=2):
        """n engines of this configuration for batches that run side by side (EnginePool)"""
        return EnginePool(self, n)


class EnginePool:
    """N engines of one configuration on one graph, each with its own HIP stream (csrc/walk.cpp: Engine::Engine) and its own host thread:
    batches handed to the pool run side by side on the device.  A walk launch lasts as long as its longest strands while most of its
    wavefronts are done long before; the next batch of another engine takes the compute units they leave (6.47 -> 4.57 ms per
    batch at C3 with two engines).  The reference's unit of work is one TraversalEngine per thread of a caller that walks seeds in batches
    (Partition, Call): this is that, for the device.

        pool = TraversalEngineFactory()...make_pool(2)
        for contigs, walk_lengths in pool.walk_batches(list_of_seed_batches): ...        # results in the order of the batches
    """

    def __init__(self, factory, n=2):
        self.engines = [factory.make() for _ in range(max(1, int(n)))]

    def map(self, fn, items):
        """fn(engine, item) for every item, dealt out to the engines in turn, one host thread per engine (ctypes calls drop the GIL);
        -> results in the order of the items.  An engine is only ever used by its own thread."""
        import threading
        items = list(items)
        out, err = [None] * len(items), []

        def work(i):
            try:
                for j in range(i, len(items), len(self.engines)):
                    out[j] = fn(self.engines[i], items[j])
            except BaseException as ex:     # noqa: BLE001 — raised again in the caller's thread
                err.append(ex)
        if len(self.engines) == 1 or len(items) <= 1:
            for j, it in enumerate(items):
                out[j] = fn(self.engines[0], it)
            return out
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(self.engines))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]
        return out

    def walk_batches(self, seed_batches):
        """[(contigs, walk lengths)] of every batch of seeds (TraversalEngine.walk_batch), batches overlapping on the device"""
        return self.map(lambda e, seeds: e.walk_batch(seeds), seed_batches)

    def dfs_batches(self, batches):
        """batches: (sources, sinks) pairs as TraversalEngine.dfs_batch takes them -> one list of graphs per batch"""
        return self.map(lambda e, b: e.dfs_batch(b[0], b[1]), batches)

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []


class TraversalEngine:
    """J/utils/traversal/TraversalEngine.java"""

    def __init__(self, f):
        g = f._graph
        self._lib = f._lib or g._lib
        self._d = self._lib.dll
        self._graph = g
        cfg = _native.EngineConfig()
        self._d.ldbg_engine_config_default(C.byref(cfg))
        cfg.graph = g._h
        cfg.rois = f._rois._h if f._rois is not None else None
        self._links = list(f._links)
        self._link_arr = (C.c_void_p * max(1, len(self._links)))(*[l._h for l in self._links])
        cfg.links = C.cast(self._link_arr, C.POINTER(C.c_void_p))
        cfg.nlinks = len(self._links)
        for name, vals in (("traversal", f._trav), ("joining", sorted(f._join)), ("recruitment", sorted(f._recruit)),
                           ("secondary", sorted(f._secondary))):
            arr = getattr(cfg, name + "_colors")
            for i, c in enumerate(vals[:_native.MAX_COLORS]):
                arr[i] = c
            setattr(cfg, "n_" + name, len(vals))
        cfg.direction, cfg.combination_operator = f._dir, f._op
        cfg.stopping_rule = -1 if f._stopper is None else (STOPPING_RULES.index(f._stopper) if isinstance(f._stopper, str) else int(f._stopper))
        cfg.max_branch_length = f._maxlen
        cfg.connect_all_neighbors = 1 if f._connect else 0
        cfg.strict_java_flip = 1 if f._strict else 0
        self._cfg = cfg
        self._trav = list(f._trav)
        h = C.c_void_p()
        self._lib.check(self._d.ldbg_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.kmers_traversed = 0

    # ---- batch forms
    def walk_batch(self, seeds):
        """seeds: list of str or np.uint8[n,k] -> (contigs list[str], walk_len i64[n]); device-resident until fetched"""
        arena, offs, wl = self.walk_batch_arrays(seeds)
        raw = arena.tobytes()
        return [raw[offs[i]:offs[i + 1]].decode() for i in range(len(wl))], wl

    def _arena(self, nbytes):
        """page-locked result arena of this engine (ldbg_host_alloc), kept from batch to batch and grown when needed"""
        if getattr(self, "_pin_cap", 0) < nbytes:
            if getattr(self, "_pin_ptr", None):
                self._d.ldbg_host_free(self._pin_ptr)
            cap = int(nbytes * 1.25) + 4096
            p = C.c_void_p()
            self._lib.check(self._d.ldbg_host_alloc(C.c_int64(cap), C.byref(p)))
            self._pin_ptr, self._pin_cap = p, cap
            self._pin_np = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cap,))
        return self._pin_np[:nbytes]

    def walk_batch_arrays(self, seeds, fetch=True, pinned=False):
        """-> (contig arena u8[total], offsets i64[n+1], walk lengths i64[n]).  pinned=True: the arena is this engine's page-locked block
        (the download runs at the bus rate); it is valid until the next batch of this engine.
        seeds: strings, an (n, k) uint8 numpy array, or an (n, k) uint8 tensor ON THIS ENGINE'S DEVICE (anything with .data_ptr(), .is_cuda,
        .shape: the seeds are used where they are, ldbg_engine_walk_batch_run_device; the caller has synchronised the stream that wrote them)."""
        k = self._graph.getKmerSize()
        total, trav = C.c_int64(), C.c_int64()
        if hasattr(seeds, "data_ptr") and getattr(seeds, "is_cuda", False):
            if seeds.dim() != 2 or seeds.shape[1] != k or seeds.element_size() != 1 or not seeds.is_contiguous():
                raise ValueError("device seeds: a contiguous (n, %d) uint8 tensor" % k)
            n = int(seeds.shape[0])
            self._lib.check(self._d.ldbg_engine_walk_batch_run_device(self._h, C.c_void_p(seeds.data_ptr()), C.c_int64(n),
                                                                      C.byref(total), C.byref(trav)))
        else:
            if isinstance(seeds, np.ndarray):
                a = np.ascontiguousarray(seeds, dtype=np.uint8)
            else:
                a = np.frombuffer(b"".join(_as_bytes(s) for s in seeds), dtype=np.uint8).reshape(len(seeds), k)
            n = a.shape[0]
            self._lib.check(self._d.ldbg_engine_walk_batch_run(self._h, a.ctypes.data_as(C.c_char_p), C.c_int64(n),
                                                               C.byref(total), C.byref(trav)))
        self.kmers_traversed = trav.value
        self.last_total_bytes = total.value
        if not fetch:
            return None, None, None
        arena = self._arena(max(1, total.value)) if pinned else np.empty(max(1, total.value), dtype=np.uint8)
        offs = np.zeros(n + 1, dtype=np.int64)
        wl = np.zeros(max(1, n), dtype=np.int64)
        self._lib.check(self._d.ldbg_engine_walk_batch_fetch(self._h, arena.ctypes.data_as(C.c_char_p), C.c_int64(total.value),
                                                             offs.ctypes.data_as(C.c_void_p), wl.ctypes.data_as(C.c_void_p)))
        return arena[:total.value], offs, wl[:n]

    def walk_vertices(self, i):
        """vertices of walk i of the last batch -> list[CortexVertex]"""
        W, k = self._graph.getKmerBits(), self._graph.getKmerSize()
        ln = C.c_int64()
        st = self._d.ldbg_engine_walk_vertices(self._h, C.c_int64(i), C.c_int64(0), C.byref(ln), None, None, None, None)
        if st not in (0, 7):
            self._lib.check(st)
        n = ln.value
        if n == 0:
            return []
        words = np.empty((n, W), dtype=np.uint64)
        rec = np.empty(n, dtype=np.int64)
        copy = np.empty(n, dtype=np.int32)
        index = np.empty(n, dtype=np.int32)
        self._lib.check(self._d.ldbg_engine_walk_vertices(self._h, C.c_int64(i), C.c_int64(n), C.byref(ln),
                                                          words.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p),
                                                          copy.ctypes.data_as(C.c_void_p), index.ctypes.data_as(C.c_void_p)))
        out = []
        cache = {}
        for j in range(n):
            r = int(rec[j])
            if r >= 0 and r not in cache:
                cache[r] = self._graph.getRecord(r)
            kmer = CortexRecord(words[j], [0], [0], k).getKmerAsString()
            out.append(CortexVertex(kmer, cache.get(r), int(copy[j]), int(index[j])))
        return out

    # ---- TraversalEngine members
    def walk(self, seed):        # :108-110
        self.walk_batch_arrays([seed], fetch=False)
        return self.walk_vertices(0)

    def dfs(self, source, *sinks):   # :64-106, and the Collection form :37-62
        """dfs(String source, String... sinks) -> DfsGraph, or None where the reference returns null.
        dfs(Collection<String> sources[, Collection<String> sinks]) — source is a list / tuple / set: every source is searched (all of them
        towards all the sinks, in one device batch) and the graphs that came back are merged in source order with Graphs.addGraph."""
        if not isinstance(source, (str, bytes)):
            sources = list(source)
            all_sinks = list(sinks[0]) if (len(sinks) == 1 and sinks[0] is not None and not isinstance(sinks[0], (str, bytes))) else [x for x in sinks if x is not None]
            n = len(sources)
            if n == 0:
                return None
            batch = self.dfs_batch_arrays(*self._dfs_arrays(sources, [all_sinks] * n))
            which = np.arange(n, dtype=np.int64)
            res = C.c_void_p()
            self._lib.check(self._d.ldbg_dfs_result_merge(batch.h, which.ctypes.data_as(C.c_void_p), C.c_int64(n), C.byref(res)))
            return _DfsBatch(self, res).graph(0)
        if len(sinks) == 1 and not isinstance(sinks[0], (str, bytes)):
            sinks = tuple(sinks[0])
        return self.dfs_batch([source], [list(sinks)])[0]

    def _dfs_arrays(self, sources, sinks):
        n = len(sources)
        src = np.frombuffer(b"".join(_as_bytes(s) for s in sources), dtype=np.uint8)
        sink_buf, off = None, None
        if sinks is not None:
            flat = [_as_bytes(x) for ss in sinks for x in ss]
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(ss) for ss in sinks])
            sink_buf = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8)
        return src, n, sink_buf, off

    # ---- neighbourhood and assemble
    def neighbours_batch(self, kmers, forward=True):
        """getNextVertices / getPrevVertices (:147-239) of every k-mer in one launch -> list (per k-mer) of lists of CortexVertex, each in
        the iteration order of the HashSet the reference returns"""
        k, W = self._graph.getKmerSize(), self._graph.getKmerBits()
        if isinstance(kmers, np.ndarray):
            a = np.ascontiguousarray(kmers, dtype=np.uint8).reshape(-1, k)
        else:
            a = np.frombuffer(b"".join(_as_bytes(s) for s in kmers), dtype=np.uint8).reshape(len(kmers), k)
        n = a.shape[0]
        offs = np.zeros(n + 1, dtype=np.int64)
        words = np.zeros((max(1, 4 * n), W), dtype=np.uint64)
        rec = np.zeros(max(1, 4 * n), dtype=np.int64)
        P = lambda x: x.ctypes.data_as(C.c_void_p)
        self._lib.check(self._d.ldbg_engine_neighbours_batch(self._h, a.ctypes.data_as(C.c_char_p), C.c_int64(n), C.c_int(1 if forward else 0), P(offs), P(words), P(rec),
                                                             C.c_int64(4 * n)))
        out, cache = [], {}
        for i in range(n):
            vs = []
            for j in range(int(offs[i]), int(offs[i + 1])):
                r = int(rec[j])
                if r >= 0 and r not in cache:
                    cache[r] = self._graph.getRecord(r)
                vs.append(CortexVertex(CortexRecord(words[j], [0], [0], k).getKmerAsString(), cache.get(r)))
            out.append(vs)
        return out

    def getNextVertices(self, sk): return self.neighbours_batch([sk], True)[0]     # :195-239
    def getPrevVertices(self, sk): return self.neighbours_batch([sk], False)[0]    # :147-193

    def assemble(self, seed):    # :112-145
        """seek(seed), next() while hasNext(), previous() while hasPrevious() — on the device in two launches -> list[CortexVertex]"""
        k, W = self._graph.getKmerSize(), self._graph.getKmerBits()
        ln = C.c_int64()
        st = self._d.ldbg_engine_assemble(self._h, _as_bytes(seed), C.c_int64(0), C.byref(ln), None, None)
        if st not in (0, 7):
            self._lib.check(st)
        n = ln.value
        words = np.zeros((max(1, n), W), dtype=np.uint64)
        rec = np.zeros(max(1, n), dtype=np.int64)
        self._lib.check(self._d.ldbg_engine_assemble(self._h, _as_bytes(seed), C.c_int64(n), C.byref(ln), words.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p)))
        out, cache = [], {}
        for j in range(ln.value):
            r = int(rec[j])
            if r >= 0 and r not in cache:
                cache[r] = self._graph.getRecord(r)
            out.append(CortexVertex(CortexRecord(words[j], [0], [0], k).getKmerAsString(), cache.get(r)))
        # the seed vertex carries the string it was given (:115-118); one that is no k-mer over ACGT (an N, lower case: no record, so no
        # neighbour either) has no packed form to come back in
        sb = _as_bytes(seed)
        if len(out) == 1 and out[0].getCortexRecord() is None and any(c not in b"ACGT" for c in sb):
            out[0] = CortexVertex(sb.decode(), None)
        return out

    def dfs_batch(self, sources, sinks=None):
        """dfs(source, sinks...) for every source in one device launch; sinks: per source a list of k-mers (or None).
        -> list of DfsGraph / None"""
        n = len(sources)
        batch = self.dfs_batch_arrays(*self._dfs_arrays(sources, sinks))
        return [batch.graph(i) for i in range(n)]

    def dfs_batch_arrays(self, src, n, sink_buf=None, sink_off=None):
        """array form: src u8[n*k], sink_buf u8[*], sink_off i64[n+1] (CSR over sink k-mers) -> result batch handle
        (graph(i) materialises seed i)"""
        res = C.c_void_p()
        t0 = C.c_int64()
        self._lib.check(self._d.ldbg_engine_dfs_kmers_traversed(self._h, C.byref(t0)))
        self._lib.check(self._d.ldbg_engine_dfs_batch(
            self._h, src.ctypes.data_as(C.c_char_p), C.c_int64(n),
            sink_buf.ctypes.data_as(C.c_char_p) if sink_buf is not None else None,
            sink_off.ctypes.data_as(C.c_void_p) if sink_off is not None else None, C.byref(res)))
        batch = _DfsBatch(self, res)
        t = C.c_int64()
        self._lib.check(self._d.ldbg_engine_dfs_kmers_traversed(self._h, C.byref(t)))
        self.dfs_kmers_traversed = t.value - t0.value
        return batch

    def seek(self, sk):          # :321-335
        self._lib.check(self._d.ldbg_engine_seek(self._h, _as_bytes(sk)))

    def hasNext(self):
        y = C.c_int()
        self._lib.check(self._d.ldbg_engine_has_next(self._h, C.byref(y)))
        return bool(y.value)

    def hasPrevious(self):
        y = C.c_int()
        self._lib.check(self._d.ldbg_engine_has_previous(self._h, C.byref(y)))
        return bool(y.value)

    def _step(self, fn):
        k = self._graph.getKmerSize()
        buf = C.create_string_buffer(k + 1)
        rec = C.c_int64()
        self._lib.check(fn(self._h, buf, C.byref(rec)))
        r = self._graph.getRecord(rec.value) if rec.value >= 0 else None
        return CortexVertex(buf.value.decode(), r)

    def next(self): return self._step(self._d.ldbg_engine_next)           # :241-279
    def previous(self): return self._step(self._d.ldbg_engine_previous)   # :281-319

    def close(self):
        if getattr(self, "_pin_ptr", None):
            self._pin_np = None
            self._d.ldbg_host_free(self._pin_ptr)
            self._pin_ptr, self._pin_cap = None, 0
        if getattr(self, "_h", None):
            self._lib.check(self._d.ldbg_engine_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def profile_reset(lib=None):
    (lib or _native.default_lib()).dll.ldbg_profile_reset()


def profile_get(family, lib=None):
    ms, n = C.c_double(), C.c_int64()
    (lib or _native.default_lib()).dll.ldbg_profile_get(family.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value
