"""Host-side mirror of Corticall's graph API over libldbg (no compute here: every lookup and record
fetch is a HIP kernel behind the C ABI).

Mirrors  uk.ac.ox.well.cortexjdk.utils.io.graph.DeBruijnGraph          (J/utils/io/graph/DeBruijnGraph.java:16-53)
         ...utils.io.graph.cortex.CortexGraph / CortexRecord / CortexHeader / CortexColor
Method names follow the Java ones so that reference tests read the same here.
"""
import ctypes as C

import numpy as np

from . import _native

_ALPHA = b"ACGT"


def _as_bytes(k):
    return k.encode() if isinstance(k, str) else bytes(k)


class CortexRecord:
    """J/utils/io/graph/cortex/CortexRecord.java — value object (packed k-mer words, coverages, edge bytes)."""

    def __init__(self, words, coverages, edges, kmer_size, index=-1):
        self.words = tuple(int(w) for w in words)
        self.coverages = tuple(int(np.int32(np.uint32(c))) for c in coverages)   # Java int view of the u32 (Q5)
        self.edges = tuple(int(e) & 0xFF for e in edges)
        self.kmerSize = kmer_size
        self.index = index

    def getKmerSize(self): return self.kmerSize
    def getKmerBits(self): return (self.kmerSize + 31) // 32
    def getNumColors(self): return len(self.coverages)
    def getCoverage(self, c): return self.coverages[c]
    def getCoverages(self): return list(self.coverages)
    def getEdges(self): return list(self.edges)

    def getKmerAsString(self):
        k, W = self.kmerSize, len(self.words)
        out = bytearray(k)
        for i in range(k):
            bit = 2 * (k - 1 - i)
            out[i] = _ALPHA[(self.words[W - 1 - (bit >> 6)] >> (bit & 63)) & 3]
        return out.decode()

    def getKmerAsBytes(self): return self.getKmerAsString().encode()

    def getEdgesAsString(self, c):
        e = self.edges[c]
        left, right = e >> 4, e & 0xF
        s = bytearray(b"........")
        for i in range(4):
            if left & (1 << (3 - i)):
                s[i] = b"acgt"[i]
            if right & (1 << i):
                s[i + 4] = b"ACGT"[i]
        return s.decode()

    def getInEdgesAsStrings(self, c, complement=False):
        alpha = "TGCA" if complement else "ACGT"
        left = self.edges[c] >> 4
        return [alpha[i] for i in range(4) if left & (1 << (3 - i))]

    def getOutEdgesAsStrings(self, c, complement=False):
        alpha = "TGCA" if complement else "ACGT"
        right = self.edges[c] & 0xF
        return [alpha[i] for i in range(4) if right & (1 << i)]

    def getInDegree(self, c): return len(self.getInEdgesAsStrings(c))
    def getOutDegree(self, c): return len(self.getOutEdgesAsStrings(c))

    def toString(self):
        return " ".join([self.getKmerAsString()] + [str(c) for c in self.coverages] +
                        [self.getEdgesAsString(c) for c in range(len(self.edges))])

    __str__ = toString

    def __eq__(self, o):
        return isinstance(o, CortexRecord) and (self.words, self.coverages, self.edges) == (o.words, o.coverages, o.edges)

    def __hash__(self):
        return hash((self.words, self.coverages, self.edges))


class CortexGraph:
    """J/utils/io/graph/cortex/CortexGraph.java — here: a .ctx file resident in MI355X HBM."""

    def __init__(self, path, device=0, lib=None, image=None, device_records=None):
        """image: bytes-like .ctx image to load instead of the file at `path` (ldbg_graph_open_memory)
        device_records: (header bytes, device pointer, number of records) — the records are in device memory already (ldbg_graph_open_device)"""
        self._lib = lib or _native.default_lib()
        self._d = self._lib.dll
        self.path = str(path)
        h = C.c_void_p()
        if device_records is not None:
            hdr, ptr, n = device_records
            hb = np.frombuffer(hdr, dtype=np.uint8)
            self._lib.check(self._d.ldbg_graph_open_device(hb.ctypes.data_as(C.c_void_p), C.c_int64(hb.size), C.c_void_p(ptr), C.c_int64(n), int(device), C.byref(h)))
        elif image is None:
            self._lib.check(self._d.ldbg_graph_open(self.path.encode(), int(device), C.byref(h)))
        else:
            buf = np.frombuffer(image, dtype=np.uint8)
            self._lib.check(self._d.ldbg_graph_open_memory(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size), int(device), C.byref(h)))
        self._h = h
        k, W, Cc, N, v = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        self._lib.check(self._d.ldbg_graph_info(h, C.byref(k), C.byref(W), C.byref(Cc), C.byref(N), C.byref(v)))
        self._k, self._W, self._C, self._N, self._version = k.value, W.value, Cc.value, N.value, v.value
        self._pos = 0

    @classmethod
    def _from_handle(cls, handle, lib, name, owner=None):
        """a view of a graph handle owned by something else (the image of a sharded table): close() leaves the handle alone"""
        g = cls.__new__(cls)
        g._lib, g._d, g.path, g._h, g._owner, g._borrowed = lib, lib.dll, str(name), handle, owner, True
        k, W, Cc, N, v = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        lib.check(lib.dll.ldbg_graph_info(handle, C.byref(k), C.byref(W), C.byref(Cc), C.byref(N), C.byref(v)))
        g._k, g._W, g._C, g._N, g._version = k.value, W.value, Cc.value, N.value, v.value
        g._pos = 0
        return g

    # ---- header getters (CortexGraph.java:323-336)
    def getFile(self): return self.path
    def getVersion(self): return self._version
    def getKmerSize(self): return self._k
    def getKmerBits(self): return self._W
    def getNumColors(self): return self._C
    def getNumRecords(self): return self._N
    def hasColor(self, c): return 0 <= c < self._C

    def getSampleName(self, color):
        buf = C.create_string_buffer(4096)
        self._lib.check(self._d.ldbg_graph_sample_name(self._h, int(color), buf, 4096))
        return buf.value.decode()

    def getColor(self, color):
        info = _native.ColorInfo()
        name = C.create_string_buffer(4096)
        self._lib.check(self._d.ldbg_graph_color_info(self._h, int(color), C.byref(info), name, 4096))
        return {
            "sampleName": self.getSampleName(color), "meanReadLength": info.mean_read_length,
            "totalSequence": info.total_sequence, "tipClippingApplied": bool(info.tip_clipping),
            "lowCovgSupernodesRemoved": bool(info.low_covg_supernodes_removed),
            "lowCovgKmersRemoved": bool(info.low_covg_kmers_removed),
            "cleanedAgainstGraph": bool(info.cleaned_against_graph),
            "lowCovSupernodesThreshold": info.low_cov_supernodes_threshold,
            "lowCovKmerThreshold": info.low_cov_kmer_threshold,
            "cleanedAgainstGraphName": name.value.decode(),
        }

    def getColors(self): return [self.getColor(c) for c in range(self._C)]

    def getColorForSampleName(self, name):
        c = C.c_int()
        self._lib.check(self._d.ldbg_graph_color_for_sample_name(self._h, str(name).encode(), C.byref(c)))
        return c.value

    def getColorsForSampleNames(self, names):
        return [self.getColorForSampleName(n) for n in (names or [])]

    # ---- bulk forms (the batch-first boundary)
    def records(self, first, n):
        """records [first, first+n) -> (words u64[n,W], cov i32[n,C], edges u8[n,C])"""
        n = int(n)
        words = np.empty((n, self._W), dtype=np.uint64)
        cov = np.empty((n, self._C), dtype=np.uint32)
        edges = np.empty((n, self._C), dtype=np.uint8)
        self._lib.check(self._d.ldbg_graph_records(self._h, C.c_int64(first), C.c_int64(n),
                                                   words.ctypes.data_as(C.c_void_p), cov.ctypes.data_as(C.c_void_p),
                                                   edges.ctypes.data_as(C.c_void_p)))
        return words, cov.view(np.int32), edges

    def find_batch(self, kmers, with_payload=True):
        """kmers: list of str/bytes, or np.uint8[n,k] ASCII -> (idx i64[n], cov i32[n,C], edges u8[n,C]); -1 = null"""
        if isinstance(kmers, np.ndarray):
            a = np.ascontiguousarray(kmers, dtype=np.uint8)
        else:
            a = np.frombuffer(b"".join(_as_bytes(x) for x in kmers), dtype=np.uint8).reshape(len(kmers), self._k)
        n = a.shape[0]
        if a.ndim != 2 or a.shape[1] != self._k:
            raise ValueError("k-mers must all have length k=%d" % self._k)
        idx = np.empty(n, dtype=np.int64)
        cov = np.zeros((n, self._C), dtype=np.uint32) if with_payload else None
        edges = np.zeros((n, self._C), dtype=np.uint8) if with_payload else None
        self._lib.check(self._d.ldbg_graph_find_ascii(
            self._h, a.ctypes.data_as(C.c_char_p), C.c_int64(n), idx.ctypes.data_as(C.c_void_p),
            cov.ctypes.data_as(C.c_void_p) if with_payload else None,
            edges.ctypes.data_as(C.c_void_p) if with_payload else None))
        return idx, (cov.view(np.int32) if with_payload else None), edges

    def find_packed(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, self._W)
        idx = np.empty(w.shape[0], dtype=np.int64)
        self._lib.check(self._d.ldbg_graph_find(self._h, w.ctypes.data_as(C.c_void_p), C.c_int64(w.shape[0]),
                                                idx.ctypes.data_as(C.c_void_p), None, None))
        return idx

    # ---- scalar DeBruijnGraph methods = batch of one
    def getRecord(self, i):
        if i < 0:
            raise _native.CortexJDKException("Record index is prefix of range (%d vs 0-%d)" % (i, self._N - 1))
        if i >= self._N:
            return None    # Q2
        w, c, e = self.records(i, 1)
        return CortexRecord(w[0], c[0], e[0], self._k, i)

    def findRecord(self, kmer):
        kb = _as_bytes(kmer.getKmerAsBytes() if hasattr(kmer, "getKmerAsBytes") else kmer)
        idx, cov, edges = self.find_batch([kb])
        if idx[0] < 0:
            return None
        w, _, _ = self.records(int(idx[0]), 1)
        return CortexRecord(w[0], cov[0], edges[0], self._k, int(idx[0]))

    # ---- Iterable<CortexRecord>, Iterator<CortexRecord>
    def position(self, i=None):
        if i is None:
            return self._pos
        if i < 0:
            raise _native.CortexJDKException("Record index is prefix of range (%d vs 0-%d)" % (i, self._N - 1))
        self._pos = i

    def __iter__(self):
        chunk = 1 << 16
        for first in range(0, self._N, chunk):
            n = min(chunk, self._N - first)
            w, c, e = self.records(first, n)
            for j in range(n):
                self._pos = first + j + 1
                yield CortexRecord(w[j], c[j], e[j], self._k, first + j)

    def remove(self):
        raise NotImplementedError("UnsupportedOperationException")

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.check(self._d.ldbg_graph_close(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CortexCollection(CortexGraph):
    """J/utils/io/graph/cortex/CortexCollection.java — several sorted graphs of one k-mer size presented as ONE graph with every
    member's colours side by side (:34-58).  The merge is done once, on the device (ldbg_graph_open_collection: the radix sort of
    Join without the file), so findRecord (:160-188), the iterator (:218-293) and traversal engines over the collection run on a
    resident table like any CortexGraph.

    Quirks kept: getNumRecords() is 0 (:95-98); getFile / position / getRecord are unsupported (:190-193, 205-213, 300-303);
    getColorForSampleName is -1 unless exactly one colour carries the name (:111-116); findRecord asks every member graph, and a
    member of two records or fewer never answers while the iterator still yields its records."""

    def __init__(self, *graphs, device=0, lib=None):
        if len(graphs) == 1 and isinstance(graphs[0], (list, tuple)):
            graphs = tuple(graphs[0])
        self._members = [g for g in graphs if isinstance(g, CortexGraph)]
        paths = [g.getFile() if isinstance(g, CortexGraph) else str(g) for g in graphs]
        self._lib = lib or (self._members[0]._lib if self._members else _native.default_lib())
        self._d = self._lib.dll
        self.path = "<collection>"
        self._paths = paths
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = C.c_void_p()
        self._lib.check(self._d.ldbg_graph_open_collection(arr, len(paths), 1, int(device), C.byref(h)))
        self._h = h
        k, W, Cc, N, v = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        self._lib.check(self._d.ldbg_graph_info(h, C.byref(k), C.byref(W), C.byref(Cc), C.byref(N), C.byref(v)))
        self._k, self._W, self._C, self._N, self._version = k.value, W.value, Cc.value, N.value, 6
        self._pos = 0
        # the iterator's view differs from findRecord's only when a member has two records or fewer
        self._member_info = []
        first = 0
        for p in paths:
            hd = _ctx_header_of(p)
            self._member_info.append((p, first, hd["C"], hd["N"]))
            first += hd["C"]
        self._iter_view = None
        if any(n <= 2 for _, _, _, n in self._member_info):
            hi = C.c_void_p()
            self._lib.check(self._d.ldbg_graph_open_collection(arr, len(paths), 0, int(device), C.byref(hi)))
            self._iter_view = CortexGraph._from_handle(hi, self._lib, "<collection iterator>")
            self._iter_view._borrowed = False

    def getFile(self):
        raise NotImplementedError("UnsupportedOperationException")

    def getNumRecords(self):
        return 0

    def getGraph(self, color):
        for i, (p, first, nc, _) in enumerate(self._member_info):
            if first <= color < first + nc:
                for g in self._members:
                    if g.getFile() == p:
                        return g
                g = CortexGraph(p, lib=self._lib)
                self._members.append(g)
                return g
        raise _native.CortexJDKException("Color doesn't exist in graph.")

    def getColorsForSampleNames(self, names):
        names = set(names or [])
        return [c for c in range(self._C) if self.getSampleName(c) in names]

    def getColorForSampleName(self, name):
        cols = self.getColorsForSampleNames([name])
        return cols[0] if len(cols) == 1 else -1

    def position(self, i=None):
        raise NotImplementedError("UnsupportedOperationException")

    def getRecord(self, i):
        raise NotImplementedError("UnsupportedOperationException")

    def __iter__(self):
        view = self._iter_view or self
        n, chunk = view._N, 1 << 16
        for first in range(0, n, chunk):
            m = min(chunk, n - first)
            w, c, e = CortexGraph.records(view, first, m)
            for j in range(m):
                yield CortexRecord(w[j], c[j], e[j], self._k, first + j)

    def close(self):
        if self._iter_view is not None:
            self._iter_view.close()
            self._iter_view = None
        for g in self._members:
            g.close()
        CortexGraph.close(self)


def _ctx_header_of(path):
    """k, W, C and the record count of a .ctx file, from its header"""
    import os
    from .distributed import ctx_header
    with open(path, "rb") as f:
        raw = f.read(1 << 22)
    h = ctx_header(raw)
    h["N"] = (os.path.getsize(path) - h["data_offset"]) // (8 * h["W"] + 5 * h["C"])
    return h
