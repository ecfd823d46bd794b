"""Partition — the reference's seed loop over the novel k-mers (J/commands/discover/call/Partition.java:57-219, without the
optional k-mer / variant tables and with linkNovels = false, i.e. ContigStopper): every ROI k-mer that no earlier contig
has claimed is walked, the ROI k-mers on the walk are marked with it (a longer walk takes a k-mer over), and the distinct
contigs are written as FASTA.

The walks do not depend on the marking, so all ROI k-mers are walked in ONE device batch; the order-dependent bookkeeping
(`used` in ROI order, longest walk wins, reverse-complement de-duplication, TreeSet output order) is replayed on the host
from the walk lengths and the per-walk ROI hits the device reports (ldbg_engine_walk_roi_hits)."""
import ctypes as C

import numpy as np

from . import _native
from .traversal import AND, BOTH, OR, ContigStopper, TraversalEngineFactory

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _revcomp(s):
    return s.encode().translate(_COMP)[::-1].decode()


def unpack_kmers(words, k):
    """packed words u64[n, W] -> ASCII u8[n, k]"""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(len(words), -1)
    W = w.shape[1]
    out = np.empty((w.shape[0], k), dtype=np.uint8)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(k):
        bit = 2 * (k - 1 - i)
        out[:, i] = alpha[((w[:, W - 1 - (bit >> 6)] >> np.uint64(bit & 63)) & np.uint64(3)).astype(np.int64)]
    return out


class Partition:
    def __init__(self, graph, rois, links=()):
        self.GRAPH, self.ROIS, self.LINKS = graph, rois, list(links)

    def execute(self):
        """-> the FASTA text the reference prints"""
        g, rois = self.GRAPH, self.ROIS
        k = g.getKmerSize()
        color = g.getColorForSampleName(rois.getSampleName(0))        # getTraversalColor :266-268
        f = (TraversalEngineFactory(lib=g._lib).traversalColors(color).traversalDirection(BOTH).combinationOperator(OR)
             .graph(g).rois(rois).stoppingRule(ContigStopper))
        if self.LINKS:
            f.links(*self.LINKS)
        e = f.make()
        n = rois.getNumRecords()
        if n == 0:
            return ""
        words, _, _ = rois.records(0, n)
        seeds = unpack_kmers(words, k)                                  # loadRois: TreeMap order = the ROI graph's order
        arena, offs, wl = e.walk_batch_arrays(seeds)
        idx, _, _ = g.find_batch(seeds, with_payload=False)
        hit_off = np.zeros(n + 1, dtype=np.int64)
        has_null = np.zeros(n, dtype=np.uint8)
        hits = np.zeros(1, dtype=np.uint32)
        st = e._d.ldbg_engine_walk_roi_hits(e._h, hit_off.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), C.c_int64(0),
                                            has_null.ctypes.data_as(C.c_void_p))
        if st not in (0, 7):
            e._lib.check(st)
        hits = np.zeros(max(1, int(hit_off[n])), dtype=np.uint32)
        e._lib.check(e._d.ldbg_engine_walk_roi_hits(e._h, hit_off.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p),
                                                    C.c_int64(len(hits)), has_null.ctypes.data_as(C.c_void_p)))
        raw = arena.tobytes()
        seed_str = seeds.tobytes().decode()

        def contig_of(i):
            return raw[offs[i]:offs[i + 1]].decode() if wl[i] > 0 else seed_str[i * k:(i + 1) * k]

        # the loop :98-184 — order matters: a k-mer claimed by an earlier walk is not walked itself
        used = [None] * n
        wlen = [int(x) if x > 0 else 1 for x in wl]
        for i in range(n):
            if used[i] is not None:
                continue
            if has_null[i] or (wl[i] == 0 and idx[i] < 0):
                # countNovels / markUsedRois call TreeMap.containsKey(v.getCanonicalKmer()) with null for a vertex without a record
                raise _native.JavaNullPointerException("Partition: a vertex without a record reached used.containsKey (seed %s)"
                                                       % seed_str[i * k:(i + 1) * k])
            mine = hits[hit_off[i]:hit_off[i + 1]] if wl[i] > 0 else (i,)
            for r in mine:                                              # markUsedRois :238-257
                r = int(r)
                if used[r] is None or wlen[i] > wlen[used[r]]:
                    used[r] = i
        contigs = set()                                                 # :186-198
        for i in range(n):
            if used[i] is not None:
                fw = contig_of(used[i])
                if fw not in contigs and _revcomp(fw) not in contigs:
                    contigs.add(fw)
        out = []
        for num, part in enumerate(sorted(contigs)):                    # TreeSet<String> order :200-214
            km = np.frombuffer(part.encode(), dtype=np.uint8)
            windows = np.lib.stride_tricks.sliding_window_view(km, k) if len(km) >= k else np.zeros((0, k), dtype=np.uint8)
            if n > 2:
                ridx, _, _ = rois.find_batch(np.ascontiguousarray(windows), with_payload=False)
                num_novels = int((ridx >= 0).sum())
            else:           # findRecord never finds anything in a graph of <= 2 records (Q1); the reference uses a map here
                keys = {seed_str[j * k:(j + 1) * k] for j in range(n)}
                num_novels = sum(1 for wdw in windows if min(wdw.tobytes().decode(), _revcomp(wdw.tobytes().decode())) in keys)
            out.append(">partition%d len=%d numNovels=%d" % (num, len(part) - k + 1, num_novels))
            out.append(part)
        e.close()
        return "\n".join(out) + ("\n" if out else "")


def java_bytes_hash(kmers):
    """java.util.Arrays.hashCode(byte[]) of every row of an ASCII u8[n, k] array (what CanonicalKmer.hashCode returns,
    J/utils/kmer/CanonicalKmer.java:72-74), as u32"""
    a = np.ascontiguousarray(kmers, dtype=np.uint8)
    h = np.ones(a.shape[0], dtype=np.uint32)
    for i in range(a.shape[1]):
        h = h * np.uint32(31) + a[:, i].astype(np.uint32)
    return h


def java_hashmap_order(hashes):
    """iteration order of a java.util.HashMap filled by n successive put() calls of new keys with these hashCodes: by bucket of the
    final table (16 doubling while n > 0.75 capacity; index = (h ^ h >>> 16) & (capacity - 1)), insertion order within a bucket
    (resizes split a bucket without reordering it).  Bins of 8 and more keys are trees in Java 8; their iteration order is not
    emulated."""
    h = np.asarray(hashes, dtype=np.uint32)
    cap = 16
    while len(h) > (cap * 3) // 4:
        cap *= 2
    bucket = (h ^ (h >> np.uint32(16))) & np.uint32(cap - 1)
    return np.argsort(bucket, kind="stable")


class FindTips:
    """FindTips — chains of novel k-mers anchored at one end only (J/commands/prefilter/FindTips.java:30-137): every ROI k-mer that
    no earlier walk has covered is walked (child colour, BOTH, AND, ContigStopper, links if given); the walk is a tip when one of its
    ends is a novel k-mer without neighbours beyond it; the ROI records on tip walks are written as a graph.

    As in Partition the walks do not depend on the bookkeeping, so all ROI k-mers are walked in one device batch and the loop over
    `used.keySet()` — a HashMap, so in Java hash order — is replayed on the host."""

    def __init__(self, graph, rois, parents, links=()):
        self.GRAPH, self.ROI, self.PARENTS, self.LINKS = graph, rois, list(parents), list(links)
        self.numTipChains = 0
        self.tips = []                 # ROI record numbers of the tip k-mers, ascending

    @staticmethod
    def _neighbour_counts(e, kmers, forward):
        """sizes of getNextVertices / getPrevVertices (TraversalEngine.java:147-239) of every k-mer (u8[m, k]): ldbg_engine_neighbours_batch"""
        m = kmers.shape[0]
        offs = np.zeros(m + 1, dtype=np.int64)
        a = np.ascontiguousarray(kmers)
        e._lib.check(e._d.ldbg_engine_neighbours_batch(e._h, a.ctypes.data_as(C.c_char_p), C.c_int64(m), C.c_int(1 if forward else 0),
                                                       offs.ctypes.data_as(C.c_void_p), None, None, C.c_int64(4 * m)))
        return offs[1:] - offs[:-1]

    def execute(self, out=None):
        g, roi = self.GRAPH, self.ROI
        k = g.getKmerSize()
        child = g.getColorForSampleName(roi.getSampleName(0))
        parents = g.getColorsForSampleNames(self.PARENTS)
        n = roi.getNumRecords()
        self.numTipChains, self.tips = 0, []
        if n > 0:
            f = (TraversalEngineFactory(lib=g._lib).traversalDirection(BOTH).combinationOperator(AND).traversalColors(child)
                 .joiningColors(*parents).stoppingRule(ContigStopper).rois(roi).graph(g))
            if self.LINKS:
                f.links(*self.LINKS)
            e = f.make()
            words, _, _ = roi.records(0, n)
            seeds = unpack_kmers(words, k)                                  # rr.getCanonicalKmer(): records hold canonical k-mers
            arena, offs, wl = e.walk_batch_arrays(seeds)
            hit_off = np.zeros(n + 1, dtype=np.int64)
            has_null = np.zeros(n, dtype=np.uint8)
            hits = np.zeros(1, dtype=np.uint32)
            st = e._d.ldbg_engine_walk_roi_hits(e._h, hit_off.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), C.c_int64(0),
                                                has_null.ctypes.data_as(C.c_void_p))
            if st not in (0, 7):
                e._lib.check(st)
            hits = np.zeros(max(1, int(hit_off[n])), dtype=np.uint32)
            e._lib.check(e._d.ldbg_engine_walk_roi_hits(e._h, hit_off.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p),
                                                        C.c_int64(len(hits)), has_null.ctypes.data_as(C.c_void_p)))
            # the two ends of every walk: is the end a novel k-mer, and does the graph go on beyond it (:80-86)
            walked = np.nonzero(wl > 0)[0]
            ends = np.empty((2 * len(walked), k), dtype=np.uint8)
            for j, i in enumerate(walked):
                c = arena[offs[i]:offs[i + 1]]
                ends[2 * j], ends[2 * j + 1] = c[:k], c[len(c) - k:]
            tip_of = np.zeros(n, dtype=bool)
            if len(walked):
                comp = np.zeros(256, dtype=np.uint8)
                for a, b in zip(b"ACGT", b"TGCA"):
                    comp[a] = b
                rc = comp[ends[:, ::-1]]
                lower = rc.view("S%d" % k).ravel() < ends.view("S%d" % k).ravel()
                canon = np.where(lower[:, None], rc, ends)
                if n > 2:
                    ridx, _, _ = roi.find_batch(np.ascontiguousarray(canon), with_payload=False)
                    novel = ridx >= 0
                else:       # (used.containsKey is a HashMap lookup; findRecord never finds anything in a graph of <= 2 records, Q1)
                    keys = {seeds[j].tobytes() for j in range(n)}
                    novel = np.array([canon[j].tobytes() in keys for j in range(len(canon))], dtype=bool)
                gidx, _, _ = g.find_batch(np.ascontiguousarray(ends), with_payload=False)
                novel = novel & (gidx >= 0)                                  # CortexVertex.getCanonicalKmer() is null without a record (:45)
                # e.getPrevVertices(first).size() / e.getNextVertices(last).size() (:85-88): the engine's own neighbourhood (traversal colour,
                # Java-hash orientation of the record, Q6), asked for all ends in one launch each
                n_prev = self._neighbour_counts(e, ends, False)
                n_next = self._neighbour_counts(e, ends, True)
                left = novel[0::2] & (n_prev[0::2] == 0)
                right = novel[1::2] & (n_next[1::2] == 0)
                tip_of[walked] = left | right
            used = np.zeros(n, dtype=bool)
            is_tip = np.zeros(n, dtype=bool)
            for i in java_hashmap_order(java_bytes_hash(seeds)):            # for (CanonicalKmer rr : used.keySet()) :62
                if used[i] or wl[i] == 0:
                    continue
                mine = hits[hit_off[i]:hit_off[i + 1]]
                used[mine] = True
                if tip_of[i]:
                    self.numTipChains += 1
                    is_tip[mine] = True
            self.tips = [int(x) for x in np.nonzero(is_tip)[0]]
            e.close()
        if out is not None:
            idx = np.asarray(self.tips, dtype=np.int64)
            g._lib.check(g._d.ldbg_ctx_write_records(roi.getFile().encode(), idx.ctypes.data_as(C.c_void_p), C.c_int64(len(idx)), str(out).encode()))
        return self.numTipChains, len(self.tips)


class Sort:
    """J/commands/utils/Sort.java:20-49 — the records of a Cortex graph in k-mer order (device radix sort, ldbg_sort_ctx)"""

    def __init__(self, cortex_graph_path, out_path, device=0, lib=None):
        self.path, self.out, self.device = str(cortex_graph_path), str(out_path), device
        self._lib = lib or _native.default_lib()

    def execute(self):
        n = C.c_int64()
        self._lib.check(self._lib.dll.ldbg_sort_ctx(self.path.encode(), self.out.encode(), int(self.device), C.byref(n)))
        return n.value


class Join:
    """J/commands/utils/Join.java:16-60 — several sorted graphs as one, colours side by side (ldbg_join_ctx)"""

    def __init__(self, graph_paths, out_path, device=0, lib=None):
        self.paths, self.out, self.device = [str(p) for p in graph_paths], str(out_path), device
        self._lib = lib or _native.default_lib()

    def execute(self):
        n = C.c_int64()
        arr = (C.c_char_p * len(self.paths))(*[p.encode() for p in self.paths])
        self._lib.check(self._lib.dll.ldbg_join_ctx(arr, len(self.paths), self.out.encode(), int(self.device), C.byref(n)))
        return n.value
