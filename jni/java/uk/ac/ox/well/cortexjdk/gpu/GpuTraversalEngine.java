package uk.ac.ox.well.cortexjdk.gpu;

import org.jgrapht.graph.DirectedWeightedPseudograph;
import uk.ac.ox.well.cortexjdk.utils.exceptions.CortexJDKException;
import uk.ac.ox.well.cortexjdk.utils.io.graph.cortex.CortexRecord;
import uk.ac.ox.well.cortexjdk.utils.kmer.CanonicalKmer;
import uk.ac.ox.well.cortexjdk.utils.kmer.CortexByteKmer;
import uk.ac.ox.well.cortexjdk.utils.stoppingrules.TraversalStoppingRule;
import uk.ac.ox.well.cortexjdk.utils.traversal.CortexEdge;
import uk.ac.ox.well.cortexjdk.utils.traversal.CortexVertex;
import uk.ac.ox.well.cortexjdk.utils.traversal.CortexVertexFactory;
import uk.ac.ox.well.cortexjdk.utils.traversal.TraversalEngineConfiguration;

import java.util.ArrayList;
import java.util.Arrays;
import java.util.Collection;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import java.util.Set;

import static uk.ac.ox.well.cortexjdk.utils.traversal.TraversalEngineConfiguration.GraphCombinationOperator.AND;
import static uk.ac.ox.well.cortexjdk.utils.traversal.TraversalEngineConfiguration.TraversalDirection.BOTH;
import static uk.ac.ox.well.cortexjdk.utils.traversal.TraversalEngineConfiguration.TraversalDirection.FORWARD;

/**
 * The TraversalEngine facade (utils/traversal/TraversalEngine.java:33-339) over libldbg: same configuration object, same public
 * methods, results as the reference's own value objects.  TraversalEngineFactory.make() returns one of these when the configured
 * graph is a GpuCortexGraph (INTEGRATION.md).  The batch forms (walkAll, dfsAll) are what seed loops such as Partition.execute
 * (commands/discover/call/Partition.java:98-219) should call: one device launch for all seeds.
 */
public class GpuTraversalEngine {
    /** J/utils/stoppingrules/*.java in the order of ldbg_stopping_rule (include/ldbg.h) */
    private static final List<String> STOPPERS = Arrays.asList(
            "ContigStopper", "CycleCollapsingContigStopper", "DestinationStopper", "ExplorationStopper", "NovelPartitionStopper",
            "NovelKmerLimitedContigStopper", "NovelContinuationStopper", "BubbleClosingStopper", "BubbleOpeningStopper", "ContaminantStopper",
            "DustStopper", "GapClosingStopper", "NahrStopper", "NovelKmerAggregationStopper", "OrphanStopper", "PairedReadClosingStopper",
            "TipBeginningStopper", "TipEndStopper", "VisualizationStopper");

    private final TraversalEngineConfiguration ec;
    private final GpuCortexGraph graph;
    private final long handle;
    private final int k, w;

    public GpuTraversalEngine(TraversalEngineConfiguration ec) {
        this.ec = ec;
        if (!(ec.getGraph() instanceof GpuCortexGraph)) { throw new CortexJDKException("GpuTraversalEngine needs a GpuCortexGraph"); }
        graph = (GpuCortexGraph) ec.getGraph();
        k = graph.getKmerSize(); w = graph.getKmerBits();
        long[] links = ec.getLinks().stream().filter(l -> l instanceof GpuCortexLinks).mapToLong(l -> ((GpuCortexLinks) l).handle).toArray();
        if (links.length != ec.getLinks().size()) { throw new CortexJDKException("link annotations of a GpuTraversalEngine must be GpuCortexLinks"); }
        long rois = ec.getRois() instanceof GpuCortexGraph ? ((GpuCortexGraph) ec.getRois()).handle : 0;
        Class<? extends TraversalStoppingRule<CortexVertex, CortexEdge>> rule = ec.getStoppingRule();
        int stopper = rule == null ? -1 : STOPPERS.indexOf(rule.getSimpleName());
        handle = create(graph.handle, rois, links, ints(ec.getTraversalColors()), ints(ec.getJoiningColors()), ints(ec.getRecruitmentColors()),
                ints(ec.getSecondaryColors()), ec.getTraversalDirection() == BOTH ? 0 : (ec.getTraversalDirection() == FORWARD ? 1 : 2),
                ec.getGraphCombinationOperator() == AND ? 1 : 0, stopper, ec.getMaxBranchLength(), ec.connectAllNeighbors());
    }

    public final TraversalEngineConfiguration getConfiguration() { return ec; }
    public void close() { destroy(handle); }

    // ---- walk (TraversalEngine.java:108-110 = toWalk(dfs(seed)) with the first traversal colour)
    public List<CortexVertex> walk(CanonicalKmer seed) { return walk(seed.getKmerAsString()); }
    public List<CortexVertex> walk(String seed) {
        long[] offsets = new long[2], walkLen = new long[1];
        walkBatch(handle, seed.getBytes(), 1, offsets, walkLen);
        return vertices(0);
    }

    /** walk + TraversalUtils.toContig for all seeds in ONE device launch: seed -> contig (empty string: empty walk) */
    public Map<String, String> walkAll(Collection<String> seeds) {
        int n = seeds.size();
        byte[] flat = new byte[n * k];
        int i = 0;
        for (String s : seeds) { System.arraycopy(s.getBytes(), 0, flat, (i++) * k, k); }
        long[] offsets = new long[n + 1], walkLen = new long[n];
        byte[] arena = walkBatch(handle, flat, n, offsets, walkLen);
        Map<String, String> out = new LinkedHashMap<>();
        i = 0;
        for (String s : seeds) { out.put(s, new String(arena, (int) offsets[i], (int) (offsets[i + 1] - offsets[i]))); i++; }
        return out;
    }

    /** List<CortexVertex> of walk i of the last walkAll / walk call */
    public List<CortexVertex> vertices(long i) {
        int n = (int) walkVertices(handle, i, 0, null, null, null, null);
        long[] words = new long[Math.max(1, n) * w], rec = new long[Math.max(1, n)];
        int[] copy = new int[Math.max(1, n)], index = new int[Math.max(1, n)];
        if (n > 0) { walkVertices(handle, i, n, words, rec, copy, index); }
        return makeVertices(n, words, rec, copy, index);
    }

    /** Partition.markUsedRois inputs for the last walkAll (the engine needs a ROI graph): ROI record numbers per walk */
    public int[][] roiHits(int n, boolean[] hasNullOut) {
        long[] off = new long[n + 1];
        byte[] hn = new byte[n];
        int[] hits = walkRoiHits(handle, n, off, hn);
        int[][] out = new int[n][];
        for (int i = 0; i < n; i++) { out[i] = Arrays.copyOfRange(hits, (int) off[i], (int) off[i + 1]); if (hasNullOut != null) { hasNullOut[i] = hn[i] != 0; } }
        return out;
    }

    // ---- dfs (TraversalEngine.java:37-106)
    public DirectedWeightedPseudograph<CortexVertex, CortexEdge> dfs(CanonicalKmer source) { return dfs(source.getKmerAsString()); }
    public DirectedWeightedPseudograph<CortexVertex, CortexEdge> dfs(String source, String... sinks) {
        List<String[]> sk = new ArrayList<>();
        sk.add(sinks);
        return dfsAll(java.util.Collections.singletonList(source), sk).get(0);
    }

    /** dfs(Collection sources[, Collection sinks]) (TraversalEngine.java:37-62): every source towards all the sinks in ONE device launch, the
     *  graphs that came back merged in source order (Graphs.addGraph) by the library */
    public DirectedWeightedPseudograph<CortexVertex, CortexEdge> dfs(Collection<String> sources) { return dfs(sources, null); }
    public DirectedWeightedPseudograph<CortexVertex, CortexEdge> dfs(Collection<String> sources, Collection<String> sinks) {
        int n = sources.size(), m = sinks == null ? 0 : sinks.size();
        if (n == 0) { return null; }
        byte[] src = new byte[n * k];
        int i = 0;
        for (String s : sources) { System.arraycopy(s.getBytes(), 0, src, (i++) * k, k); }
        byte[] sk = new byte[Math.max(1, n * m) * k];
        long[] so = new long[n + 1];
        for (i = 0; i < n; i++) {
            int j = 0;
            if (sinks != null) { for (String s : sinks) { System.arraycopy(s.getBytes(), 0, sk, (i * m + (j++)) * k, k); } }
            so[i + 1] = (long) (i + 1) * m;
        }
        long res = dfsBatch(handle, src, n, sk, so);
        long merged = 0;
        try {
            merged = dfsMerge(res, n);
            return graphOf(merged, 0);
        } finally {
            if (merged != 0) { dfsFree(merged); }
            dfsFree(res);
        }
    }

    // ---- neighbourhood (TraversalEngine.java:147-239) and assemble (:112-145)
    public Set<CortexVertex> getNextVertices(CortexByteKmer sk) { return neighbours(java.util.Collections.singletonList(new String(sk.getKmer())), true).get(0); }
    public Set<CortexVertex> getPrevVertices(CortexByteKmer sk) { return neighbours(java.util.Collections.singletonList(new String(sk.getKmer())), false).get(0); }

    /** getNextVertices / getPrevVertices of many k-mers in ONE device launch; every set iterates in the reference's HashSet order */
    public List<Set<CortexVertex>> neighbours(List<String> kmers, boolean forward) {
        int n = kmers.size();
        byte[] flat = new byte[n * k];
        for (int i = 0; i < n; i++) { System.arraycopy(kmers.get(i).getBytes(), 0, flat, i * k, k); }
        long[] off = new long[n + 1], words = new long[Math.max(1, 4 * n) * w], rec = new long[Math.max(1, 4 * n)];
        neighboursBatch(handle, flat, n, forward, off, words, rec);
        int total = (int) off[n];
        List<CortexVertex> all = makeVertices(total, words, rec, new int[Math.max(1, total)], new int[Math.max(1, total)]);
        List<Set<CortexVertex>> out = new ArrayList<>(n);
        for (int i = 0; i < n; i++) { out.add(new java.util.LinkedHashSet<>(all.subList((int) off[i], (int) off[i + 1]))); }
        return out;
    }

    public List<CortexVertex> assemble(String seed) {
        int n = (int) assemble(handle, seed.getBytes(), 0, null, null);
        long[] words = new long[Math.max(1, n) * w], rec = new long[Math.max(1, n)];
        assemble(handle, seed.getBytes(), n, words, rec);
        List<CortexVertex> vs = makeVertices(n, words, rec, new int[Math.max(1, n)], new int[Math.max(1, n)]);
        // the seed vertex carries the string it was given (TraversalEngine.java:115-118); one that is no k-mer over ACGT (an N, lower case:
        // no record, so no neighbour either) has no packed form to come back in
        if (n == 1 && rec[0] < 0 && !seed.matches("[ACGT]*")) {
            vs.set(0, new CortexVertexFactory().bases(seed).record(null).make());
        }
        return vs;
    }

    private DirectedWeightedPseudograph<CortexVertex, CortexEdge> graphOf(long res, long i) {
        long[] sz = dfsSizes(res, i);
        if (sz[0] != 0) { return null; }
        int nv = (int) sz[1], ne = (int) sz[2];
        long[] words = new long[Math.max(1, nv) * w], rec = new long[Math.max(1, nv)];
        int[] copy = new int[Math.max(1, nv)], index = new int[Math.max(1, nv)];
        int[] es = new int[Math.max(1, ne)], ed = new int[Math.max(1, ne)], ecol = new int[Math.max(1, ne)];
        dfsGet(res, i, words, rec, copy, index, es, ed, ecol);
        List<CortexVertex> vs = makeVertices(nv, words, rec, copy, index);
        DirectedWeightedPseudograph<CortexVertex, CortexEdge> g = new DirectedWeightedPseudograph<>(CortexEdge.class);
        for (CortexVertex v : vs) { g.addVertex(v); }
        for (int e = 0; e < ne; e++) { g.addEdge(vs.get(es[e]), vs.get(ed[e]), new CortexEdge(vs.get(es[e]), vs.get(ed[e]), ecol[e], 1.0)); }
        return g;
    }

    /** dfs(source, sinks...) for many sources in ONE device launch; null entries where the reference returns null */
    public List<DirectedWeightedPseudograph<CortexVertex, CortexEdge>> dfsAll(List<String> sources, List<String[]> sinks) {
        int n = sources.size();
        byte[] src = new byte[n * k];
        for (int i = 0; i < n; i++) { System.arraycopy(sources.get(i).getBytes(), 0, src, i * k, k); }
        long[] so = new long[n + 1];
        int total = 0;
        for (int i = 0; i < n; i++) { total += sinks == null ? 0 : sinks.get(i).length; so[i + 1] = total; }
        byte[] sk = new byte[Math.max(1, total) * k];
        int p = 0;
        if (sinks != null) { for (String[] ss : sinks) { for (String s : ss) { System.arraycopy(s.getBytes(), 0, sk, (p++) * k, k); } } }
        long res = dfsBatch(handle, src, n, sk, so);
        List<DirectedWeightedPseudograph<CortexVertex, CortexEdge>> out = new ArrayList<>();
        try {
            for (int i = 0; i < n; i++) {
                long[] sz = dfsSizes(res, i);
                if (sz[0] != 0) { out.add(null); continue; }
                int nv = (int) sz[1], ne = (int) sz[2];
                long[] words = new long[Math.max(1, nv) * w], rec = new long[Math.max(1, nv)];
                int[] copy = new int[Math.max(1, nv)], index = new int[Math.max(1, nv)];
                int[] es = new int[Math.max(1, ne)], ed = new int[Math.max(1, ne)], ecol = new int[Math.max(1, ne)];
                dfsGet(res, i, words, rec, copy, index, es, ed, ecol);
                List<CortexVertex> vs = makeVertices(nv, words, rec, copy, index);
                DirectedWeightedPseudograph<CortexVertex, CortexEdge> g = new DirectedWeightedPseudograph<>(CortexEdge.class);
                for (CortexVertex v : vs) { g.addVertex(v); }
                for (int e = 0; e < ne; e++) { g.addEdge(vs.get(es[e]), vs.get(ed[e]), new CortexEdge(vs.get(es[e]), vs.get(ed[e]), ecol[e], 1.0)); }
                out.add(g);
            }
        } finally {
            dfsFree(res);
        }
        return out;
    }

    // ---- cursor (TraversalEngine.java:241-339)
    public void seek(String sk) { seek(handle, sk.getBytes()); }
    public boolean hasNext() { return hasNext(handle); }
    public boolean hasPrevious() { return hasPrevious(handle); }
    public CortexVertex next() { return step(true); }
    public CortexVertex previous() { return step(false); }

    private CortexVertex step(boolean forward) {
        byte[] km = new byte[k];
        long rec = forward ? next(handle, km) : previous(handle, km);
        return new CortexVertexFactory().bases(new String(km)).record(rec < 0 ? null : graph.getRecord(rec)).make();
    }

    private List<CortexVertex> makeVertices(int n, long[] words, long[] rec, int[] copy, int[] index) {
        List<CortexVertex> out = new ArrayList<>(n);
        for (int v = 0; v < n; v++) {
            long[] bk = new long[w];
            for (int j = 0; j < w; j++) { bk[j] = Long.reverseBytes(words[v * w + (w - 1 - j)]); }
            String sk = new String(CortexRecord.decodeBinaryKmer(bk, k, w));
            out.add(new CortexVertexFactory().bases(sk).record(rec[v] < 0 ? null : graph.getRecord(rec[v])).copyIndex(copy[v]).index(index[v]).make());
        }
        return out;
    }

    private static int[] ints(Collection<Integer> c) { return c == null ? new int[0] : c.stream().mapToInt(Integer::intValue).toArray(); }

    private static native long create(long graph, long rois, long[] links, int[] trav, int[] join, int[] recruit, int[] secondary,
                                      int direction, int op, int stopper, int maxLen, boolean connectAll);
    private static native void destroy(long h);
    private static native byte[] walkBatch(long h, byte[] seeds, int n, long[] offsets, long[] walkLen);
    private static native long walkVertices(long h, long walk, long capacity, long[] words, long[] rec, int[] copyIndex, int[] index);
    private static native int[] walkRoiHits(long h, int n, long[] offsets, byte[] hasNull);
    private static native void seek(long h, byte[] kmer);
    private static native boolean hasNext(long h);
    private static native boolean hasPrevious(long h);
    private static native long next(long h, byte[] kmerOut);
    private static native long previous(long h, byte[] kmerOut);
    private static native long dfsBatch(long h, byte[] sources, int n, byte[] sinks, long[] sinkOffsets);
    private static native long[] dfsSizes(long res, long i);
    private static native void dfsGet(long res, long i, long[] words, long[] rec, int[] copyIndex, int[] index, int[] edgeSrc, int[] edgeDst, int[] edgeColor);
    private static native String dfsWalk(long res, long i, byte[] seed, int color);
    private static native void dfsFree(long res);
    private static native long dfsKmersTraversed(long h);
    private static native long dfsMerge(long res, int n);
    private static native void neighboursBatch(long h, byte[] kmers, int n, boolean forward, long[] offsets, long[] words, long[] rec);
    private static native long assemble(long h, byte[] seed, long capacity, long[] words, long[] rec);

    /** toContig(toWalk(g, seed, colour)) of a dfs result computed on the library side (used by gap-closing callers) */
    public static String dfsContig(long res, long i, String seed, int color) { return dfsWalk(res, i, seed.getBytes(), color); }
    public long kmersTraversedByDfs() { return dfsKmersTraversed(handle); }
}
