package uk.ac.ox.well.cortexjdk.gpu;

import uk.ac.ox.well.cortexjdk.utils.exceptions.CortexJDKException;
import uk.ac.ox.well.cortexjdk.utils.io.graph.DeBruijnGraph;
import uk.ac.ox.well.cortexjdk.utils.io.graph.cortex.CortexColor;
import uk.ac.ox.well.cortexjdk.utils.io.graph.cortex.CortexHeader;
import uk.ac.ox.well.cortexjdk.utils.io.graph.cortex.CortexRecord;
import uk.ac.ox.well.cortexjdk.utils.kmer.CanonicalKmer;
import uk.ac.ox.well.cortexjdk.utils.kmer.CortexByteKmer;

import java.io.File;
import java.util.ArrayList;
import java.util.Collection;
import java.util.Iterator;
import java.util.List;
import java.util.NoSuchElementException;

/**
 * A Cortex graph resident in MI355X HBM behind Corticall's own graph interface
 * (utils/io/graph/DeBruijnGraph.java:16-53): drop-in for CortexGraph on the traversal path.
 * Every lookup / record fetch is a HIP kernel behind libldbg's C ABI (include/ldbg.h); scalar methods
 * are batches of one, the batch forms (findRecords, getRecords) are what loops should use.
 */
public class GpuCortexGraph implements DeBruijnGraph {
    static { System.loadLibrary("ldbg_jni"); }

    private static final int CHUNK = 1 << 16;

    private final File file;
    final long handle;
    private final int kmerSize, kmerBits, numColors, version;
    private final long numRecords;
    private CortexHeader header;

    private long position = 0;
    private long chunkFirst = -1;
    private CortexRecord[] chunk = new CortexRecord[0];

    public GpuCortexGraph(String path) { this(new File(path), 0); }
    public GpuCortexGraph(File f) { this(f, 0); }
    public GpuCortexGraph(File f, int device) {
        file = f;
        handle = open(f.getAbsolutePath(), device);
        long[] i = info(handle);
        kmerSize = (int) i[0]; kmerBits = (int) i[1]; numColors = (int) i[2]; numRecords = i[3]; version = (int) i[4];
    }

    /**
     * Several graphs as one (CortexCollection.java:34-58): the members merged into one resident table, every member's colours side by
     * side.  findView: the table CortexCollection.findRecord answers from (a member of two records or fewer never answers, :160-188);
     * otherwise the records its iterator yields (:218-293).  getFile() is null for such a graph.
     */
    public static GpuCortexGraph collection(java.util.List<File> members, boolean findView, int device) {
        String[] paths = new String[members.size()];
        for (int i = 0; i < paths.length; i++) { paths[i] = members.get(i).getAbsolutePath(); }
        return new GpuCortexGraph(openCollection(paths, findView, device));
    }
    private GpuCortexGraph(long h) {
        file = null;
        handle = h;
        long[] i = info(handle);
        kmerSize = (int) i[0]; kmerBits = (int) i[1]; numColors = (int) i[2]; numRecords = i[3]; version = (int) i[4];
    }

    // ---- seeking / iterating (CortexGraph.java:183-258)
    @Override public long position() { return position; }
    @Override public void position(long i) {
        if (i < 0) { throw new CortexJDKException("Record index is prefix of range (" + i + " vs 0-" + (numRecords - 1) + ")"); }
        position = i;
    }
    @Override public Iterator<CortexRecord> iterator() { position = 0; return this; }
    @Override public boolean hasNext() { return position < numRecords; }
    @Override public CortexRecord next() {
        if (!hasNext()) { throw new NoSuchElementException(); }
        return getRecord(position++);
    }
    @Override public void remove() { throw new UnsupportedOperationException(); }
    @Override public void close() { close(handle); }

    // ---- records (getRecord: CortexGraph.java:232-258; null beyond the last record, quirk Q2)
    @Override public CortexRecord getRecord(long i) {
        if (i < 0) { throw new CortexJDKException("Record index is prefix of range (" + i + " vs 0-" + (numRecords - 1) + ")"); }
        if (i >= numRecords) { return null; }
        if (i < chunkFirst || i >= chunkFirst + chunk.length) {
            int n = (int) Math.min(CHUNK, numRecords - i);
            chunk = getRecords(i, n);
            chunkFirst = i;
        }
        return chunk[(int) (i - chunkFirst)];
    }

    /** records [first, first + n) in one device call */
    public CortexRecord[] getRecords(long first, int n) {
        long[] words = new long[n * kmerBits];
        int[] cov = new int[n * numColors];
        byte[] edges = new byte[n * numColors];
        records(handle, first, n, words, cov, edges);
        CortexRecord[] out = new CortexRecord[n];
        for (int r = 0; r < n; r++) { out[r] = makeRecord(words, cov, edges, r); }
        return out;
    }

    private CortexRecord makeRecord(long[] words, int[] cov, byte[] edges, int r) {
        long[] bk = new long[kmerBits];
        // the library's words are most-significant first, host order; CortexRecord keeps the file's little-endian longs as
        // Java (big-endian) longs, least significant word first: CortexRecord.java:291-334
        for (int w = 0; w < kmerBits; w++) { bk[w] = Long.reverseBytes(words[r * kmerBits + (kmerBits - 1 - w)]); }
        int[] c = new int[numColors];
        byte[] e = new byte[numColors];
        System.arraycopy(cov, r * numColors, c, 0, numColors);
        System.arraycopy(edges, r * numColors, e, 0, numColors);
        return new CortexRecord(bk, c, e, kmerSize, kmerBits);
    }

    // ---- findRecord (CortexGraph.java:272-321): canonicalised on the device; null for an absent or non-ACGT k-mer
    @Override public CortexRecord findRecord(byte[] bk) { return findRecords(new byte[][] { bk })[0]; }
    @Override public CortexRecord findRecord(CortexByteKmer bk) { return findRecord(bk.getKmer()); }
    @Override public CortexRecord findRecord(CanonicalKmer ck) { return findRecord(ck.getKmerAsBytes()); }
    @Override public CortexRecord findRecord(String sk) { return findRecord(sk.getBytes()); }

    /** findRecord for many k-mers in one device call (null entries where the reference returns null) */
    public CortexRecord[] findRecords(byte[][] kmers) {
        int n = kmers.length;
        byte[] flat = new byte[n * kmerSize];
        for (int i = 0; i < n; i++) {
            if (kmers[i].length != kmerSize) { throw new CortexJDKException("k-mer of length " + kmers[i].length + " given to a graph with k=" + kmerSize); }
            System.arraycopy(kmers[i], 0, flat, i * kmerSize, kmerSize);
        }
        int[] cov = new int[n * numColors];
        byte[] edges = new byte[n * numColors];
        long[] idx = findRecords(handle, flat, n, cov, edges);
        CortexRecord[] out = new CortexRecord[n];
        for (int i = 0; i < n; i++) { out[i] = idx[i] < 0 ? null : getRecord(idx[i]); }
        return out;
    }

    // ---- graph information (CortexGraph.java:323-357)
    @Override public File getFile() { return file; }
    @Override public CortexHeader getHeader() {
        if (header == null) {
            CortexHeader h = new CortexHeader();
            h.setVersion(version); h.setKmerSize(kmerSize); h.setKmerBits(kmerBits); h.setNumColors(numColors);
            for (int c = 0; c < numColors; c++) { h.addColor(getColor(c)); }
            header = h;
        }
        return header;
    }
    @Override public int getVersion() { return version; }
    @Override public int getKmerSize() { return kmerSize; }
    @Override public int getKmerBits() { return kmerBits; }
    @Override public int getNumColors() { return numColors; }
    @Override public long getNumRecords() { return numRecords; }

    @Override public List<CortexColor> getColors() {
        List<CortexColor> out = new ArrayList<>();
        for (int c = 0; c < numColors; c++) { out.add(getColor(c)); }
        return out;
    }
    @Override public boolean hasColor(int color) { return color >= 0 && color < numColors; }
    @Override public CortexColor getColor(int color) {
        String[] cleanedAgainst = new String[1];
        long[] i = colorInfo(handle, color, cleanedAgainst);
        CortexColor cc = new CortexColor();
        cc.setSampleName(sampleName(handle, color));
        cc.setMeanReadLength((int) i[0]); cc.setTotalSequence(i[1]);
        cc.setTipClippingApplied(i[2] != 0); cc.setLowCovgSupernodesRemoved(i[3] != 0); cc.setLowCovgKmersRemoved(i[4] != 0);
        cc.setCleanedAgainstGraph(i[5] != 0);
        cc.setLowCovSupernodesThreshold((int) i[6]); cc.setLowCovKmerThreshold((int) i[7]);
        cc.setCleanedAgainstGraphName(cleanedAgainst[0]);
        return cc;
    }
    @Override public int getColorForSampleName(String sampleName) { return colorForSampleName(handle, sampleName); }
    @Override public List<Integer> getColorsForSampleNames(Collection<String> sampleNames) {
        List<Integer> out = new ArrayList<>();
        if (sampleNames != null) { for (String s : sampleNames) { out.add(getColorForSampleName(s)); } }
        return out;
    }
    @Override public String getSampleName(int color) { return sampleName(handle, color); }

    @Override public String toString() { return "GpuCortexGraph{" + file + ", k=" + kmerSize + ", colors=" + numColors + ", records=" + numRecords + "}"; }

    private static native long open(String path, int device);
    private static native long openCollection(String[] paths, boolean findView, int device);
    private static native void close(long h);
    private static native long[] info(long h);
    private static native String sampleName(long h, int color);
    private static native long[] colorInfo(long h, int color, String[] cleanedAgainstOut);
    private static native int colorForSampleName(long h, String name);
    private static native long[] findRecords(long h, byte[] kmers, int n, int[] cov, byte[] edges);
    private static native void records(long h, long first, int n, long[] words, int[] cov, byte[] edges);
}
