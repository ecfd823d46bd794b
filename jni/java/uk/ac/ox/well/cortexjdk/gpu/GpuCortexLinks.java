package uk.ac.ox.well.cortexjdk.gpu;

import uk.ac.ox.well.cortexjdk.utils.io.graph.ConnectivityAnnotations;
import uk.ac.ox.well.cortexjdk.utils.io.graph.cortex.CortexHeader;
import uk.ac.ox.well.cortexjdk.utils.io.graph.links.CortexJunctionsRecord;
import uk.ac.ox.well.cortexjdk.utils.io.graph.links.CortexLinksRecord;
import uk.ac.ox.well.cortexjdk.utils.kmer.CortexBinaryKmer;

import java.io.File;
import java.util.ArrayList;
import java.util.List;

/**
 * Link annotations (.ctp.gz / .ctp.bgz) bound to a GpuCortexGraph: utils/io/graph/ConnectivityAnnotations.java:15-45.
 * A traversal engine made with these links keeps them in HBM; get()/containsKey() answer from the parsed host copy.
 */
public class GpuCortexLinks implements ConnectivityAnnotations {
    private final File file;
    final long handle;
    private final int version, numColors, kmerSize;
    private final long numKmersInGraph, numKmersWithLinks, numLinks;

    public GpuCortexLinks(String path, GpuCortexGraph graph) { this(new File(path), graph); }
    public GpuCortexLinks(File f, GpuCortexGraph graph) {
        file = f;
        handle = open(f.getAbsolutePath(), graph.handle);
        long[] i = info(handle);
        version = (int) i[0]; numColors = (int) i[1]; kmerSize = (int) i[2];
        numKmersInGraph = i[3]; numKmersWithLinks = i[4]; numLinks = i[5];
    }

    @Override public File getFile() { return file; }
    @Override public int size() { return (int) numKmersWithLinks; }
    @Override public boolean isEmpty() { return numKmersWithLinks == 0; }
    @Override public boolean containsKey(Object key) { return recordText(key) != null; }

    /** CortexLinksMap.get (links/CortexLinksMap.java:22-60): the junction records in the reference's HashSet iteration order */
    @Override public CortexLinksRecord get(Object key) {
        String text = recordText(key);
        if (text == null) { return null; }
        String[] lines = text.split("\n");
        String kmer = lines[0].split("\\s+")[0];
        List<CortexJunctionsRecord> juncs = new ArrayList<>();
        for (int l = 1; l < lines.length; l++) {
            if (lines[l].isEmpty()) { continue; }
            String[] f = lines[l].split("\\s+");            // F|R numJunctions coverages junctions
            String[] covs = f[2].split(",");
            int[] cov = new int[covs.length];
            for (int c = 0; c < covs.length; c++) { cov[c] = Integer.parseInt(covs[c]); }
            juncs.add(new CortexJunctionsRecord(f[0].equals("F"), -1, Integer.parseInt(f[1]), cov, f[3]));
        }
        return new CortexLinksRecord(kmer, juncs);
    }

    private String recordText(Object key) {
        CortexBinaryKmer bk = convert(key);
        return get(handle, new CortexBinaryKmerText(bk, kmerSize).ascii());
    }

    @Override public CortexHeader getHeader() {
        CortexHeader h = new CortexHeader();
        h.setVersion(version); h.setKmerSize(kmerSize); h.setNumColors(numColors);
        return h;
    }
    @Override public String getSource() { return sampleName(handle, 0); }

    public int getVersion() { return version; }
    public long getNumKmersInGraph() { return numKmersInGraph; }
    public long getNumLinks() { return numLinks; }
    public String getSampleName(int color) { return sampleName(handle, color); }
    public void close() { close(handle); }

    /** ASCII form of a CortexBinaryKmer (2 bits per base, CortexRecord.decodeBinaryKmer: CortexRecord.java:311-334) */
    private static final class CortexBinaryKmerText {
        private final long[] words; private final int k;
        CortexBinaryKmerText(CortexBinaryKmer bk, int k) { this.words = bk.getBinaryKmer(); this.k = k; }
        byte[] ascii() {
            return uk.ac.ox.well.cortexjdk.utils.io.graph.cortex.CortexRecord.decodeBinaryKmer(words, k, words.length);
        }
    }

    private static native long open(String path, long graph);
    private static native void close(long h);
    private static native long[] info(long h);
    private static native String sampleName(long h, int color);
    private static native String get(long h, byte[] kmer);
}
