package uk.ac.ox.well.cortexjdk.gpu;

import java.io.File;
import java.util.List;

/** Sort and Join on the device: commands/utils/Sort.java:20-49, commands/utils/Join.java:16-60 (over CortexCollection.java:218-293). */
public final class GpuCortexTools {
    static { System.loadLibrary("ldbg_jni"); }
    private GpuCortexTools() {}

    /** the records of `in` in k-mer order under the rewritten header; returns the number of records */
    public static long sort(File in, File out) { return sort(in.getAbsolutePath(), out.getAbsolutePath(), 0); }

    /** the union of the k-mers of several sorted graphs, each graph's colours side by side */
    public static long join(List<File> ins, File out) {
        String[] paths = new String[ins.size()];
        for (int i = 0; i < paths.length; i++) { paths[i] = ins.get(i).getAbsolutePath(); }
        return join(paths, out.getAbsolutePath(), 0);
    }

    /** CortexGraphWriter over a selection: the header of `in` and its records `indices`, in that order (what FindTips writes) */
    public static void writeRecords(File in, long[] indices, File out) { writeRecords(in.getAbsolutePath(), indices, out.getAbsolutePath()); }

    public static int devices() { return deviceCount(); }

    private static native long sort(String in, String out, int device);
    private static native long join(String[] ins, String out, int device);
    private static native void writeRecords(String in, long[] indices, String out);
    private static native int deviceCount();
}
