/* JNI glue for a Java host: the classes under jni/java/uk/ac/ox/well/cortexjdk/gpu/ (GpuCortexGraph implements DeBruijnGraph,
 * GpuCortexLinks implements ConnectivityAnnotations, GpuTraversalEngine mirrors the TraversalEngine facade, GpuCortexTools = Sort /
 * Join) declare these natives, which call the C ABI of include/ldbg.h one to one.  This repository's image has no JDK, so the glue
 * is shipped as source; tests/test_jni_glue.py checks that it compiles against a declaration-only jni.h (tests/jni_stub), that
 * every `native` method of the Java classes has its Java_... definition here and that every ldbg_* entry point a Java host needs is
 * called.  Build on a host with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/ldbg_jni.c -Lcorticall_amd/_build -lldbg -o libldbg_jni.so
 * Non-zero statuses are rethrown as the exception the reference would have thrown. */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ldbg.h"

#define G(h) ((ldbg_graph*)(intptr_t)(h))
#define L(h) ((ldbg_links*)(intptr_t)(h))
#define E(h) ((ldbg_engine*)(intptr_t)(h))
#define R(h) ((ldbg_dfs_result*)(intptr_t)(h))
#define JNIFN(cls, name) JNIEXPORT JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_##cls##_##name

static void rethrow(JNIEnv* env, ldbg_status st) {
    const char* cls = "uk/ac/ox/well/cortexjdk/utils/exceptions/CortexJDKException";
    if (st == LDBG_ERR_NULLPOINTER) cls = "java/lang/NullPointerException";
    else if (st == LDBG_ERR_NOSUCHELEMENT) cls = "java/util/NoSuchElementException";
    else if (st == LDBG_ERR_UNSUPPORTED) cls = "java/lang/UnsupportedOperationException";
    else if (st == LDBG_ERR_ARG) cls = "java/lang/IllegalArgumentException";
    jclass c = (*env)->FindClass(env, cls);
    if (!c) { (*env)->ExceptionClear(env); c = (*env)->FindClass(env, "java/lang/RuntimeException"); }
    if (c) (*env)->ThrowNew(env, c, ldbg_last_error());
}
#define CHECK(call, ret) do { ldbg_status st__ = (call); if (st__ != LDBG_OK) { rethrow(env, st__); return ret; } } while (0)
static jstring cstr(JNIEnv* env, const char* s) { return (*env)->NewStringUTF(env, s); }

/* ------------------------------------------------------------------ GpuCortexGraph: DeBruijnGraph.java:16-53 */
jlong JNIFN(GpuCortexGraph, open)(JNIEnv* env, jclass c, jstring path, jint device) {
    const char* p = (*env)->GetStringUTFChars(env, path, NULL);
    ldbg_graph* g = NULL;
    ldbg_status st = ldbg_graph_open(p, device, &g);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (st != LDBG_OK) { rethrow(env, st); return 0; }
    return (jlong)(intptr_t)g;
}
/* new CortexCollection(graphs...): the members merged into one resident table; findView: the graph findRecord answers from */
jlong JNIFN(GpuCortexGraph, openCollection)(JNIEnv* env, jclass c, jobjectArray ins, jboolean findView, jint device) {
    jsize np = (*env)->GetArrayLength(env, ins);
    const char** paths = (const char**)calloc((size_t)(np > 0 ? np : 1), sizeof(*paths));
    jstring* held = (jstring*)calloc((size_t)(np > 0 ? np : 1), sizeof(*held));
    for (jsize i = 0; i < np; i++) { held[i] = (jstring)(*env)->GetObjectArrayElement(env, ins, i); paths[i] = (*env)->GetStringUTFChars(env, held[i], NULL); }
    ldbg_graph* g = NULL;
    ldbg_status st = ldbg_graph_open_collection(paths, (int)np, findView ? 1 : 0, device, &g);
    for (jsize i = 0; i < np; i++) (*env)->ReleaseStringUTFChars(env, held[i], paths[i]);
    free(paths); free(held);
    if (st != LDBG_OK) { rethrow(env, st); return 0; }
    return (jlong)(intptr_t)g;
}
void JNIFN(GpuCortexGraph, close)(JNIEnv* env, jclass c, jlong h) { CHECK(ldbg_graph_close(G(h)), ); }
/* getKmerSize / getKmerBits / getNumColors / getNumRecords / getVersion as a long[5] */
jlongArray JNIFN(GpuCortexGraph, info)(JNIEnv* env, jclass c, jlong h) {
    int k, W, C, v; int64_t N;
    CHECK(ldbg_graph_info(G(h), &k, &W, &C, &N, &v), NULL);
    jlong vals[5] = {k, W, C, N, v};
    jlongArray out = (*env)->NewLongArray(env, 5);
    if (out) (*env)->SetLongArrayRegion(env, out, 0, 5, vals);
    return out;
}
jstring JNIFN(GpuCortexGraph, sampleName)(JNIEnv* env, jclass c, jlong h, jint color) {
    char buf[4096];
    CHECK(ldbg_graph_sample_name(G(h), color, buf, (int)sizeof buf), NULL);
    return cstr(env, buf);
}
/* CortexColor fields of one colour: {meanReadLength, totalSequence, tipClipping, lowCovgSupernodesRemoved, lowCovgKmersRemoved,
 * cleanedAgainstGraph, lowCovSupernodesThreshold, lowCovKmerThreshold}; the cleaned-against graph name through nameOut[0] */
jlongArray JNIFN(GpuCortexGraph, colorInfo)(JNIEnv* env, jclass c, jlong h, jint color, jobjectArray nameOut) {
    ldbg_color_info ci;
    char name[4096];
    CHECK(ldbg_graph_color_info(G(h), color, &ci, name, (int)sizeof name), NULL);
    jlong vals[8] = {ci.mean_read_length, (jlong)ci.total_sequence, ci.tip_clipping, ci.low_covg_supernodes_removed, ci.low_covg_kmers_removed,
                     ci.cleaned_against_graph, ci.low_cov_supernodes_threshold, ci.low_cov_kmer_threshold};
    jlongArray out = (*env)->NewLongArray(env, 8);
    if (out) (*env)->SetLongArrayRegion(env, out, 0, 8, vals);
    if (nameOut) (*env)->SetObjectArrayElement(env, nameOut, 0, cstr(env, name));
    return out;
}
jint JNIFN(GpuCortexGraph, colorForSampleName)(JNIEnv* env, jclass c, jlong h, jstring name) {
    const char* p = (*env)->GetStringUTFChars(env, name, NULL);
    int color = -1;
    ldbg_status st = ldbg_graph_color_for_sample_name(G(h), p, &color);
    (*env)->ReleaseStringUTFChars(env, name, p);
    if (st != LDBG_OK) { rethrow(env, st); return -1; }
    return color;
}
/* findRecord in bulk: kmers = n*k ASCII bytes; returns record indices (-1 = null), fills cov (n*C ints) and edges (n*C bytes) */
jlongArray JNIFN(GpuCortexGraph, findRecords)(JNIEnv* env, jclass c, jlong h, jbyteArray kmers, jint n, jintArray cov, jbyteArray edges) {
    jbyte* km = (*env)->GetByteArrayElements(env, kmers, NULL);
    jint* cv = (*env)->GetIntArrayElements(env, cov, NULL);
    jbyte* ed = (*env)->GetByteArrayElements(env, edges, NULL);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    ldbg_status st = ldbg_graph_find_ascii(G(h), (const char*)km, n, idx, (uint32_t*)cv, (uint8_t*)ed);
    (*env)->ReleaseByteArrayElements(env, kmers, km, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, cov, cv, 0);
    (*env)->ReleaseByteArrayElements(env, edges, ed, 0);
    jlongArray out = NULL;
    if (st == LDBG_OK) { out = (*env)->NewLongArray(env, n); if (out) (*env)->SetLongArrayRegion(env, out, 0, n, (const jlong*)idx); }
    free(idx);
    if (st != LDBG_OK) rethrow(env, st);
    return out;
}
/* Iterator<CortexRecord> / getRecord in bulk */
void JNIFN(GpuCortexGraph, records)(JNIEnv* env, jclass c, jlong h, jlong first, jint n, jlongArray words, jintArray cov, jbyteArray edges) {
    jlong* w = (*env)->GetLongArrayElements(env, words, NULL);
    jint* cv = (*env)->GetIntArrayElements(env, cov, NULL);
    jbyte* ed = (*env)->GetByteArrayElements(env, edges, NULL);
    ldbg_status st = ldbg_graph_records(G(h), first, n, (uint64_t*)w, (uint32_t*)cv, (uint8_t*)ed);
    (*env)->ReleaseLongArrayElements(env, words, w, 0);
    (*env)->ReleaseIntArrayElements(env, cov, cv, 0);
    (*env)->ReleaseByteArrayElements(env, edges, ed, 0);
    if (st != LDBG_OK) rethrow(env, st);
}

/* ------------------------------------------------------------------ GpuCortexLinks: ConnectivityAnnotations.java:15-45 */
jlong JNIFN(GpuCortexLinks, open)(JNIEnv* env, jclass c, jstring path, jlong graph) {
    const char* p = (*env)->GetStringUTFChars(env, path, NULL);
    ldbg_links* l = NULL;
    ldbg_status st = ldbg_links_open(p, G(graph), &l);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (st != LDBG_OK) { rethrow(env, st); return 0; }
    return (jlong)(intptr_t)l;
}
void JNIFN(GpuCortexLinks, close)(JNIEnv* env, jclass c, jlong h) { CHECK(ldbg_links_close(L(h)), ); }
/* {version, numColors, kmerSize, numKmersInGraph, numKmersWithLinks, numLinks} */
jlongArray JNIFN(GpuCortexLinks, info)(JNIEnv* env, jclass c, jlong h) {
    int v, nc, k; int64_t a, b, d;
    CHECK(ldbg_links_info(L(h), &v, &nc, &k, &a, &b, &d), NULL);
    jlong vals[6] = {v, nc, k, a, b, d};
    jlongArray out = (*env)->NewLongArray(env, 6);
    if (out) (*env)->SetLongArrayRegion(env, out, 0, 6, vals);
    return out;
}
jstring JNIFN(GpuCortexLinks, sampleName)(JNIEnv* env, jclass c, jlong h, jint color) {
    char buf[4096];
    CHECK(ldbg_links_sample_name(L(h), color, buf, (int)sizeof buf), NULL);
    return cstr(env, buf);
}
/* get(key): the record as the text lines of the link file ("KMER n" + one line per junction record, in the reference's HashSet order),
 * or null when the k-mer has no links (containsKey == false) */
jstring JNIFN(GpuCortexLinks, get)(JNIEnv* env, jclass c, jlong h, jbyteArray kmer) {
    jsize k = (*env)->GetArrayLength(env, kmer);
    char* km = (char*)malloc((size_t)k + 1);
    (*env)->GetByteArrayRegion(env, kmer, 0, k, (jbyte*)km);
    km[k] = 0;
    int found = 0;
    int64_t cap = 1 << 16;
    char* buf = (char*)malloc((size_t)cap);
    ldbg_status st = ldbg_links_get(L(h), km, &found, buf, cap);
    while (st == LDBG_ERR_CAPACITY && cap < (1LL << 30)) { cap *= 8; buf = (char*)realloc(buf, (size_t)cap); st = ldbg_links_get(L(h), km, &found, buf, cap); }
    jstring out = NULL;
    if (st == LDBG_OK && found) out = cstr(env, buf);
    free(buf); free(km);
    if (st != LDBG_OK) rethrow(env, st);
    return out;
}

/* ------------------------------------------------------------------ GpuTraversalEngine: TraversalEngine.java:33-339 */
/* TraversalEngineFactory.make(): colours as int[]; links as long[] of handles (the engine itself refuses more than it supports) */
jlong JNIFN(GpuTraversalEngine, create)(JNIEnv* env, jclass c, jlong graph, jlong rois, jlongArray links, jintArray trav, jintArray join,
        jintArray recruit, jintArray secondary, jint direction, jint op, jint stopper, jint maxLen, jboolean connectAll) {
    ldbg_engine_config cfg;
    ldbg_engine_config_default(&cfg);
    cfg.graph = G(graph);
    cfg.rois = G(rois);
    jsize nl = links ? (*env)->GetArrayLength(env, links) : 0;
    const ldbg_links** lk = (const ldbg_links**)calloc((size_t)(nl > 0 ? nl : 1), sizeof(*lk));
    if (nl) { jlong* p = (*env)->GetLongArrayElements(env, links, NULL); for (jsize i = 0; i < nl; i++) lk[i] = L(p[i]); (*env)->ReleaseLongArrayElements(env, links, p, JNI_ABORT); }
    cfg.links = lk; cfg.nlinks = (int)nl;
#define COPY_COLOURS(arr, dst, cnt) do { jsize n_ = (arr) ? (*env)->GetArrayLength(env, (arr)) : 0; \
        if (n_ > LDBG_MAX_COLORS) { free(lk); rethrow(env, LDBG_ERR_ARG); return 0; } \
        if (n_) { (*env)->GetIntArrayRegion(env, (arr), 0, n_, (jint*)(dst)); } \
        (cnt) = (int)n_; } while (0)
    COPY_COLOURS(trav, cfg.traversal_colors, cfg.n_traversal);
    COPY_COLOURS(join, cfg.joining_colors, cfg.n_joining);
    COPY_COLOURS(recruit, cfg.recruitment_colors, cfg.n_recruitment);
    COPY_COLOURS(secondary, cfg.secondary_colors, cfg.n_secondary);
    cfg.direction = direction; cfg.combination_operator = op; cfg.stopping_rule = stopper;
    cfg.max_branch_length = maxLen; cfg.connect_all_neighbors = connectAll ? 1 : 0;
    ldbg_engine* e = NULL;
    ldbg_status st = ldbg_engine_create(&cfg, &e);
    free(lk);
    if (st != LDBG_OK) { rethrow(env, st); return 0; }
    return (jlong)(intptr_t)e;
}
void JNIFN(GpuTraversalEngine, destroy)(JNIEnv* env, jclass c, jlong h) { CHECK(ldbg_engine_destroy(E(h)), ); }
/* walk(seed) + toContig for n seeds: returns the contigs back to back; offsets[n+1]; walkLen[n] vertices per walk */
jbyteArray JNIFN(GpuTraversalEngine, walkBatch)(JNIEnv* env, jclass c, jlong h, jbyteArray seeds, jint n, jlongArray offsets, jlongArray walkLen) {
    jbyte* sd = (*env)->GetByteArrayElements(env, seeds, NULL);
    int64_t total = 0, trav = 0;
    ldbg_status st = ldbg_engine_walk_batch_run(E(h), (const char*)sd, n, &total, &trav);
    (*env)->ReleaseByteArrayElements(env, seeds, sd, JNI_ABORT);
    if (st != LDBG_OK) { rethrow(env, st); return NULL; }
    jbyteArray out = (*env)->NewByteArray(env, (jsize)total);
    if (!out) return NULL;
    jbyte* o = (*env)->GetByteArrayElements(env, out, NULL);
    jlong* off = (*env)->GetLongArrayElements(env, offsets, NULL);
    jlong* wl = walkLen ? (*env)->GetLongArrayElements(env, walkLen, NULL) : NULL;
    st = ldbg_engine_walk_batch_fetch(E(h), (char*)o, total, (int64_t*)off, (int64_t*)wl);
    (*env)->ReleaseByteArrayElements(env, out, o, 0);
    (*env)->ReleaseLongArrayElements(env, offsets, off, 0);
    if (wl) (*env)->ReleaseLongArrayElements(env, walkLen, wl, 0);
    if (st != LDBG_OK) { rethrow(env, st); return NULL; }
    return out;
}
/* List<CortexVertex> of walk i of the last batch: returns the number of vertices; with arrays of that capacity fills k-mer words
 * (len x W), record index (-1 = null CortexRecord), copyIndex, index.  Call with capacity 0 (arrays null) to learn the length. */
jlong JNIFN(GpuTraversalEngine, walkVertices)(JNIEnv* env, jclass c, jlong h, jlong walk, jlong capacity, jlongArray words, jlongArray rec,
        jintArray copyIndex, jintArray index) {
    int64_t len = 0;
    if (capacity <= 0 || !words) {
        ldbg_status st = ldbg_engine_walk_vertices(E(h), walk, 0, &len, NULL, NULL, NULL, NULL);
        if (st != LDBG_OK && st != LDBG_ERR_CAPACITY) { rethrow(env, st); return -1; }
        return len;
    }
    jlong* w = (*env)->GetLongArrayElements(env, words, NULL);
    jlong* r = (*env)->GetLongArrayElements(env, rec, NULL);
    jint* ci = (*env)->GetIntArrayElements(env, copyIndex, NULL);
    jint* ix = (*env)->GetIntArrayElements(env, index, NULL);
    ldbg_status st = ldbg_engine_walk_vertices(E(h), walk, capacity, &len, (uint64_t*)w, (int64_t*)r, (int32_t*)ci, (int32_t*)ix);
    (*env)->ReleaseLongArrayElements(env, words, w, 0);
    (*env)->ReleaseLongArrayElements(env, rec, r, 0);
    (*env)->ReleaseIntArrayElements(env, copyIndex, ci, 0);
    (*env)->ReleaseIntArrayElements(env, index, ix, 0);
    if (st != LDBG_OK) { rethrow(env, st); return -1; }
    return len;
}
/* Partition.markUsedRois inputs: ROI record numbers every walk of the last batch passes through; offsets[n+1]; hasNull[n] */
jintArray JNIFN(GpuTraversalEngine, walkRoiHits)(JNIEnv* env, jclass c, jlong h, jint n, jlongArray offsets, jbyteArray hasNull) {
    jlong* off = (*env)->GetLongArrayElements(env, offsets, NULL);
    jbyte* hn = (*env)->GetByteArrayElements(env, hasNull, NULL);
    int64_t cap = 1 << 16;
    uint32_t* hits = (uint32_t*)malloc((size_t)cap * 4);
    ldbg_status st = ldbg_engine_walk_roi_hits(E(h), (int64_t*)off, hits, cap, (uint8_t*)hn);
    if (st == LDBG_ERR_CAPACITY) { cap = off[n]; hits = (uint32_t*)realloc(hits, (size_t)(cap > 0 ? cap : 1) * 4); st = ldbg_engine_walk_roi_hits(E(h), (int64_t*)off, hits, cap, (uint8_t*)hn); }
    jintArray out = NULL;
    if (st == LDBG_OK) { out = (*env)->NewIntArray(env, (jsize)off[n]); if (out) (*env)->SetIntArrayRegion(env, out, 0, (jsize)off[n], (const jint*)hits); }
    free(hits);
    (*env)->ReleaseLongArrayElements(env, offsets, off, 0);
    (*env)->ReleaseByteArrayElements(env, hasNull, hn, 0);
    if (st != LDBG_OK) rethrow(env, st);
    return out;
}
/* seek / hasNext / hasPrevious / next / previous: TraversalEngine.java:241-339 */
void JNIFN(GpuTraversalEngine, seek)(JNIEnv* env, jclass c, jlong h, jbyteArray kmer) {
    jsize k = (*env)->GetArrayLength(env, kmer);
    char* km = (char*)malloc((size_t)k + 1);
    (*env)->GetByteArrayRegion(env, kmer, 0, k, (jbyte*)km);
    km[k] = 0;
    ldbg_status st = ldbg_engine_seek(E(h), km);
    free(km);
    if (st != LDBG_OK) rethrow(env, st);
}
jboolean JNIFN(GpuTraversalEngine, hasNext)(JNIEnv* env, jclass c, jlong h) {
    int yes = 0;
    CHECK(ldbg_engine_has_next(E(h), &yes), JNI_FALSE);
    return yes ? JNI_TRUE : JNI_FALSE;
}
jboolean JNIFN(GpuTraversalEngine, hasPrevious)(JNIEnv* env, jclass c, jlong h) {
    int yes = 0;
    CHECK(ldbg_engine_has_previous(E(h), &yes), JNI_FALSE);
    return yes ? JNI_TRUE : JNI_FALSE;
}
static jlong cursor_step(JNIEnv* env, jlong h, jbyteArray kmerOut, int forward) {
    jsize k = (*env)->GetArrayLength(env, kmerOut);
    char* km = (char*)malloc((size_t)k + 1);
    int64_t rec = -1;
    ldbg_status st = forward ? ldbg_engine_next(E(h), km, &rec) : ldbg_engine_previous(E(h), km, &rec);
    if (st == LDBG_OK) (*env)->SetByteArrayRegion(env, kmerOut, 0, k, (const jbyte*)km);
    free(km);
    if (st != LDBG_OK) { rethrow(env, st); return -1; }
    return rec;
}
jlong JNIFN(GpuTraversalEngine, next)(JNIEnv* env, jclass c, jlong h, jbyteArray kmerOut) { return cursor_step(env, h, kmerOut, 1); }
jlong JNIFN(GpuTraversalEngine, previous)(JNIEnv* env, jclass c, jlong h, jbyteArray kmerOut) { return cursor_step(env, h, kmerOut, 0); }
/* dfs(source, sinks...) for n sources: sinks as CSR over ASCII k-mers (sinkOffsets[n+1], may be null) -> result handle */
jlong JNIFN(GpuTraversalEngine, dfsBatch)(JNIEnv* env, jclass c, jlong h, jbyteArray sources, jint n, jbyteArray sinks, jlongArray sinkOffsets) {
    jbyte* src = (*env)->GetByteArrayElements(env, sources, NULL);
    jbyte* sk = sinks ? (*env)->GetByteArrayElements(env, sinks, NULL) : NULL;
    jlong* so = sinkOffsets ? (*env)->GetLongArrayElements(env, sinkOffsets, NULL) : NULL;
    ldbg_dfs_result* r = NULL;
    ldbg_status st = ldbg_engine_dfs_batch(E(h), (const char*)src, n, (const char*)sk, (const int64_t*)so, &r);
    (*env)->ReleaseByteArrayElements(env, sources, src, JNI_ABORT);
    if (sk) (*env)->ReleaseByteArrayElements(env, sinks, sk, JNI_ABORT);
    if (so) (*env)->ReleaseLongArrayElements(env, sinkOffsets, so, JNI_ABORT);
    if (st != LDBG_OK) { rethrow(env, st); return 0; }
    return (jlong)(intptr_t)r;
}
/* dfs(Collection<String> sources, Collection<String> sinks) (TraversalEngine.java:37-62): results 0 .. n-1 of `res` merged in that order
 * with Graphs.addGraph -> a result handle holding ONE graph (index 0) */
jlong JNIFN(GpuTraversalEngine, dfsMerge)(JNIEnv* env, jclass c, jlong res, jint n) {
    int64_t* which = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
    for (jint i = 0; i < n; i++) which[i] = i;
    ldbg_dfs_result* out = NULL;
    ldbg_status st = ldbg_dfs_result_merge(R(res), which, n, &out);
    free(which);
    if (st != LDBG_OK) { rethrow(env, st); return 0; }
    return (jlong)(intptr_t)out;
}
/* getNextVertices / getPrevVertices (TraversalEngine.java:147-239) of n k-mers: offsets[n+1]; words[4 n W], rec[4 n] filled up to offsets[n] */
void JNIFN(GpuTraversalEngine, neighboursBatch)(JNIEnv* env, jclass c, jlong h, jbyteArray kmers, jint n, jboolean forward, jlongArray offsets, jlongArray words, jlongArray rec) {
    jbyte* km = (*env)->GetByteArrayElements(env, kmers, NULL);
    jlong* off = (*env)->GetLongArrayElements(env, offsets, NULL);
    jlong* w = (*env)->GetLongArrayElements(env, words, NULL);
    jlong* r = (*env)->GetLongArrayElements(env, rec, NULL);
    ldbg_status st = ldbg_engine_neighbours_batch(E(h), (const char*)km, n, forward ? 1 : 0, (int64_t*)off, (uint64_t*)w, (int64_t*)r, 4 * (int64_t)n);
    (*env)->ReleaseByteArrayElements(env, kmers, km, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, offsets, off, 0);
    (*env)->ReleaseLongArrayElements(env, words, w, 0);
    (*env)->ReleaseLongArrayElements(env, rec, r, 0);
    if (st != LDBG_OK) rethrow(env, st);
}
/* assemble(seed) (TraversalEngine.java:112-145): returns the number of vertices; with arrays of that capacity fills words (len x W) and rec */
jlong JNIFN(GpuTraversalEngine, assemble)(JNIEnv* env, jclass c, jlong h, jbyteArray seed, jlong capacity, jlongArray words, jlongArray rec) {
    jsize k = (*env)->GetArrayLength(env, seed);
    char* sd = (char*)malloc((size_t)k + 1);
    (*env)->GetByteArrayRegion(env, seed, 0, k, (jbyte*)sd);
    sd[k] = 0;
    int64_t len = 0;
    ldbg_status st;
    if (capacity <= 0 || !words) {
        st = ldbg_engine_assemble(E(h), sd, 0, &len, NULL, NULL);
        free(sd);
        if (st != LDBG_OK && st != LDBG_ERR_CAPACITY) { rethrow(env, st); return -1; }
        return len;
    }
    jlong* w = (*env)->GetLongArrayElements(env, words, NULL);
    jlong* r = (*env)->GetLongArrayElements(env, rec, NULL);
    st = ldbg_engine_assemble(E(h), sd, capacity, &len, (uint64_t*)w, (int64_t*)r);
    free(sd);
    (*env)->ReleaseLongArrayElements(env, words, w, 0);
    (*env)->ReleaseLongArrayElements(env, rec, r, 0);
    if (st != LDBG_OK) { rethrow(env, st); return -1; }
    return len;
}
/* {isNull, vertices, edges} of result i */
jlongArray JNIFN(GpuTraversalEngine, dfsSizes)(JNIEnv* env, jclass c, jlong res, jlong i) {
    int is_null = 1; int64_t nv = 0, ne = 0;
    CHECK(ldbg_dfs_result_sizes(R(res), i, &is_null, &nv, &ne), NULL);
    jlong vals[3] = {is_null, nv, ne};
    jlongArray out = (*env)->NewLongArray(env, 3);
    if (out) (*env)->SetLongArrayRegion(env, out, 0, 3, vals);
    return out;
}
/* the DirectedWeightedPseudograph of result i: vertices (k-mer words nv x W, record, copyIndex, index) and edges (source, target, colour) in insertion order */
void JNIFN(GpuTraversalEngine, dfsGet)(JNIEnv* env, jclass c, jlong res, jlong i, jlongArray words, jlongArray rec, jintArray copyIndex, jintArray index,
        jintArray edgeSrc, jintArray edgeDst, jintArray edgeColor) {
    jlong* w = (*env)->GetLongArrayElements(env, words, NULL);
    jlong* r = (*env)->GetLongArrayElements(env, rec, NULL);
    jint* ci = (*env)->GetIntArrayElements(env, copyIndex, NULL);
    jint* ix = (*env)->GetIntArrayElements(env, index, NULL);
    jint* es = (*env)->GetIntArrayElements(env, edgeSrc, NULL);
    jint* ed = (*env)->GetIntArrayElements(env, edgeDst, NULL);
    jint* ec = (*env)->GetIntArrayElements(env, edgeColor, NULL);
    ldbg_status st = ldbg_dfs_result_get(R(res), i, (uint64_t*)w, (int64_t*)r, (int32_t*)ci, (int32_t*)ix, (int32_t*)es, (int32_t*)ed, (int32_t*)ec);
    (*env)->ReleaseLongArrayElements(env, words, w, 0);
    (*env)->ReleaseLongArrayElements(env, rec, r, 0);
    (*env)->ReleaseIntArrayElements(env, copyIndex, ci, 0);
    (*env)->ReleaseIntArrayElements(env, index, ix, 0);
    (*env)->ReleaseIntArrayElements(env, edgeSrc, es, 0);
    (*env)->ReleaseIntArrayElements(env, edgeDst, ed, 0);
    (*env)->ReleaseIntArrayElements(env, edgeColor, ec, 0);
    if (st != LDBG_OK) rethrow(env, st);
}
/* TraversalUtils.toWalk(g, seed, colour) + toContig on result i */
jstring JNIFN(GpuTraversalEngine, dfsWalk)(JNIEnv* env, jclass c, jlong res, jlong i, jbyteArray seed, jint color) {
    jsize k = (*env)->GetArrayLength(env, seed);
    char* km = (char*)malloc((size_t)k + 1);
    (*env)->GetByteArrayRegion(env, seed, 0, k, (jbyte*)km);
    km[k] = 0;
    int64_t cap = 1 << 16, len = 0;
    char* buf = (char*)malloc((size_t)cap);
    ldbg_status st = ldbg_dfs_result_walk(R(res), i, km, color, buf, cap, &len);
    if (st == LDBG_ERR_CAPACITY) { cap = len + 16; buf = (char*)realloc(buf, (size_t)cap); st = ldbg_dfs_result_walk(R(res), i, km, color, buf, cap, &len); }
    jstring out = NULL;
    if (st == LDBG_OK) { buf[len < cap ? len : cap - 1] = 0; out = cstr(env, buf); }
    free(buf); free(km);
    if (st != LDBG_OK) rethrow(env, st);
    return out;
}
void JNIFN(GpuTraversalEngine, dfsFree)(JNIEnv* env, jclass c, jlong res) { CHECK(ldbg_dfs_result_free(R(res)), ); }
jlong JNIFN(GpuTraversalEngine, dfsKmersTraversed)(JNIEnv* env, jclass c, jlong h) {
    int64_t n = 0;
    CHECK(ldbg_engine_dfs_kmers_traversed(E(h), &n), 0);
    return n;
}

/* ------------------------------------------------------------------ GpuCortexTools: commands/utils/Sort.java:20-49, Join.java:16-60 */
jlong JNIFN(GpuCortexTools, sort)(JNIEnv* env, jclass c, jstring in, jstring out, jint device) {
    const char* a = (*env)->GetStringUTFChars(env, in, NULL);
    const char* b = (*env)->GetStringUTFChars(env, out, NULL);
    int64_t n = 0;
    ldbg_status st = ldbg_sort_ctx(a, b, device, &n);
    (*env)->ReleaseStringUTFChars(env, in, a);
    (*env)->ReleaseStringUTFChars(env, out, b);
    if (st != LDBG_OK) { rethrow(env, st); return -1; }
    return n;
}
jlong JNIFN(GpuCortexTools, join)(JNIEnv* env, jclass c, jobjectArray ins, jstring out, jint device) {
    jsize np = (*env)->GetArrayLength(env, ins);
    const char** paths = (const char**)calloc((size_t)(np > 0 ? np : 1), sizeof(*paths));
    jstring* held = (jstring*)calloc((size_t)(np > 0 ? np : 1), sizeof(*held));
    for (jsize i = 0; i < np; i++) { held[i] = (jstring)(*env)->GetObjectArrayElement(env, ins, i); paths[i] = (*env)->GetStringUTFChars(env, held[i], NULL); }
    const char* b = (*env)->GetStringUTFChars(env, out, NULL);
    int64_t n = 0;
    ldbg_status st = ldbg_join_ctx(paths, (int)np, b, device, &n);
    for (jsize i = 0; i < np; i++) (*env)->ReleaseStringUTFChars(env, held[i], paths[i]);
    (*env)->ReleaseStringUTFChars(env, out, b);
    free(paths); free(held);
    if (st != LDBG_OK) { rethrow(env, st); return -1; }
    return n;
}
/* CortexGraphWriter over a selection of records (FindTips.java:112-131 and the other filters) */
void JNIFN(GpuCortexTools, writeRecords)(JNIEnv* env, jclass c, jstring in, jlongArray indices, jstring out) {
    const char* a = (*env)->GetStringUTFChars(env, in, NULL);
    const char* b = (*env)->GetStringUTFChars(env, out, NULL);
    jsize n = (*env)->GetArrayLength(env, indices);
    jlong* idx = (*env)->GetLongArrayElements(env, indices, NULL);
    ldbg_status st = ldbg_ctx_write_records(a, (const int64_t*)idx, (int64_t)n, b);
    (*env)->ReleaseLongArrayElements(env, indices, idx, JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, in, a);
    (*env)->ReleaseStringUTFChars(env, out, b);
    if (st != LDBG_OK) { rethrow(env, st); }
}
jint JNIFN(GpuCortexTools, deviceCount)(JNIEnv* env, jclass c) {
    int n = 0;
    CHECK(ldbg_device_count(&n), 0);
    return n;
}
