/* JNI glue for a Java host: GpuCortexGraph / GpuTraversalEngine (jni/java/...) call these natives, which call the
 * C ABI of include/ldbg.h one to one.  Not compiled in this repository's image (no JDK / jni.h here);
 * build on a host with a JDK:  gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include \
 *                              ldbg_jni.c -L../corticall_amd/_build -lldbg -o libldbg_jni.so
 * Non-zero statuses are rethrown as the exception the reference would have thrown. */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "ldbg.h"

static void rethrow(JNIEnv* env, ldbg_status st) {
    const char* cls = "uk/ac/ox/well/cortexjdk/utils/exceptions/CortexJDKException";
    if (st == LDBG_ERR_NULLPOINTER) cls = "java/lang/NullPointerException";
    else if (st == LDBG_ERR_NOSUCHELEMENT) cls = "java/util/NoSuchElementException";
    else if (st == LDBG_ERR_UNSUPPORTED) cls = "java/lang/UnsupportedOperationException";
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), ldbg_last_error());
}
#define CHECK(call) do { ldbg_status st__ = (call); if (st__ != LDBG_OK) { rethrow(env, st__); return 0; } } while (0)

/* new CortexGraph(path)  ->  long handle */
JNIEXPORT jlong JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuCortexGraph_open(JNIEnv* env, jclass c, jstring path, jint device) {
    const char* p = (*env)->GetStringUTFChars(env, path, NULL);
    ldbg_graph* g = NULL;
    ldbg_status st = ldbg_graph_open(p, device, &g);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (st != LDBG_OK) { rethrow(env, st); return 0; }
    return (jlong)(intptr_t)g;
}
JNIEXPORT void JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuCortexGraph_close(JNIEnv* env, jclass c, jlong h) {
    ldbg_graph_close((ldbg_graph*)(intptr_t)h);
}
/* getKmerSize / getKmerBits / getNumColors / getNumRecords packed into a long[4] */
JNIEXPORT jlongArray JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuCortexGraph_info(JNIEnv* env, jclass c, jlong h) {
    int k, W, C, v; int64_t N;
    CHECK(ldbg_graph_info((ldbg_graph*)(intptr_t)h, &k, &W, &C, &N, &v));
    jlong vals[5] = {k, W, C, N, v};
    jlongArray out = (*env)->NewLongArray(env, 5);
    (*env)->SetLongArrayRegion(env, out, 0, 5, vals);
    return out;
}
/* findRecord in bulk: kmers = n*k ASCII bytes; returns record indices (-1 = null), fills cov (n*C ints) and edges (n*C bytes) */
JNIEXPORT jlongArray JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuCortexGraph_findRecords(JNIEnv* env, jclass c, jlong h,
        jbyteArray kmers, jint n, jintArray cov, jbyteArray edges) {
    jbyte* km = (*env)->GetByteArrayElements(env, kmers, NULL);
    jint* cv = (*env)->GetIntArrayElements(env, cov, NULL);
    jbyte* ed = (*env)->GetByteArrayElements(env, edges, NULL);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    ldbg_status st = ldbg_graph_find_ascii((ldbg_graph*)(intptr_t)h, (const char*)km, n, idx, (uint32_t*)cv, (uint8_t*)ed);
    (*env)->ReleaseByteArrayElements(env, kmers, km, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, cov, cv, 0);
    (*env)->ReleaseByteArrayElements(env, edges, ed, 0);
    jlongArray out = NULL;
    if (st == LDBG_OK) { out = (*env)->NewLongArray(env, n); (*env)->SetLongArrayRegion(env, out, 0, n, (const jlong*)idx); }
    free(idx);
    if (st != LDBG_OK) rethrow(env, st);
    return out;
}
/* Iterator<CortexRecord> / getRecord in bulk */
JNIEXPORT void JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuCortexGraph_records(JNIEnv* env, jclass c, jlong h, jlong first, jint n,
        jlongArray words, jintArray cov, jbyteArray edges) {
    jlong* w = (*env)->GetLongArrayElements(env, words, NULL);
    jint* cv = (*env)->GetIntArrayElements(env, cov, NULL);
    jbyte* ed = (*env)->GetByteArrayElements(env, edges, NULL);
    ldbg_status st = ldbg_graph_records((ldbg_graph*)(intptr_t)h, first, n, (uint64_t*)w, (uint32_t*)cv, (uint8_t*)ed);
    (*env)->ReleaseLongArrayElements(env, words, w, 0);
    (*env)->ReleaseIntArrayElements(env, cov, cv, 0);
    (*env)->ReleaseByteArrayElements(env, edges, ed, 0);
    if (st != LDBG_OK) rethrow(env, st);
}
/* TraversalEngineFactory.make(): colours as int[]; links as long[] of handles */
JNIEXPORT jlong JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuTraversalEngine_create(JNIEnv* env, jclass c, jlong graph, jlong rois,
        jlongArray links, jintArray trav, jintArray join, jintArray recruit, jintArray secondary, jint direction, jint op,
        jint stopper, jint maxLen, jboolean connectAll) {
    ldbg_engine_config cfg;
    ldbg_engine_config_default(&cfg);
    cfg.graph = (const ldbg_graph*)(intptr_t)graph;
    cfg.rois = (const ldbg_graph*)(intptr_t)rois;
    jsize nl = links ? (*env)->GetArrayLength(env, links) : 0;
    const ldbg_links* lk[16];
    if (nl > 16) nl = 16;
    if (nl) { jlong* p = (*env)->GetLongArrayElements(env, links, NULL); for (jsize i = 0; i < nl; i++) lk[i] = (const ldbg_links*)(intptr_t)p[i]; (*env)->ReleaseLongArrayElements(env, links, p, JNI_ABORT); }
    cfg.links = lk; cfg.nlinks = (int)nl;
#define COPY_COLOURS(arr, dst, cnt) do { jsize n_ = (arr) ? (*env)->GetArrayLength(env, (arr)) : 0; if (n_ > LDBG_MAX_COLORS) n_ = LDBG_MAX_COLORS; \
        if (n_) (*env)->GetIntArrayRegion(env, (arr), 0, n_, (jint*)(dst)); (cnt) = (int)n_; } while (0)
    COPY_COLOURS(trav, cfg.traversal_colors, cfg.n_traversal);
    COPY_COLOURS(join, cfg.joining_colors, cfg.n_joining);
    COPY_COLOURS(recruit, cfg.recruitment_colors, cfg.n_recruitment);
    COPY_COLOURS(secondary, cfg.secondary_colors, cfg.n_secondary);
    cfg.direction = direction; cfg.combination_operator = op; cfg.stopping_rule = stopper;
    cfg.max_branch_length = maxLen; cfg.connect_all_neighbors = connectAll ? 1 : 0;
    ldbg_engine* e = NULL;
    CHECK(ldbg_engine_create(&cfg, &e));
    return (jlong)(intptr_t)e;
}
/* walk(seed) for n seeds: returns the contigs back to back; offsets[n+1] */
JNIEXPORT jbyteArray JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuTraversalEngine_walkBatch(JNIEnv* env, jclass c, jlong h,
        jbyteArray seeds, jint n, jlongArray offsets) {
    ldbg_engine* e = (ldbg_engine*)(intptr_t)h;
    jbyte* sd = (*env)->GetByteArrayElements(env, seeds, NULL);
    int64_t total = 0, trav = 0;
    ldbg_status st = ldbg_engine_walk_batch_run(e, (const char*)sd, n, &total, &trav);
    (*env)->ReleaseByteArrayElements(env, seeds, sd, JNI_ABORT);
    if (st != LDBG_OK) { rethrow(env, st); return NULL; }
    jbyteArray out = (*env)->NewByteArray(env, (jsize)total);
    jbyte* o = (*env)->GetByteArrayElements(env, out, NULL);
    jlong* off = (*env)->GetLongArrayElements(env, offsets, NULL);
    st = ldbg_engine_walk_batch_fetch(e, (char*)o, total, (int64_t*)off, NULL);
    (*env)->ReleaseByteArrayElements(env, out, o, 0);
    (*env)->ReleaseLongArrayElements(env, offsets, off, 0);
    if (st != LDBG_OK) { rethrow(env, st); return NULL; }
    return out;
}
/* seek / hasNext / next (previous is symmetrical) */
JNIEXPORT void JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuTraversalEngine_seek(JNIEnv* env, jclass c, jlong h, jbyteArray kmer) {
    jbyte* k = (*env)->GetByteArrayElements(env, kmer, NULL);
    ldbg_status st = ldbg_engine_seek((ldbg_engine*)(intptr_t)h, (const char*)k);
    (*env)->ReleaseByteArrayElements(env, kmer, k, JNI_ABORT);
    if (st != LDBG_OK) rethrow(env, st);
}
JNIEXPORT jboolean JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuTraversalEngine_hasNext(JNIEnv* env, jclass c, jlong h) {
    int yes = 0;
    CHECK(ldbg_engine_has_next((ldbg_engine*)(intptr_t)h, &yes));
    return yes ? JNI_TRUE : JNI_FALSE;
}
JNIEXPORT jlong JNICALL Java_uk_ac_ox_well_cortexjdk_gpu_GpuTraversalEngine_next(JNIEnv* env, jclass c, jlong h, jbyteArray kmerOut) {
    jbyte* k = (*env)->GetByteArrayElements(env, kmerOut, NULL);
    int64_t rec = -1;
    ldbg_status st = ldbg_engine_next((ldbg_engine*)(intptr_t)h, (char*)k, &rec);
    (*env)->ReleaseByteArrayElements(env, kmerOut, k, 0);
    if (st != LDBG_OK) { rethrow(env, st); return -1; }
    return rec;
}
