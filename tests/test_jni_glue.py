"""The JNI boundary (jni/ldbg_jni.c + jni/java/.../gpu/*.java) without a JDK: the glue compiles against a declaration-only jni.h,
every `native` method of the Java classes has its Java_... definition with a matching arity, and every ldbg_* entry point a Java
host needs is called from the glue."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "jni", "ldbg_jni.c")
JAVA = os.path.join(ROOT, "jni", "java", "uk", "ac", "ox", "well", "cortexjdk", "gpu")

# what a Java host binds (the sharded-table entry points are driven from torch.distributed, not from Java)
NEEDED = [
    "ldbg_last_error", "ldbg_device_count", "ldbg_sort_ctx", "ldbg_join_ctx",
    "ldbg_graph_open", "ldbg_graph_close", "ldbg_graph_info", "ldbg_graph_sample_name", "ldbg_graph_color_info", "ldbg_graph_color_for_sample_name",
    "ldbg_graph_records", "ldbg_graph_find_ascii",
    "ldbg_links_open", "ldbg_links_close", "ldbg_links_info", "ldbg_links_sample_name", "ldbg_links_get",
    "ldbg_engine_config_default", "ldbg_engine_create", "ldbg_engine_destroy",
    "ldbg_engine_walk_batch_run", "ldbg_engine_walk_batch_fetch", "ldbg_engine_walk_vertices", "ldbg_engine_walk_roi_hits",
    "ldbg_engine_dfs_batch", "ldbg_dfs_result_sizes", "ldbg_dfs_result_get", "ldbg_dfs_result_walk", "ldbg_dfs_result_free", "ldbg_engine_dfs_kmers_traversed",
    "ldbg_dfs_result_merge", "ldbg_engine_neighbours_batch", "ldbg_engine_assemble",
    "ldbg_engine_seek", "ldbg_engine_has_next", "ldbg_engine_has_previous", "ldbg_engine_next", "ldbg_engine_previous",
]


def test_glue_compiles_against_the_abi():
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                           "-I", os.path.join(ROOT, "include"), GLUE])


def test_every_needed_export_is_called():
    src = open(GLUE).read()
    from corticall_amd._native import EXPORTS
    for name in NEEDED:
        assert name in EXPORTS, name
        assert re.search(r"\b%s\s*\(" % name, src), "%s is not called from jni/ldbg_jni.c" % name


def test_every_native_method_has_its_definition():
    src = open(GLUE).read()
    defs = {}
    for m in re.finditer(r"JNIFN\((\w+),\s*(\w+)\)\(([^)]*)\)", src):
        defs[(m.group(1), m.group(2))] = len([a for a in m.group(3).split(",") if a.strip()]) - 2      # minus JNIEnv*, jclass
    natives = {}
    for fn in sorted(f for f in os.listdir(JAVA) if f.endswith(".java")):
        cls = fn[:-5]
        text = open(os.path.join(JAVA, fn)).read()
        for m in re.finditer(r"private static native [\w\[\]]+ (\w+)\(([^)]*)\);", text, re.S):
            natives[(cls, m.group(1))] = len([a for a in m.group(2).split(",") if a.strip()])
    assert natives, "no native methods found"
    for key, arity in natives.items():
        assert key in defs, "native %s.%s has no Java_... definition in ldbg_jni.c" % key
        assert defs[key] == arity, (key, defs[key], arity)
    for key in defs:
        assert key in natives, "ldbg_jni.c defines %s.%s, which no Java class declares" % key
    # the seam: GpuCortexGraph implements every method of DeBruijnGraph (DeBruijnGraph.java:16-53)
    g = open(os.path.join(JAVA, "GpuCortexGraph.java")).read()
    for method in ("position", "iterator", "hasNext", "next", "remove", "close", "getRecord", "findRecord", "getFile", "getHeader", "getVersion",
                   "getKmerSize", "getKmerBits", "getNumColors", "getNumRecords", "getColors", "hasColor", "getColor", "getColorForSampleName",
                   "getColorsForSampleNames", "getSampleName", "toString"):
        assert re.search(r"public [\w<>\[\], ]+ %s\(" % method, g), method
    l = open(os.path.join(JAVA, "GpuCortexLinks.java")).read()
    for method in ("getFile", "size", "isEmpty", "containsKey", "get", "getHeader", "getSource"):
        assert re.search(r"public [\w<>\[\], ]+ %s\(" % method, l), method
    e = open(os.path.join(JAVA, "GpuTraversalEngine.java")).read()
    for method in ("getConfiguration", "dfs", "walk", "next", "previous", "seek", "hasNext", "hasPrevious", "getNextVertices", "getPrevVertices", "assemble"):
        assert re.search(r"public [\w<>\[\], ]+ %s\(" % method, e), method
