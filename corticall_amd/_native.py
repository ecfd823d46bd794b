"""ctypes binding of libldbg.so (include/ldbg.h).

The library is the product: hand-written HIP kernels for gfx950 behind a C ABI.  There is no
Python or CPU implementation behind these calls — if the shared object is missing, or no MI355X is
visible, every compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libldbg.so")
if os.environ.get("LDBG_DIAG_LIB") == "1":
    # the same sources built with the walk kernel's timers compiled in (make -C corticall_amd/csrc diag): profiling sessions only (tools/)
    LIB_PATH = os.path.join(_HERE, "_build_diag", "libldbg.so")
elif os.environ.get("LDBG_DIAG_LIB", "").startswith("variant"):
    # a tuning variant (make -C corticall_amd/csrc variant VARIANT_FLAGS=...; copies kept as _build_variantNAME): experiments of tools/ only
    LIB_PATH = os.path.join(_HERE, "_build_" + os.environ["LDBG_DIAG_LIB"], "libldbg.so")

LDBG_OK = 0
STATUS_NAMES = {
    1: "CortexJDKException", 2: "NullPointerException", 3: "NoSuchElementException", 4: "UnsupportedOperation",
    5: "HipError", 6: "IllegalArgument", 7: "CapacityError",
}
MAX_COLORS = 32


class CortexJDKException(RuntimeError):
    """uk.ac.ox.well.cortexjdk.utils.exceptions.CortexJDKException"""


class JavaNullPointerException(RuntimeError):
    pass


class NoSuchElementException(RuntimeError):
    pass


class LdbgError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class EngineConfig(C.Structure):
    _fields_ = [
        ("graph", C.c_void_p), ("rois", C.c_void_p), ("links", C.POINTER(C.c_void_p)), ("nlinks", C.c_int),
        ("traversal_colors", C.c_int * MAX_COLORS), ("n_traversal", C.c_int),
        ("joining_colors", C.c_int * MAX_COLORS), ("n_joining", C.c_int),
        ("recruitment_colors", C.c_int * MAX_COLORS), ("n_recruitment", C.c_int),
        ("secondary_colors", C.c_int * MAX_COLORS), ("n_secondary", C.c_int),
        ("direction", C.c_int), ("combination_operator", C.c_int), ("stopping_rule", C.c_int),
        ("max_branch_length", C.c_int), ("connect_all_neighbors", C.c_int), ("strict_java_flip", C.c_int),
    ]


class ColorInfo(C.Structure):
    _fields_ = [
        ("mean_read_length", C.c_uint32), ("total_sequence", C.c_uint64),
        ("tip_clipping", C.c_uint8), ("low_covg_supernodes_removed", C.c_uint8),
        ("low_covg_kmers_removed", C.c_uint8), ("cleaned_against_graph", C.c_uint8),
        ("low_cov_supernodes_threshold", C.c_uint32), ("low_cov_kmer_threshold", C.c_uint32),
    ]


# every symbol include/ldbg.h declares (tests check that the built library exports all of them)
EXPORTS = [
    "ldbg_last_error", "ldbg_version", "ldbg_device_count", "ldbg_kmer_encode", "ldbg_kmer_decode",
    "ldbg_sort_ctx", "ldbg_join_ctx", "ldbg_ctx_write_records", "ldbg_graph_open", "ldbg_graph_open_memory", "ldbg_graph_open_device", "ldbg_graph_open_collection", "ldbg_graph_close", "ldbg_graph_info", "ldbg_graph_device", "ldbg_graph_set_shard",
    "ldbg_graph_sample_name", "ldbg_graph_color_info", "ldbg_graph_color_for_sample_name",
    "ldbg_graph_records", "ldbg_graph_records_dev", "ldbg_graph_find", "ldbg_graph_find_ascii", "ldbg_graph_find_dev", "ldbg_shard_owner_dev", "ldbg_shard_owner", "ldbg_shard_nbr_queries", "ldbg_shard_set_nbr",
    "ldbg_image_create", "ldbg_image_destroy", "ldbg_image_graph", "ldbg_image_row_bytes", "ldbg_image_clear", "ldbg_image_request", "ldbg_image_reset_requests", "ldbg_image_bucket",
    "ldbg_image_serve", "ldbg_image_serve_chain", "ldbg_image_insert", "ldbg_image_lookup", "ldbg_image_counters",
    "ldbg_engine_sharded_walk_begin", "ldbg_engine_sharded_walk_round", "ldbg_engine_sharded_walk_finish", "ldbg_engine_sharded_dfs_batch",
    "ldbg_links_open", "ldbg_links_close", "ldbg_links_index", "ldbg_links_source", "ldbg_links_info", "ldbg_links_sample_name", "ldbg_links_get",
    "ldbg_engine_config_default", "ldbg_engine_create", "ldbg_engine_destroy",
    "ldbg_engine_walk_batch", "ldbg_engine_walk_batch_run", "ldbg_engine_walk_batch_run_device", "ldbg_engine_walk_batch_fetch", "ldbg_host_alloc", "ldbg_host_free", "ldbg_engine_walk_vertices", "ldbg_engine_walk_roi_hits",
    "ldbg_engine_dfs_batch", "ldbg_dfs_result_sizes", "ldbg_dfs_result_get", "ldbg_dfs_result_walk", "ldbg_dfs_result_merge", "ldbg_dfs_result_free", "ldbg_engine_neighbours_batch", "ldbg_engine_assemble",
    "ldbg_engine_dfs_kmers_traversed",
    "ldbg_engine_seek", "ldbg_engine_has_next", "ldbg_engine_has_previous", "ldbg_engine_next", "ldbg_engine_previous",
    "ldbg_profile_reset", "ldbg_profile_get",
]


class NativeLib:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C corticall_amd/csrc).  corticall_amd has no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        d.ldbg_last_error.restype = C.c_char_p
        d.ldbg_version.restype = C.c_char_p
        for name in EXPORTS:
            getattr(d, name)   # raises AttributeError if a declared symbol is not exported
        d.ldbg_engine_config_default.restype = None
        try:                    # the host simulation of the test suite (tests/hostsim) exports this; libldbg.so does not
            d.ldbg_hostsim_set_lanes
            self.is_hostsim = True
        except AttributeError:
            self.is_hostsim = False

    def check(self, status):
        if status == LDBG_OK:
            return
        msg = self.dll.ldbg_last_error().decode(errors="replace")
        if status == 1:
            raise CortexJDKException(msg)
        if status == 2:
            raise JavaNullPointerException(msg)
        if status == 3:
            raise NoSuchElementException(msg)
        raise LdbgError(status, msg)

    def device_count(self):
        n = C.c_int()
        self.dll.ldbg_device_count(C.byref(n))
        return n.value


_default = None


def default_lib():
    global _default
    if _default is None:
        _default = NativeLib()
    return _default
