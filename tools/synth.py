"""ctypes wrapper of tools/_build/libldbg_synth.so (synthetic bench inputs; not product, not oracle)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libldbg_synth.so")


class SynthParams(C.Structure):
    _fields_ = [("genome_len", C.c_int64), ("k", C.c_int32), ("n_chrom", C.c_int32), ("colours", C.c_int32),
                ("with_links", C.c_int32), ("seed", C.c_uint64), ("gc", C.c_double), ("snv_rate", C.c_double),
                ("n_indels", C.c_int32), ("n_dnm", C.c_int32), ("n_tandem", C.c_int32), ("n_repeat_families", C.c_int32),
                ("repeat_copies", C.c_int32), ("repeat_len_min", C.c_int32), ("repeat_len_max", C.c_int32),
                ("read_len", C.c_int32), ("read_stride", C.c_int32), ("n_seeds", C.c_int32), ("threads", C.c_int32)]


class SynthStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_records", "n_link_kmers", "n_links", "n_seeds", "n_novel_seeds", "child_len")]


def generate(prefix, genome_len, k, colours=3, with_links=True, seed=0xC0FFEE03, n_chrom=14, gc=0.5, snv_rate=0.001,
             n_indels=2000, n_dnm=500, n_tandem=20, n_repeat_families=200, repeat_copies=6, repeat_len=(60, 400),
             read_len=250, read_stride=8, n_seeds=50000, threads=None):
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    lib = C.CDLL(_SO)
    p = SynthParams(genome_len, k, n_chrom, colours, 1 if with_links else 0, seed, gc, snv_rate, n_indels, n_dnm, n_tandem,
                    n_repeat_families, repeat_copies, repeat_len[0], repeat_len[1], read_len, read_stride, n_seeds,
                    threads or min(16, os.cpu_count() or 1))
    st = SynthStats()
    rc = lib.ldbg_synth_generate(C.byref(p), prefix.encode(), C.byref(st))
    if rc != 0:
        raise RuntimeError("synth failed rc=%d" % rc)
    return {n: getattr(st, n) for n, _ in SynthStats._fields_}
