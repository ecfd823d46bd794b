"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Two regimes:

* replicas — the table fits one GPU (configs C2-C4 are ~2 GB): every rank loads the graph, the seeds are
  partitioned, there is NO data-path collective; `partition()` / `gather_strings()` below are all it needs.
* hash-sharded table — `ShardedCortexGraph`: rank r keeps the records whose canonical k-mer hashes to r (still
  sorted, so the per-shard lookup is the unchanged HIP find kernel).  A batch of findRecord queries is routed
  with ONE exchange each way: canonicalise + owner on the device (ldbg_shard_owner_dev), bucket by owner,
  all-to-all the queries, local find (ldbg_graph_find_dev), all-to-all the answers back.

torch is plumbing here (device buffers, process group, collectives); every computation on k-mers is a kernel
behind the C ABI.
"""
import ctypes as C
import os
import struct

import numpy as np

from . import _native
from .graph import CortexGraph


def partition(n, rank, world):
    """contiguous share of n independent units (seeds) for `rank` -> (first, count)"""
    per, extra = divmod(int(n), int(world))
    first = rank * per + min(rank, extra)
    return first, per + (1 if rank < extra else 0)


def gather_strings(local, group=None):
    """all ranks' lists of strings, concatenated in rank order, on every rank"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, list(local), group=group)
    return [s for part in out for s in part]


def ctx_header(buf):
    """Cortex v6 header (CortexGraph.java:66-168, docs/ctx_spec.md) -> dict(k, W, C, data_offset)"""
    if bytes(buf[0:6]).upper() != b"CORTEX":      # equalsIgnoreCase, like the reference and the C++ parser
        raise _native.CortexJDKException("The file does not appear to be a Cortex graph (bad magic)")
    version, k, W, Cc = struct.unpack_from("<IIII", buf, 6)
    if version != 6:
        raise _native.CortexJDKException("The file is a Cortex graph of version %d; only version 6 is supported" % version)
    p = 22 + 4 * Cc + 8 * Cc
    for _ in range(Cc):
        (ln,) = struct.unpack_from("<I", buf, p)
        p += 4 + ln
    p += 16 * Cc
    for _ in range(Cc):
        p += 4 + 4 + 4
        (ln,) = struct.unpack_from("<I", buf, p)
        p += 4 + ln
    if bytes(buf[p:p + 6]).upper() != b"CORTEX":
        raise _native.CortexJDKException("The Cortex graph header does not end with the magic word")
    return {"k": k, "W": W, "C": Cc, "data_offset": p + 6}


_LUT = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _LUT[_c] = _i
    _LUT[ord(chr(_c).lower())] = _i


def pack_kmers(ascii_kmers, k):
    """ASCII u8[n,k] -> packed words u64[n,W] (word 0 most significant); a non-ACGT k-mer gets all-ones words,
    which every kernel treats as "not a k-mer" (Q4)"""
    a = np.ascontiguousarray(ascii_kmers, dtype=np.uint8).reshape(-1, k)
    W = (k + 31) // 32
    codes = _LUT[a]
    bad = (codes == 255).any(axis=1)
    codes = np.where(codes == 255, 0, codes).astype(np.uint64)
    words = np.zeros((a.shape[0], W), dtype=np.uint64)
    for i in range(k):
        bit = 2 * (k - 1 - i)
        words[:, W - 1 - (bit >> 6)] |= codes[:, i] << np.uint64(bit & 63)
    words[bad] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return words


class ShardedCortexGraph:
    """A .ctx table hash-partitioned over the ranks of a process group."""

    def __init__(self, path, device=0, lib=None, group=None, chunk_records=1 << 22):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._lib = lib or _native.default_lib()
        self._d = self._lib.dll
        self.path = str(path)
        raw = np.memmap(self.path, dtype=np.uint8, mode="r")
        h = ctx_header(raw)
        self.k, self.W, self.C = h["k"], h["W"], h["C"]
        rec = 8 * self.W + 5 * self.C
        n_all = (raw.size - h["data_offset"]) // rec
        records = raw[h["data_offset"]:h["data_offset"] + n_all * rec].reshape(n_all, rec)
        # cut this rank's shard: owner of every record's k-mer, a chunk at a time (the rule lives in the library)
        mine = []
        for lo in range(0, n_all, chunk_records):
            blk = records[lo:lo + chunk_records]
            keys = np.ascontiguousarray(blk[:, :8 * self.W]).view("<u8").reshape(-1, self.W)
            owner = np.empty(len(blk), dtype=np.int32)
            self._lib.check(self._d.ldbg_shard_owner(self.k, keys.ctypes.data_as(C.c_void_p), C.c_int64(len(blk)), self.world,
                                                     int(device), owner.ctypes.data_as(C.c_void_p)))
            mine.append(np.ascontiguousarray(blk[owner == self.rank]))
        shard = np.concatenate(mine) if mine else np.zeros((0, rec), dtype=np.uint8)
        image = np.concatenate([np.asarray(raw[:h["data_offset"]]), shard.reshape(-1)])
        self.shard = CortexGraph(self.path + "#shard%d" % self.rank, device=device, lib=self._lib, image=image)
        self._lib.check(self._d.ldbg_graph_set_shard(self.shard._h, 1))
        self.device = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
        n = torch.tensor([self.shard.getNumRecords()], dtype=torch.int64, device=self.device)
        dist.all_reduce(n, group=group)
        self.numRecords = int(n.item())

    def getNumRecords(self): return self.numRecords
    def getKmerSize(self): return self.k
    def getNumColors(self): return self.C

    def _ptr(self, t):
        """device pointer of a tensor the library is about to read or write.  The library works on its own HIP stream and
        returns when its work is done; torch's stream is drained first so that the buffer is ready."""
        if t.is_cuda:
            self._torch.cuda.current_stream(t.device).synchronize()
        return C.c_void_p(t.data_ptr())

    def find_batch(self, kmers):
        """findRecord for this rank's queries (every rank calls, with its own batch — possibly empty).
        kmers: ASCII np.uint8[n,k] or list of str -> (found bool[n], cov i32[n,C], edges u8[n,C], owner i32[n], local_idx i64[n])"""
        torch = self._torch
        if not isinstance(kmers, np.ndarray):
            kmers = np.frombuffer(b"".join(x.encode() if isinstance(x, str) else bytes(x) for x in kmers), dtype=np.uint8)
        q = torch.from_numpy(pack_kmers(kmers, self.k).view(np.int64)).to(self.device)
        found, l_cov, l_edges, owner, l_idx = self.find_packed_dev(q)
        return (found.cpu().numpy(), l_cov.cpu().numpy(), l_edges.cpu().numpy(), owner.cpu().numpy(), l_idx.cpu().numpy())

    def find_packed_dev(self, q, return_canonical=False):
        """device form: q = packed k-mer words, int64[n, W] on this rank's device (all-ones words = not a k-mer)
        -> (found, cov, edges, owner, local_idx) as device tensors (+ the canonical words when asked); nothing touches
        the host but the split sizes"""
        torch, dist = self._torch, self._dist
        n, W, Cc, world = q.shape[0], self.W, self.C, self.world
        canon = torch.empty_like(q)
        owner = torch.empty(max(1, n), dtype=torch.int32, device=self.device)[:n]
        if n:
            self._lib.check(self._d.ldbg_shard_owner_dev(self.k, self._ptr(q), C.c_int64(n), world, self._ptr(canon), self._ptr(owner), None))
        # bucket by owner
        order = torch.argsort(owner.to(torch.int64), stable=True)
        send = canon[order].contiguous()
        send_counts = torch.bincount(owner.to(torch.int64), minlength=world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self._group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        m = int(sum(rc))
        recv = torch.empty((m, W), dtype=torch.int64, device=self.device)
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=self._group)
        # local lookups
        idx = torch.full((max(1, m),), -1, dtype=torch.int64, device=self.device)[:m]
        cov = torch.zeros((max(1, m), Cc), dtype=torch.int32, device=self.device)[:m]
        edges = torch.zeros((max(1, m), Cc), dtype=torch.uint8, device=self.device)[:m]
        if m:
            self._lib.check(self._d.ldbg_graph_find_dev(self.shard._h, self._ptr(recv), C.c_int64(m), self._ptr(idx), self._ptr(cov), self._ptr(edges), None))
        # answers travel back along the same splits
        r_idx = torch.empty(n, dtype=torch.int64, device=self.device)
        r_cov = torch.empty((n, Cc), dtype=torch.int32, device=self.device)
        r_edges = torch.empty((n, Cc), dtype=torch.uint8, device=self.device)
        dist.all_to_all_single(r_idx, idx.contiguous(), output_split_sizes=sc, input_split_sizes=rc, group=self._group)
        dist.all_to_all_single(r_cov, cov.contiguous(), output_split_sizes=sc, input_split_sizes=rc, group=self._group)
        dist.all_to_all_single(r_edges, edges.contiguous(), output_split_sizes=sc, input_split_sizes=rc, group=self._group)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(n, device=self.device)
        l_idx, l_cov, l_edges = r_idx[inv], r_cov[inv], r_edges[inv]
        found = l_idx >= 0
        if self.numRecords <= 2:       # Q1 is a property of the whole graph (CortexGraph.java:274-282)
            found = torch.zeros_like(found)
            l_idx = torch.full_like(l_idx, -1)
            l_cov, l_edges = torch.zeros_like(l_cov), torch.zeros_like(l_edges)
        if return_canonical:
            return found, l_cov, l_edges, owner, l_idx, canon
        return found, l_cov, l_edges, owner, l_idx

    # ---- walks over the partitioned table (csrc/shard.cpp)
    def build_neighbour_index(self, chunk_records=1 << 18):
        """global neighbour index of this shard: for every local record and each of its 8 possible neighbours the owner,
        the record number in the owner's shard and the orientation — one routed findRecord per edge, done once.
        Collective: every rank calls it."""
        torch, dist = self._torch, self._dist
        n_local = self.shard.getNumRecords()
        rounds = torch.tensor([(n_local + chunk_records - 1) // chunk_records], dtype=torch.int64, device=self.device)
        dist.all_reduce(rounds, op=dist.ReduceOp.MAX, group=self._group)
        for r in range(int(rounds.item())):
            first = min(n_local, r * chunk_records)
            n = max(0, min(chunk_records, n_local - first))
            words = torch.empty((max(1, 8 * n), self.W), dtype=torch.int64, device=self.device)[:8 * n]
            flips = torch.empty(max(1, 8 * n), dtype=torch.uint8, device=self.device)[:8 * n]
            if n:
                self._lib.check(self._d.ldbg_shard_nbr_queries(self.shard._h, C.c_int64(first), C.c_int64(n), self._ptr(words), self._ptr(flips)))
            _, _, _, owner, lidx = self.find_packed_dev(words)
            if n:
                owner, lidx = owner.contiguous(), lidx.contiguous()
                self._lib.check(self._d.ldbg_shard_set_nbr(self.shard._h, C.c_int64(first), C.c_int64(n), self._ptr(owner), self._ptr(lidx), self._ptr(flips)))
        rb = C.c_int()
        self._lib.check(self._d.ldbg_shard_row_bytes(self.shard._h, C.byref(rb)))
        self.row_bytes = rb.value
        self.has_neighbour_index = True

    def fetch_rows(self, req_owner, req_lidx):
        """one exchange: req_owner int32[n] (-1 = no request), req_lidx int64[n] -> (rows uint8[n, row_bytes], have uint8[n]).
        Collective: every rank calls it, possibly with no requests."""
        torch, dist = self._torch, self._dist
        n, world = req_owner.shape[0], self.world
        sel = torch.nonzero(req_owner >= 0).flatten()
        own = req_owner[sel].to(torch.int64)
        order = torch.argsort(own, stable=True)
        sel = sel[order]
        send = req_lidx[sel].contiguous()
        send_counts = torch.bincount(own, minlength=world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self._group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        m = int(sum(rc))
        recv = torch.empty(m, dtype=torch.int64, device=self.device)
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=self._group)
        served = torch.zeros((max(1, m), self.row_bytes), dtype=torch.uint8, device=self.device)[:m]
        if m:
            self._lib.check(self._d.ldbg_shard_rows(self.shard._h, self._ptr(recv), C.c_int64(m), self._ptr(served)))
        back = torch.empty((int(sel.shape[0]), self.row_bytes), dtype=torch.uint8, device=self.device)
        dist.all_to_all_single(back, served.contiguous(), output_split_sizes=sc, input_split_sizes=rc, group=self._group)
        rows = torch.zeros((max(1, n), self.row_bytes), dtype=torch.uint8, device=self.device)[:n]
        have = torch.zeros(max(1, n), dtype=torch.uint8, device=self.device)[:n]
        rows[sel] = back
        have[sel] = 1
        return rows, have

    def close(self):
        self.shard.close()


class ShardedTraversalEngine:
    """TraversalEngine.walk over a ShardedCortexGraph (ContigStopper, no link annotations, odd k): bulk-synchronous —
    every step of all walks in flight on all ranks is one exchange of row requests and one of rows."""

    def __init__(self, sgraph, traversal_colors, recruitment_colors=(), direction=0, op=0, max_branch_length=75000):
        from .traversal import ContigStopper, TraversalEngineFactory
        self.g = sgraph
        if not getattr(sgraph, "has_neighbour_index", False):
            sgraph.build_neighbour_index()
        f = (TraversalEngineFactory(lib=sgraph._lib).traversalColors(*traversal_colors).graph(sgraph.shard).stoppingRule(ContigStopper)
             .traversalDirection(direction).combinationOperator(op).maxBranchLength(max_branch_length))
        if recruitment_colors:
            f.recruitmentColors(*recruitment_colors)
        self.engine = f.make()
        self._op_and = op == 1
        self._first = list(traversal_colors)[0]
        self._max_len = max_branch_length
        self._w = C.c_void_p()
        sgraph._lib.check(sgraph._d.ldbg_bsp_create(self.engine._h, C.byref(self._w)))
        self.kmers_traversed = 0
        self.exchanges = 0

    def walk_batch(self, seeds):
        """contigs of this rank's seeds (list of str); collective: every rank calls it (possibly with no seeds)"""
        g, torch, dist = self.g, self.g._torch, self.g._dist
        lib, d, P = g._lib, g._d, g._ptr
        k = g.k
        seeds = list(seeds)
        n = len(seeds)
        ascii_ = np.frombuffer("".join(seeds).encode(), dtype=np.uint8).reshape(n, k) if n else np.zeros((0, k), dtype=np.uint8)
        q = torch.from_numpy(pack_kmers(ascii_, k).view(np.int64)).to(g.device)
        found, cov, _, owner, lidx, canon = g.find_packed_dev(q, return_canonical=True)
        flip = (canon != q).any(dim=1).to(torch.uint8).contiguous() if n else torch.zeros(0, dtype=torch.uint8, device=g.device)
        ns = 2 * n
        req_owner = torch.full((max(1, ns),), -1, dtype=torch.int32, device=g.device)[:ns]
        req_lidx = torch.full((max(1, ns),), -1, dtype=torch.int64, device=g.device)[:ns]
        owner, lidx = owner.contiguous(), lidx.contiguous()
        lib.check(d.ldbg_bsp_start(self._w, C.c_int64(n), P(owner), P(lidx), P(flip), P(req_owner), P(req_lidx)))
        while True:
            pending = (req_owner >= 0).sum().to(torch.int64).reshape(1)
            dist.all_reduce(pending, group=g._group)
            if int(pending.item()) == 0:
                break
            rows, have = g.fetch_rows(req_owner, req_lidx)
            self.exchanges += 1
            lib.check(d.ldbg_bsp_step(self._w, P(have), P(rows.contiguous()), P(req_owner), P(req_lidx)))
        stride = self._max_len + 4
        strand_n = np.zeros(max(1, ns), dtype=np.uint32)
        status = np.zeros(max(1, ns), dtype=np.uint32)
        iters = np.zeros(max(1, ns), dtype=np.uint32)
        bases = np.zeros((max(1, ns), stride), dtype=np.uint8)
        if ns:
            lib.check(d.ldbg_bsp_results(self._w, strand_n.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
                                         iters.ctypes.data_as(C.c_void_p), bases.ctypes.data_as(C.c_void_p), C.c_int64(stride)))
        self.kmers_traversed = int(iters[:ns].sum())
        for s_ in range(ns):
            if status[s_] == 1:
                raise _native.JavaNullPointerException("getNextVertices: record missing while recruitment colours are set (seed %d)" % (s_ // 2))
            if status[s_] == 8:
                raise _native.LdbgError(7, "a walk outgrew its visited table / path buffer")
            if status[s_] == 11:
                raise _native.LdbgError(4, "a walk met a quirk-Q6 vertex: not supported over a sharded table")
        found_h, cov_h = found.cpu().numpy(), cov.cpu().numpy()
        alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
        out = []
        for i in range(n):
            nr, nf = int(strand_n[2 * i]), int(strand_n[2 * i + 1])
            null_r, null_f = status[2 * i] == 3, status[2 * i + 1] == 3
            is_null = (null_r or null_f) if self._op_and else (null_r and null_f)
            seed_ok = bool(found_h[i]) and int(cov_h[i][self._first]) > 0          # toWalk's seed test, TraversalUtils.java:392-397
            if is_null or nr + nf == 0 or not seed_ok:
                out.append("")
                continue
            rev = alpha[bases[2 * i][1:nr]][::-1].tobytes().decode() if nr > 1 else ""
            fwd = alpha[bases[2 * i + 1][1:nf]].tobytes().decode() if nf > 1 else ""
            out.append(rev + seeds[i] + fwd)
        return out

    def close(self):
        if self._w:
            self.g._d.ldbg_bsp_destroy(self._w)
            self._w = None
