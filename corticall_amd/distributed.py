"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Two regimes:

* replicas — the table fits one GPU (configs C2-C4 are ~2 GB): every rank loads the graph, the seeds are
  partitioned, there is NO data-path collective; `partition()` / `gather_strings()` below are all it needs.
* hash-sharded table — `ShardedCortexGraph`: rank r keeps the records whose canonical k-mer hashes to r (still
  sorted, so the per-shard lookup is the unchanged HIP find kernel).  Loading: every rank scans 1/G of the file and the
  records travel to their owners in one all-to-all.  A batch of findRecord queries is routed with ONE exchange each
  way: canonicalise + owner on the device (ldbg_shard_owner_dev), bucket by owner, all-to-all the queries, local find
  (ldbg_graph_find_dev), all-to-all the answers back.
* traversals over the sharded table — `ShardedTraversalEngine`: every rank keeps a local IMAGE of the rows it has
  been sent (csrc/image.h) and runs the unchanged walk kernel on it; strands suspend where a row is missing.  One
  bulk-synchronous round = kernel launch -> requests bucketed by owner on the device -> all-to-all -> owners serve
  rows -> all-to-all -> insert, all queued on one stream without a host synchronisation; the host looks at the
  "anyone still walking" count only every few rounds.

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


def pack_kmers(ascii_kmers, k, return_valid=False):
    """ASCII u8[n,k] -> packed words u64[n,W] (word 0 most significant).  A string with a non-ACGT byte is not a k-mer (Q4): its
    words are zero and valid[i] is False — validity travels beside the words (at k = 32, 64, ... no bit pattern is free)"""
    a = np.ascontiguousarray(ascii_kmers, dtype=np.uint8).reshape(-1, k)
    W = (k + 31) // 32
    codes = _LUT[a]
    bad = (codes == 255).any(axis=1)
    codes = np.where(codes == 255, 0, codes).astype(np.uint64)
    words = np.zeros((a.shape[0], W), dtype=np.uint64)
    for i in range(k):
        bit = 2 * (k - 1 - i)
        words[:, W - 1 - (bit >> 6)] |= codes[:, i] << np.uint64(bit & 63)
    words[bad] = 0
    return (words, ~bad) if return_valid else words


class ShardedCortexGraph:
    """A .ctx table hash-partitioned over the ranks of a process group."""

    def __init__(self, path, device=0, lib=None, group=None, chunk_records=1 << 22):      # (chunk: 4 M records = 130 MB at k <= 64, 3 colours)
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._lib = lib or _native.default_lib()
        self._d = self._lib.dll
        self.path = str(path)
        self.device = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
        raw = np.memmap(self.path, dtype=np.uint8, mode="r")
        h = ctx_header(raw)
        self.k, self.W, self.C = h["k"], h["W"], h["C"]
        rec = 8 * self.W + 5 * self.C
        n_all = (raw.size - h["data_offset"]) // rec
        # Every rank scans ONE slice of the file (1/world of the records) and sends each record to its owner — on the DEVICE: a chunk of the
        # slice is uploaded, ldbg_shard_owner_dev names the owners, a stable sort by owner buckets the records (a slice is sorted and the sort
        # keeps the order), one all-to-all per chunk carries them over (a chunk is ~130 MB: a single collective of more than 2^31 bytes does
        # not arrive whole).  The file is sorted and the slices ascend with the rank, so what a shard receives — source rank after source
        # rank, chunk after chunk — IS its sorted table: nothing to merge, no copy back to the host; the library lays the records out where
        # they are (ldbg_graph_open_device).
        first, cnt = partition(n_all, self.rank, self.world)
        records = raw[h["data_offset"] + first * rec:h["data_offset"] + (first + cnt) * rec].reshape(cnt, rec)
        W8 = 8 * self.W
        nch = torch.tensor([-(-cnt // chunk_records)], dtype=torch.int64, device=self.device)
        dist.all_reduce(nch, op=dist.ReduceOp.MAX, group=group)
        pieces = [[] for _ in range(self.world)]
        for c in range(int(nch.item())):
            lo, hi = min(cnt, c * chunk_records), min(cnt, (c + 1) * chunk_records)
            m = hi - lo
            if m:
                blk = torch.from_numpy(np.array(records[lo:hi])).to(self.device)                    # [m, rec] u8
                kb = torch.empty((m, W8), dtype=torch.uint8, device=self.device)                    # (an explicit copy: rows of exactly W8 bytes, whatever m is)
                kb.copy_(blk[:, :W8])
                keys = kb.view(torch.int64)                                                         # [m, W]: the file's key words are the packed words
                canon = torch.empty_like(keys)
                owner = torch.empty(m, dtype=torch.int32, device=self.device)
                self._lib.check(self._d.ldbg_shard_owner_dev(self.k, self._ptr(keys), C.c_int64(m), self.world, self._ptr(canon), self._ptr(owner), None))
                owner = owner.to(torch.int64)
                send = blk[torch.argsort(owner, stable=True)].contiguous()
                counts = torch.bincount(owner, minlength=self.world)
                del blk, keys, canon, kb
            else:
                send = torch.zeros((0, rec), dtype=torch.uint8, device=self.device)
                counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
            rcounts = torch.empty_like(counts)
            dist.all_to_all_single(rcounts, counts, group=group)
            sc, rc = counts.tolist(), rcounts.tolist()
            got = torch.empty((int(sum(rc)), rec), dtype=torch.uint8, device=self.device)
            dist.all_to_all_single(got.view(-1), send.view(-1), output_split_sizes=[int(x) * rec for x in rc], input_split_sizes=[int(x) * rec for x in sc], group=group)
            at = 0
            for src in range(self.world):
                if rc[src]:
                    pieces[src].append(got[at:at + rc[src]])
                at += rc[src]
        parts = [p_ for src in range(self.world) for p_ in pieces[src]]
        shard = torch.cat(parts) if parts else torch.zeros((0, rec), dtype=torch.uint8, device=self.device)
        del pieces, parts
        if shard.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()
        self.shard = CortexGraph(self.path + "#shard%d" % self.rank, device=device, lib=self._lib,
                                 device_records=(bytes(raw[:h["data_offset"]]), shard.data_ptr(), int(shard.shape[0])))
        del shard
        self._lib.check(self._d.ldbg_graph_set_shard(self.shard._h, 1))
        n = torch.tensor([self.shard.getNumRecords()], dtype=torch.int64, device=self.device)
        dist.all_reduce(n, group=group)
        self.numRecords = int(n.item())
        self.has_neighbour_index = False

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
        words, valid = pack_kmers(kmers, self.k, return_valid=True)
        q = torch.from_numpy(words.view(np.int64)).to(self.device)
        found, l_cov, l_edges, owner, l_idx = self.find_packed_dev(q, valid=torch.from_numpy(valid).to(self.device))
        return (found.cpu().numpy(), l_cov.cpu().numpy(), l_edges.cpu().numpy(), owner.cpu().numpy(), l_idx.cpu().numpy())

    def find_packed_dev(self, q, return_canonical=False, valid=None):
        """device form: q = packed k-mer words, int64[n, W] on this rank's device; valid (bool[n], optional): False = not a k-mer,
        the query misses (Q4).  -> (found, cov, edges, owner, local_idx) as device tensors (+ the canonical words when asked);
        nothing touches the host but the split sizes"""
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
        if valid is not None and n:
            bad = ~valid.to(torch.bool)
            l_idx = torch.where(bad, torch.full_like(l_idx, -1), l_idx)
            l_cov = torch.where(bad[:, None], torch.zeros_like(l_cov), l_cov)
            l_edges = torch.where(bad[:, None], torch.zeros_like(l_edges), l_edges)
        found = l_idx >= 0
        if self.numRecords <= 2:       # Q1 is a property of the whole graph (CortexGraph.java:274-282)
            found = torch.zeros_like(found)
            l_idx = torch.full_like(l_idx, -1)
            l_cov, l_edges = torch.zeros_like(l_cov), torch.zeros_like(l_edges)
        if return_canonical:
            return found, l_cov, l_edges, owner, l_idx, canon
        return found, l_cov, l_edges, owner, l_idx

    # ---- the global neighbour index (csrc/shard.cpp)
    def build_neighbour_index(self, chunk_records=1 << 18):
        """global neighbour index of this shard: for every local record and each of its 8 possible neighbours the owner,
        the record number in the owner's shard and the orientation — one routed findRecord per edge, done once.
        Collective: every rank calls it."""
        torch, dist = self._torch, self._dist
        n_local = self.shard.getNumRecords()
        rounds = torch.tensor([(n_local + chunk_records - 1) // chunk_records], dtype=torch.int64, device=self.device)
        dist.all_reduce(rounds, op=dist.ReduceOp.MAX, group=self._group)
        saved_n = self.numRecords
        self.numRecords = max(self.numRecords, 3)       # (Q1 concerns findRecord's callers, not the index of what is stored)
        try:
            for r in range(int(rounds.item())):
                first = min(n_local, r * chunk_records)
                n = max(0, min(chunk_records, n_local - first))
                words = torch.zeros((max(1, 8 * n), self.W), dtype=torch.int64, device=self.device)[:8 * n]
                flips = torch.zeros(max(1, 8 * n), dtype=torch.uint8, device=self.device)[:8 * n]
                have = torch.zeros(max(1, 8 * n), dtype=torch.uint8, device=self.device)[:8 * n]
                if n:
                    self._lib.check(self._d.ldbg_shard_nbr_queries(self.shard._h, C.c_int64(first), C.c_int64(n), self._ptr(words), self._ptr(flips), self._ptr(have)))
                _, _, _, owner, lidx = self.find_packed_dev(words, valid=have.to(torch.bool))
                if n:
                    owner, lidx = owner.contiguous(), lidx.contiguous()
                    self._lib.check(self._d.ldbg_shard_set_nbr(self.shard._h, C.c_int64(first), C.c_int64(n), self._ptr(owner), self._ptr(lidx), self._ptr(flips)))
        finally:
            self.numRecords = saved_n
        self.has_neighbour_index = True

    def close(self):
        self.shard.close()


class _ImageFull(Exception):
    """some rank's image of the sharded table is full (seen by every rank in the same round)"""


class PeerRankFailed(RuntimeError):
    """another rank of the group failed in a collective batch; this rank's own part had no fatal error"""


class ShardedTraversalEngine:
    """TraversalEngine.walk over a ShardedCortexGraph — ContigStopper with or without link annotations, any k: every rank runs the
    walk kernel on its local image of the table (csrc/image.h); rows travel, walks do not.
    links: CortexLinks objects opened on `sgraph.shard` (each rank opens the link file against its own shard)."""

    def __init__(self, sgraph, traversal_colors, links=(), recruitment_colors=(), joining_colors=(), direction=0, op=0, max_branch_length=75000,
                 stopping_rule=None, image_rows=None, rows_per_owner=4096, check_every=8, keep_image=False, chain_depth=16, rois=None):
        self.g = sgraph
        if not sgraph.has_neighbour_index:
            sgraph.build_neighbour_index()
        self._lib, self._d = sgraph._lib, sgraph._d
        self._make = dict(traversal_colors=tuple(traversal_colors), links=tuple(links), recruitment_colors=tuple(recruitment_colors), joining_colors=tuple(joining_colors),
                          direction=direction, op=op, max_branch_length=max_branch_length, stopping_rule=stopping_rule, rois=rois)
        self._img, self.engine = None, None
        # sizing rule: an image never needs more rows than the table has records; it starts at min(records, 2^25) rows (5 GB at k <= 64) and
        # DOUBLES — on every rank together — whenever a batch fills it (_ImageFull below), so its size is never the caller's problem
        self._build(int(image_rows or min(max(sgraph.numRecords, 1024), 1 << 25)))
        self.rows_per_owner = int(rows_per_owner)
        self.chain_depth = max(1, int(chain_depth))      # row slots per request: the row asked for + rows around it its owner holds too
        self.check_every = int(check_every)
        self.keep_image = keep_image
        self.kmers_traversed = 0
        self.rounds = 0
        torch = sgraph._torch
        w, cap_o = sgraph.world, self.rows_per_owner
        self._send = torch.zeros((w, cap_o), dtype=torch.int64, device=sgraph.device)
        self._recv = torch.zeros((w, cap_o), dtype=torch.int64, device=sgraph.device)
        self._rows_out = torch.zeros((w, cap_o, self.chain_depth, self.row_bytes), dtype=torch.uint8, device=sgraph.device)
        self._rows_in = torch.zeros((w, cap_o, self.chain_depth, self.row_bytes), dtype=torch.uint8, device=sgraph.device)
        self._stats = torch.zeros(3, dtype=torch.int64, device=sgraph.device)
        self._tstream = torch.cuda.Stream(device=sgraph.device) if sgraph.device.type == "cuda" else None

    def _build(self, cap):
        """(re)create the image with `cap` rows and the engine over it"""
        from .traversal import ContigStopper, TraversalEngineFactory
        sgraph, lib, d, mk = self.g, self._lib, self._d, self._make
        if self.engine is not None:
            self.engine.close()
            self.engine = None
        if self._img is not None:
            d.ldbg_image_destroy(self._img)
            self._img = None
        self.image_rows = int(cap)
        self._img = C.c_void_p()
        lib.check(d.ldbg_image_create(sgraph.shard._h, C.c_int64(cap), C.c_int64(sgraph.numRecords), C.byref(self._img)))
        gh = C.c_void_p()
        lib.check(d.ldbg_image_graph(self._img, C.byref(gh)))
        self.image_graph = CortexGraph._from_handle(gh, lib, sgraph.path + "#image%d" % sgraph.rank, owner=self)
        rb = C.c_int()
        lib.check(d.ldbg_image_row_bytes(self._img, C.byref(rb)))
        self.row_bytes = rb.value
        f = (TraversalEngineFactory(lib=lib).traversalColors(*mk["traversal_colors"]).graph(self.image_graph).stoppingRule(mk["stopping_rule"] or ContigStopper)
             .traversalDirection(mk["direction"]).combinationOperator(mk["op"]).maxBranchLength(mk["max_branch_length"]))
        if mk["recruitment_colors"]:
            f.recruitmentColors(*mk["recruitment_colors"])
        if mk["joining_colors"]:
            f.joiningColors(*mk["joining_colors"])
        if mk["links"]:
            f.links(*mk["links"])
        if mk["rois"] is not None:          # the ROI graph of the stopping rules: a whole (small) graph, opened by every rank on its own device
            f.rois(mk["rois"])
        self.engine = f.make()

    def _grow_image(self):
        """every rank doubles its image (collective by construction: _ImageFull is raised on all ranks together)"""
        limit = max(self.g.numRecords, 1024)
        if self.image_rows >= limit:
            raise _native.LdbgError(7, "the image of the sharded table is full at %d rows although the table has only %d records" % (self.image_rows, self.g.numRecords))
        self.image_grown = getattr(self, "image_grown", 0) + 1
        self._build(min(limit, 2 * self.image_rows))

    def _stream(self):
        """the HIP stream every call of a round is queued on: the engine's own torch stream (a real stream handle — torch's default
        stream has handle 0, which the C ABI reads as "the library's stream", and nothing would order the kernels with the collectives)"""
        return C.c_void_p(self._tstream.cuda_stream) if self._tstream is not None else None

    def _exchange(self, stream, cap=None):
        """requests of this round -> rows in the image: bucket, all-to-all, serve, all-to-all, insert — all on one stream.
        cap: row requests per owner this round (every rank uses the same value; what does not fit is asked for again): the blocks that
        travel have a fixed size, so a round costs what its capacity costs, whatever is in it"""
        g, d, lib, dist = self.g, self._d, self._lib, self.g._dist
        P = lambda t: C.c_void_p(t.data_ptr())
        cap = self.rows_per_owner if cap is None else int(cap)
        n = g.world * cap
        send, recv = self._send.view(-1)[:n], self._recv.view(-1)[:n]
        rb = self.chain_depth * self.row_bytes
        rows_out, rows_in = self._rows_out.view(-1)[:n * rb], self._rows_in.view(-1)[:n * rb]
        lib.check(d.ldbg_image_bucket(self._img, g.world, C.c_uint32(cap), P(send), stream))
        dist.all_to_all_single(recv, send, group=g._group)
        lib.check(d.ldbg_image_serve_chain(self._img, g.rank, P(recv), C.c_int64(n), C.c_int(self.chain_depth), P(rows_out), stream))
        dist.all_to_all_single(rows_in, rows_out, group=g._group)
        lib.check(d.ldbg_image_insert(self._img, self.engine._h, P(rows_in), C.c_int64(n * self.chain_depth), stream))

    def _round_state(self):
        """(no rank has a strand left, row requests per owner to provide for in the coming rounds) from the counters of the round that has
        just run: one small all-reduce (MAX) of (strands in progress, requests filed)"""
        t = self._stats.clone()
        self.g._dist.all_reduce(t, op=self.g._dist.ReduceOp.MAX, group=self.g._group)
        left, nreq = int(t[0].item()), int(t[1].item())
        if int(t[2].item()):            # some rank's image is full: a strand that waits for a row it can never get would wait for ever
            raise _ImageFull()
        want = max(64, -(-3 * nreq // max(1, self.g.world)))       # three times an even spread of the busiest rank's requests
        cap = 64
        while cap < want:
            cap *= 2
        return left == 0, min(self.rows_per_owner, cap)

    # ---- a batch is a collective: every rank must leave it the same way.  Errors that depend on a rank's own seeds (a rule's
    # NullPointerException, a full link store, an exhausted pool ...) are raised where every rank stands at the same point of the
    # protocol — after the rounds — so they are CAUGHT there, the ranks agree on the worst outcome (one MAX all-reduce: 0 = done,
    # 1 = run the batch again — a store or the image has been enlarged —, 2 = failed) and then all return, all retry or all raise.
    _RETRYABLE = ("LINKSTORE_FULL", "LOG_FULL", "DEPTH_OVERFLOW")

    def _collective_batch(self, once, attempts=10):
        torch, dist = self.g._torch, self.g._dist
        for attempt in range(attempts):
            code, err, out, grow = 0, None, None, 0
            try:
                out = once()
            except _ImageFull:
                code, grow = 1, 1            # (raised on every rank together: the flag is all-reduced before anybody looks at it)
            except _native.LdbgError as ex:
                err = ex
                if "IMAGE_FULL" in str(ex):
                    code, grow = 1, 1
                else:
                    code = 1 if any(w in str(ex) for w in self._RETRYABLE) else 2     # the library has enlarged the store that was full
            except Exception as ex:       # noqa: BLE001 — whatever it is, the other ranks must not be left waiting in a collective
                code, err = 2, ex
            t = torch.tensor([code, grow], dtype=torch.int64, device=self.g.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.g._group)
            worst, grow = int(t[0].item()), int(t[1].item())
            if worst == 0:
                return out
            if worst == 2 or attempt == attempts - 1:
                if err is not None and (code == 2 or worst != 2):
                    raise err
                raise PeerRankFailed("a peer rank failed in this batch (this rank's own part was %s)" % ("fine" if err is None else "to be run again: %s" % err))
            if grow:
                self._grow_image()

    def walk_batch(self, seeds):
        """contigs of this rank's seeds (list of str); collective: every rank calls it (possibly with no seeds)"""
        return self._collective_batch(lambda: self._walk_batch_once(seeds))

    def _walk_batch_once(self, seeds):
        if self._tstream is None:
            return self._walk_batch_on_stream(seeds)
        self._tstream.wait_stream(self.g._torch.cuda.current_stream(self.g.device))
        with self.g._torch.cuda.stream(self._tstream):          # collectives issued here are ordered with this stream too
            out = self._walk_batch_on_stream(seeds)
        self._tstream.synchronize()
        return out

    def _resolve(self, kmers_ascii, stream):
        """routed findRecord of this rank's k-mers (ASCII u8[n,k]) and their rows into the image -> image slots int32[n] on the device
        (-1 = no record).  Collective."""
        g, torch = self.g, self.g._torch
        lib, d = self._lib, self._d
        P = lambda t: C.c_void_p(t.data_ptr())
        n = kmers_ascii.shape[0]
        words, valid = pack_kmers(kmers_ascii, g.k, return_valid=True)
        q = torch.from_numpy(words.view(np.int64)).to(g.device)
        found, _, _, owner, lidx = g.find_packed_dev(q, valid=torch.from_numpy(valid).to(g.device))
        keys = torch.where(found, (lidx + 1) | (owner.to(torch.int64) << 40), torch.zeros_like(lidx)).contiguous()
        slots = torch.full((max(1, n),), -1, dtype=torch.int32, device=g.device)[:n]
        while True:          # explicit requests, as many rounds as the per-owner blocks need
            missing = keys[(slots < 0) & (keys != 0)].contiguous() if n else keys
            ovf = C.c_int()
            lib.check(d.ldbg_image_counters(self._img, None, None, C.byref(ovf)))
            t = torch.tensor([int(missing.shape[0]), int(ovf.value)], dtype=torch.int64, device=g.device)
            g._dist.all_reduce(t, op=g._dist.ReduceOp.MAX, group=g._group)
            if int(t[1].item()):
                raise _ImageFull()
            if int(t[0].item()) == 0:
                break
            lib.check(d.ldbg_image_request(self._img, P(missing), C.c_int64(int(missing.shape[0])), stream))
            self._exchange(stream)
            lib.check(d.ldbg_image_reset_requests(self._img, stream))
            if n:
                lib.check(d.ldbg_image_lookup(self._img, P(keys), C.c_int64(n), P(slots), stream))
        return slots

    @staticmethod
    def _ascii(kmers, k):
        kmers = [x if isinstance(x, str) else bytes(x).decode() for x in kmers]
        n = len(kmers)
        return kmers, (np.frombuffer("".join(kmers).encode(), dtype=np.uint8).reshape(n, k) if n else np.zeros((0, k), dtype=np.uint8))

    def _walk_batch_on_stream(self, seeds):
        g, torch, dist = self.g, self.g._torch, self.g._dist
        lib, d = self._lib, self._d
        P = lambda t: C.c_void_p(t.data_ptr())
        seeds, ascii_ = self._ascii(seeds, g.k)
        n = len(seeds)
        stream = self._stream()
        if not self.keep_image:
            lib.check(d.ldbg_image_clear(self._img))
        slots = self._resolve(ascii_, stream)
        seed_buf = np.ascontiguousarray(ascii_).reshape(-1)
        lib.check(d.ldbg_engine_sharded_walk_begin(self.engine._h, self._img, seed_buf.ctypes.data_as(C.c_char_p), C.c_int64(n), P(slots), stream))
        rounds, gap, cap = 0, 1, self.rows_per_owner
        while True:
            for _ in range(gap):
                lib.check(d.ldbg_engine_sharded_walk_round(self.engine._h, P(self._stats)))
                self._exchange(stream, cap)
                rounds += 1
            done, cap = self._round_state()
            if done:
                break
            gap = min(self.check_every, gap * 2)
        self.rounds = rounds
        nrows, nreq, ovf = C.c_int64(), C.c_int64(), C.c_int()
        lib.check(d.ldbg_image_counters(self._img, C.byref(nrows), C.byref(nreq), C.byref(ovf)))
        self.image_rows_used = nrows.value
        total, trav = C.c_int64(), C.c_int64()
        lib.check(d.ldbg_engine_sharded_walk_finish(self.engine._h, C.byref(total), C.byref(trav)))
        self.kmers_traversed = trav.value
        arena = np.empty(max(1, total.value), dtype=np.uint8)
        offs = np.zeros(n + 1, dtype=np.int64)
        wl = np.zeros(max(1, n), dtype=np.int64)
        lib.check(d.ldbg_engine_walk_batch_fetch(self.engine._h, arena.ctypes.data_as(C.c_char_p), C.c_int64(total.value),
                                                 offs.ctypes.data_as(C.c_void_p), wl.ctypes.data_as(C.c_void_p)))
        raw = arena.tobytes()
        self.walk_lengths = wl[:n]
        return [raw[offs[i]:offs[i + 1]].decode() for i in range(n)]

    def dfs_batch(self, sources, sinks=None):
        """TraversalEngine.dfs(source, sinks...) for this rank's sources (the engine's stopping rule); sinks: per source a list of
        k-mers.  -> list of DfsGraph / None (a vertex's record index is its image slot: >= 0 means "has a record").  Collective."""
        return self._collective_batch(lambda: self._dfs_batch_once(sources, sinks))

    def _dfs_batch_once(self, sources, sinks):
        from .traversal import _DfsBatch
        g, torch = self.g, self.g._torch
        lib, d = self._lib, self._d
        P = lambda t: C.c_void_p(t.data_ptr())
        ctx = torch.cuda.stream(self._tstream) if self._tstream is not None else None
        if ctx is not None:
            self._tstream.wait_stream(torch.cuda.current_stream(g.device))
            ctx.__enter__()
        try:
            sources, src_ascii = self._ascii(sources, g.k)
            n = len(sources)
            sinks = sinks if sinks is not None else [[] for _ in sources]
            flat, sink_ascii = self._ascii([x for ss in sinks for x in ss], g.k)
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(ss) for ss in sinks])
            stream = self._stream()
            if not self.keep_image:
                lib.check(d.ldbg_image_clear(self._img))
            seed_slots = self._resolve(src_ascii, stream)
            sink_slots = self._resolve(sink_ascii, stream)
            state = {"rounds": 0, "gap": 1, "since": 0, "error": None, "cap": self.rows_per_owner}

            def round_done(_user):
                try:
                    self._exchange(stream, state["cap"])
                    state["rounds"] += 1
                    state["since"] += 1
                    if state["since"] < state["gap"]:
                        return 0
                    state["since"] = 0
                    state["gap"] = min(self.check_every, state["gap"] * 2)
                    done, state["cap"] = self._round_state()
                    return 1 if done else 0
                except _ImageFull as ex:           # (every rank gets here in the same round: the library gives the batch up, "IMAGE_FULL")
                    state["error"] = ex
                    return 2
                except BaseException as ex:        # never let an exception cross the C boundary
                    state["error"] = ex
                    return 2
            cb = C.CFUNCTYPE(C.c_int, C.c_void_p)(round_done)
            res = C.c_void_p()
            t0 = C.c_int64()
            eng = self.engine
            lib.check(d.ldbg_engine_dfs_kmers_traversed(eng._h, C.byref(t0)))
            src_buf = np.ascontiguousarray(src_ascii).reshape(-1)
            sink_buf = np.ascontiguousarray(sink_ascii).reshape(-1) if len(flat) else np.zeros(1, dtype=np.uint8)
            st = d.ldbg_engine_sharded_dfs_batch(eng._h, self._img, src_buf.ctypes.data_as(C.c_char_p), C.c_int64(n), sink_buf.ctypes.data_as(C.c_char_p),
                                                 off.ctypes.data_as(C.c_void_p), P(seed_slots), P(sink_slots), cb, None, P(self._stats), stream, C.byref(res))
            if state["error"] is not None:
                raise state["error"]
            lib.check(st)
            self.rounds = state["rounds"]
            t1 = C.c_int64()
            lib.check(d.ldbg_engine_dfs_kmers_traversed(eng._h, C.byref(t1)))
            self.dfs_kmers_traversed = t1.value - t0.value
            batch = _DfsBatch(eng, res)
            return [batch.graph(i) for i in range(n)]
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
                self._tstream.synchronize()

    def close(self):
        if getattr(self, "engine", None):
            self.engine.close()
            self.engine = None
        if getattr(self, "_img", None):
            self._d.ldbg_image_destroy(self._img)
            self._img = None
