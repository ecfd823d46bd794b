"""The graph utilities around the dfs path (J/utils/traversal/TraversalUtils.java): toGraph (:330-352), fillGaps (:235-315),
toWalk (:387-488), toContig (:367-381), over the part of JGraphT's DirectedWeightedPseudograph they rely on.

The searches themselves — one DestinationStopper dfs per gap source — run on the device in ONE batch per colour and direction
(TraversalEngine.dfs_batch); what is replayed on the host is the container bookkeeping: vertex and edge sets in insertion order,
edges refused when an equal CortexEdge is present (CortexEdge.java:42-57: same two vertices in either direction, same colour,
same weight), Graphs.addGraph, and the order in which a java.util.HashSet<String> hands out the gap sources."""
import numpy as np

from .partition import java_hashmap_order
from .traversal import FORWARD, OR, REVERSE, CortexVertex, DestinationStopper, TraversalEngineFactory

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _revcomp(s):
    return s.encode().translate(_COMP)[::-1].decode()


class CortexEdge:
    """J/utils/traversal/CortexEdge.java: identity = ({source, target} as a set, colour, weight)"""

    def __init__(self, s, t, color, weight=1.0):
        self.vertices, self.color, self.weight = frozenset((s, t)), color, weight

    def getColor(self): return self.color
    def getWeight(self): return self.weight
    def __eq__(self, o): return isinstance(o, CortexEdge) and (self.color, self.weight, self.vertices) == (o.color, o.weight, o.vertices)
    def __hash__(self): return hash((self.vertices, self.color, self.weight))


class Pseudograph:
    """org.jgrapht.graph.DirectedWeightedPseudograph<CortexVertex, CortexEdge> (1.0.1) as the reference uses it: LinkedHashSet-like
    vertex and edge sets, per-vertex incoming / outgoing edge lists in insertion order, addEdge refusing an edge that equals one
    already present (AbstractBaseGraph.addEdge(V, V, E): containsEdge(e))."""

    def __init__(self):
        self._v = {}            # vertex -> ([outgoing edges], [incoming edges])
        self._e = {}            # edge -> (source, target)

    def vertexSet(self): return list(self._v)
    def edgeSet(self): return list(self._e)
    def containsVertex(self, v): return v in self._v
    def getEdgeSource(self, e): return self._e[e][0]
    def getEdgeTarget(self, e): return self._e[e][1]
    def outgoingEdgesOf(self, v): return list(self._v[v][0])
    def incomingEdgesOf(self, v): return list(self._v[v][1])

    def addVertex(self, v):
        if v in self._v:
            return False
        self._v[v] = ([], [])
        return True

    def addEdge(self, s, t, e):
        if e in self._e:
            return False
        if s not in self._v or t not in self._v:
            raise ValueError("no such vertex in graph")
        self._e[e] = (s, t)
        self._v[s][0].append(e)
        self._v[t][1].append(e)
        return True

    def addGraph(self, src, rename=None):
        """Graphs.addGraph(this, src): all vertices, then all edges (each with its end points).  rename: vertex of src -> the vertex of this
        graph that stands for it (fillGaps with relabel=True)"""
        rn = (lambda v: rename.get(v, v)) if rename else (lambda v: v)
        for v in src.vertexSet():
            self.addVertex(rn(v))
        for e in src.edgeSet():
            s, t = rn(src.getEdgeSource(e)), rn(src.getEdgeTarget(e))
            self.addVertex(s)
            self.addVertex(t)
            self.addEdge(s, t, e if not rename else CortexEdge(s, t, e.getColor(), e.getWeight()))

    @staticmethod
    def fromDfsGraph(dg):
        """the graph a device dfs returned (traversal.DfsGraph), with its vertices and edges in their insertion order"""
        g = Pseudograph()
        vs = dg.vertexSet()
        for v in vs:
            g.addVertex(v)
        for s, t, c in dg.edge_tuples():
            g.addEdge(vs[s], vs[t], CortexEdge(vs[s], vs[t], c, 1.0))
        return g


def java_string_hash(s):
    h = 0
    for ch in s.encode():
        h = (31 * h + ch) & 0xFFFFFFFF
    return h


def java_string_set_order(strings):
    """iteration order of a HashSet<String> the strings were added to in this order (duplicates ignored)"""
    uniq = list(dict.fromkeys(strings))
    if not uniq:
        return []
    order = java_hashmap_order(np.array([java_string_hash(s) for s in uniq], dtype=np.uint32))
    return [uniq[i] for i in order]


def _java_small_sort(items, cmp):
    """java.util.TimSort on fewer than 32 elements (countRunAndMakeAscending + binarySort): toWalk's comparators never return 0
    (:424-428, 466-470), so the outcome depends on the algorithm"""
    v = list(items)
    n = len(v)
    if n < 2:
        return v
    assert n < 32
    hi = 1
    if cmp(v[1], v[0]) < 0:
        hi = 2
        while hi < n and cmp(v[hi], v[hi - 1]) < 0:
            hi += 1
        v[:hi] = v[:hi][::-1]
    else:
        hi = 2
        while hi < n and cmp(v[hi], v[hi - 1]) >= 0:
            hi += 1
    for start in range(hi, n):
        pivot = v[start]
        lo, r = 0, start
        while lo < r:
            mid = (lo + r) >> 1
            if cmp(pivot, v[mid]) < 0:
                r = mid
            else:
                lo = mid + 1
        v[lo + 1:start + 1] = v[lo:start]
        v[lo] = pivot
    return v


def toGraph(walk, colors):
    """:330-352 — a walk as a graph: consecutive vertices joined in every colour both of them have coverage in"""
    g = Pseudograph()
    pv = walk[0]
    g.addVertex(pv)
    for nv in walk[1:]:
        g.addVertex(nv)
        for c in colors:
            if pv.getCortexRecord().getCoverage(c) > 0 and nv.getCortexRecord().getCoverage(c) > 0:
                g.addEdge(pv, nv, CortexEdge(pv, nv, c, 1.0))
        pv = nv
    return g


def _next_kmers(v, c):
    """getAllNextKmers(record, flipped).get(c) as strings (:536-560), flipped by comparison with the canonical k-mer as fillGaps does"""
    cr, sk = v.getCortexRecord(), v.getKmerAsString()
    if cr is None:
        return None
    flipped = sk != cr.getKmerAsString()
    bases = cr.getOutEdgesAsStrings(c) if not flipped else cr.getInEdgesAsStrings(c, True)
    return {sk[1:] + b for b in bases}


def _prev_kmers(v, c):
    cr, sk = v.getCortexRecord(), v.getKmerAsString()
    if cr is None:
        return None
    flipped = sk != cr.getKmerAsString()
    bases = cr.getInEdgesAsStrings(c) if not flipped else cr.getOutEdgesAsStrings(c, True)
    return {b + sk[:-1] for b in bases}


def _dfs_collection(engine, sources, sinks):
    """TraversalEngine.dfs(Collection<String> sources, Collection<String> sinks) :41-58: one dfs per source, in the collection's
    order, the graphs merged with Graphs.addGraph — all searches in one device batch"""
    if not sources:
        return None
    batch = engine.dfs_batch(sources, [list(sinks)] * len(sources))
    out = None
    for dg in batch:
        if dg is None:
            continue
        g = dg if isinstance(dg, Pseudograph) else Pseudograph.fromDfsGraph(dg)
        if out is None:
            out = g
        else:
            out.addGraph(g)
    return out


def fillGaps(g, graph, links, colors, relabel=True, engine_factory=None):
    """:235-315 — per colour: edges the colour has between joined vertices; vertices with an edge in the colour that leaves the
    graph are sources (outgoing) / sinks (incoming); DestinationStopper searches of at most 1000 vertices from every source towards
    the sinks (forward; if that returns nothing, backwards from the sinks) are merged in.

    relabel=True (the default) is pinned by the reference's own test of this function (TraversalUtilsTest.java:19-97, SURVEY V13, five
    strings): a vertex of a search result that differs from a vertex already in the graph by `index` alone is taken to be that vertex,
    so a filled stretch is joined to g at both ends.  relabel=False is the literal reading of the sources: Graphs.addGraph joins
    vertices that are equal, and CortexVertex.equals includes `index` — the source a search starts from comes back with index 0 while
    the same k-mer in g (a walk: index -1 / +1 either side of its seed) does not, so the stretch hangs on g at its far end only and the
    test's strings do not come out.

    engine_factory(colour, direction) -> an object with dfs_batch(sources, sinks_per_source) (DfsGraph / Pseudograph / None per source) and
    close(): where the searches run.  Default: the device engine over `graph` (the golden tests plug in their own checker)."""
    filled = Pseudograph()
    filled.addGraph(g)
    colors = list(colors)
    for e in g.edgeSet():
        v0, v1 = g.getEdgeSource(e), g.getEdgeTarget(e)
        for c in colors:
            nks = _next_kmers(v0, c)
            if nks is None:
                raise _npe("fillGaps: a vertex without a record")
            if v1.getKmerAsString() in nks:
                filled.addEdge(v0, v1, CortexEdge(v0, v1, c, 1.0))
    for c in colors:
        sources, sinks = [], []
        for v in filled.vertexSet():
            vs = {filled.getEdgeTarget(e).getKmerAsString() for e in filled.outgoingEdgesOf(v) if e.getColor() == c}
            nk = _next_kmers(v, c)
            if nk is None:
                raise _npe("fillGaps: a vertex without a record")
            if nk - vs:
                sources.append(v.getKmerAsString())
            vp = {filled.getEdgeSource(e).getKmerAsString() for e in filled.incomingEdgesOf(v) if e.getColor() == c}
            if _prev_kmers(v, c) - vp:
                sinks.append(v.getKmerAsString())
        sources, sinks = java_string_set_order(sources), java_string_set_order(sinks)

        def engine(direction):
            if engine_factory is not None:
                return engine_factory(c, direction)
            f = (TraversalEngineFactory(lib=graph._lib).traversalColors(c).traversalDirection(direction).combinationOperator(OR)
                 .stoppingRule(DestinationStopper).maxBranchLength(1000).graph(graph))
            if links:
                f.links(*links)
            return f.make()

        ef = engine(FORWARD)
        fill = _dfs_collection(ef, sources, sinks)
        ef.close()
        if fill is None:
            er = engine(REVERSE)
            fill = _dfs_collection(er, sinks, sources)
            er.close()
        if fill is not None:
            rename = None
            if relabel:
                have = {}
                for v in filled.vertexSet():
                    have.setdefault((v.getKmerAsString(), v.getCortexRecord(), v.getCopyIndex()), v)
                rename = {v: have[(v.getKmerAsString(), v.getCortexRecord(), v.getCopyIndex())] for v in fill.vertexSet()
                          if (v.getKmerAsString(), v.getCortexRecord(), v.getCopyIndex()) in have}
            filled.addGraph(fill, rename)
    return filled


def _npe(msg):
    from . import _native
    return _native.JavaNullPointerException(msg)


def toWalk(g, sk, color):
    """:387-488 — the linear walk through g in one colour that passes the vertex with k-mer sk"""
    w = []
    if g is None:
        return w
    seed = None
    for v in g.vertexSet():
        cr = v.getCortexRecord()
        if v.getKmerAsString() == sk and cr is not None and cr.getCoverage(color) > 0 and (seed is None or v.getCopyIndex() < seed.getCopyIndex()):
            seed = v
    if seed is None:
        return w
    w.append(seed)
    for forward in (True, False):
        seen, cv = set(), seed
        while cv is not None and cv not in seen:
            if forward:
                nvs = [g.getEdgeTarget(e) for e in g.outgoingEdgesOf(cv) if e.getColor() == color]
            else:
                nvs = [g.getEdgeSource(e) for e in g.incomingEdgesOf(cv) if e.getColor() == color]
            if cv in nvs:
                nvs.remove(cv)                                       # List.remove(Object): the first occurrence
            nv = None
            if len(nvs) == 1:
                nv = nvs[0]
            elif len(nvs) > 1:
                if any(x.getCortexRecord() is None for x in nvs):
                    raise _npe("toWalk: a vertex without a record among the candidates")
                if all(nvs[0].getCanonicalKmer() == x.getCanonicalKmer() for x in nvs[1:]):
                    if forward:
                        nvs = _java_small_sort(nvs, lambda a, b: -1 if a.getCopyIndex() < b.getCopyIndex() else 1)
                    else:
                        nvs = _java_small_sort(nvs, lambda a, b: -1 if a.getCopyIndex() > b.getCopyIndex() else 1)
                    nv = nvs[0]
            if nv is not None:
                if forward:
                    w.append(nv)
                else:
                    w.insert(0, nv)
                seen.add(cv)
            cv = nv
    return w


def toContig(walk):
    """:367-381"""
    s = ""
    for v in walk:
        sk = v.getKmerAsString()
        s = sk if not s else s + sk[-1]
    return s


# ---------------------------------------------------------------------------------------------------------------- PathFinder
class GraphPath:
    """org.jgrapht.GraphPath as PathFinder's callers read it: getVertexList(), getEdgeList(), getWeight()"""

    def __init__(self, vertices, edges, weight):
        self._v, self._e, self._w = list(vertices), list(edges), weight

    def getVertexList(self): return list(self._v)
    def getEdgeList(self): return list(self._e)
    def getStartVertex(self): return self._v[0]
    def getEndVertex(self): return self._v[-1]
    def getWeight(self): return self._w


class PathFinder:
    """J/utils/traversal/PathFinder.java:18-83: the (at most 10) shortest simple paths between two vertices over the edges of ONE colour
    of a dfs / fillGaps graph.

    The reference delegates to org.jgrapht.alg.shortestpath.KShortestPaths (jgrapht-core 1.0.1, ivy.xml:17 — the jar is not part of
    /root/reference).  Restated here from that release's published algorithm: a Bellman-Ford iteration in which every vertex keeps a
    RANKING LIST of its k best simple paths from the start (KShortestPathsIterator); pass p extends the lists of the vertices improved in
    pass p - 1 along their outgoing edges; a candidate is dropped when it revisits one of its own vertices, or when the end vertex can no
    longer be reached from its last vertex without touching the path (the guard test, on the UNDIRECTED connectivity of the graph minus
    the path, as ConnectivityInspector over the MaskSubgraph does); a list is kept sorted by weight, a candidate of a weight already in
    the list goes right behind the first element of that weight, and the list never grows beyond k; at most |V| - 1 passes.  Edge weights
    are CortexEdge.getWeight() (AbstractBaseGraph.getEdgeWeight for a DefaultWeightedEdge).
    What this restatement does NOT pin: within one pass JGraphT visits the improved vertices in the iteration order of a
    java.util.HashSet<CortexVertex>; here they are visited in the order they were improved.  That can only permute paths of EQUAL weight
    (the reference's own test, TraversalEngineTest.java:160-208, accepts either order) — "parity unpinned" for that tie order."""

    K = 10

    def __init__(self, graph, color):
        self._out = {}             # vertex -> [(edge, target)] in insertion order (DefaultDirectedGraph: no multi-edges needed beyond equality)
        self._edges = {}
        for e in graph.edgeSet():
            if e.getColor() != color:
                continue
            s_, t_ = graph.getEdgeSource(e), graph.getEdgeTarget(e)
            self._out.setdefault(s_, [])
            self._out.setdefault(t_, [])
            if e in self._edges:       # Graph.addEdge(s, t, e) refuses an edge that is already there
                continue
            # a DefaultDirectedGraph holds no second edge from s to t either (AbstractBaseGraph.addEdge: !allowingMultipleEdges && containsEdge(s, t))
            if any(t2 == t_ for _, t2 in self._out[s_]):
                continue
            self._edges[e] = (s_, t_)
            self._out[s_].append((e, t_))

    def vertexSet(self): return list(self._out)

    def getPath(self, start, end, constraint=None, accept=True):
        ps = self.getPaths(start, end, constraint, accept)
        return ps[0] if ps else None

    def getPaths(self, start, end, constraint=None, accept=True):
        """constraint: a canonical k-mer (str); accept=True keeps the paths that pass through it, accept=False those that do not (:48-82)"""
        if not self._out or start not in self._out or end not in self._out:
            return []
        paths = self._k_shortest(start, end)
        if constraint is None:
            return paths
        out = []
        for gp in paths:
            found = any(min(v.getKmerAsString(), _revcomp(v.getKmerAsString())) == constraint for v in gp.getVertexList())
            if found == bool(accept):
                out.append(gp)
        return out

    # ---- KShortestPaths(g, 10).getPaths(start, end)
    def _undirected(self):
        adj = {v: set() for v in self._out}
        for s_, outs in self._out.items():
            for _, t_ in outs:
                adj[s_].add(t_)
                adj[t_].add(s_)
        return adj

    def _guard_disconnected(self, adj, path_vertices, reached, guard):
        if reached == guard:
            return False
        masked = set(path_vertices)
        if guard in masked:
            return True
        seen, stack = {reached}, [reached]
        while stack:
            u = stack.pop()
            for w in adj[u]:
                if w in seen or w in masked:
                    continue
                if w == guard:
                    return False
                seen.add(w)
                stack.append(w)
        return True

    def _k_shortest(self, start, end):
        k = self.K
        adj = self._undirected()
        # a ranking element: (weight, vertices tuple, edges tuple)
        seen = {start: [(0.0, (start,), ())]}
        prev = {start: list(seen[start])}
        improved = [start]
        max_hops = len(self._out) - 1
        for _ in range(max_hops):
            if not improved:
                break
            now = []
            for u in improved:
                if u == end:
                    continue
                for e, t_ in self._out[u]:
                    if t_ == start:
                        continue
                    cands = []
                    for w, vs, es in prev.get(u, ()):
                        if t_ in vs or self._guard_disconnected(adj, vs, t_, end):
                            continue
                        cands.append((w + e.getWeight(), vs + (t_,), es + (e,)))
                    relaxed = False
                    if t_ not in seen:
                        lst = cands[:k]
                        if lst:
                            seen[t_] = lst
                            relaxed = True
                    else:
                        lst = seen[t_]
                        y = 0
                        for c in cands:
                            placed = False
                            while y < len(lst):
                                if c[0] < lst[y][0]:
                                    lst.insert(y, c)
                                    placed = True
                                elif c[0] == lst[y][0]:
                                    lst.insert(y + 1, c)
                                    placed = True
                                if placed:
                                    relaxed = True
                                    if len(lst) > k:
                                        del lst[k]
                                    break
                                y += 1
                            if not placed and c[0] > lst[-1][0]:
                                if len(lst) < k:
                                    lst.append(c)
                                    relaxed = True
                                else:
                                    break
                    if relaxed and t_ not in now:
                        now.append(t_)
            for v in now:
                prev[v] = list(seen[v])
            improved = now
        return [GraphPath(vs, es, w) for w, vs, es in seen.get(end, ())] if end != start else [GraphPath((start,), (), 0.0)]
