#!/usr/bin/env python3
"""Where do the product and the oracle part ways?  (the reference's debug flag logs branch / junction / fail events,
TraversalEngine.java:435-475; a dfs graph's vertices and edges IN INSERTION ORDER are that log in another form: every branch appends its
vertices as it walks them, a junction's children are merged in the order they returned, a failed child leaves nothing.)

usage: trace_diff.py graph.ctx [links.ctp.gz] --stopper NAME --source KMER [--sink KMER ...] [--direction 0|1|2] [--max-len N] [--hostsim]
Prints the first vertex (and edge) at which the two graphs differ, with the ten entries before it: the event that diverged is the one
that appended that entry — a junction whose children came back in another order, a branch that was cut elsewhere, a child that failed
on one side only.  For walks (ContigStopper) the same comparison runs over the contig and the vertex list."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return None if len(a) == len(b) else min(len(a), len(b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("graph")
    ap.add_argument("links", nargs="?")
    ap.add_argument("--stopper", default="ContigStopper")
    ap.add_argument("--source", required=True)
    ap.add_argument("--sink", action="append", default=[])
    ap.add_argument("--direction", type=int, default=0)
    ap.add_argument("--max-len", type=int, default=75000)
    ap.add_argument("--color", type=int, default=0)
    ap.add_argument("--hostsim", action="store_true", help="the product's kernels through the CPU host simulation (tests/hostsim) instead of the GPU")
    a = ap.parse_args()
    import corticall_amd as ca
    from oracle import pyoracle as orc
    lib = None
    if a.hostsim:
        from tests import hostsim
        lib = hostsim.load()
    g = ca.CortexGraph(a.graph, lib=lib)
    og = orc.Graph(a.graph, tuned=True)
    f = (ca.TraversalEngineFactory(lib=lib).traversalColors(a.color).traversalDirection(a.direction).stoppingRule(a.stopper).maxBranchLength(a.max_len).graph(g))
    olinks = []
    if a.links:
        f.links(ca.CortexLinks(a.links, g, lib=lib))
        olinks = [orc.Links(a.links)]
    e = f.make()
    oe = orc.Engine(og, [a.color], links=olinks, stopper=a.stopper, direction=a.direction, max_length=a.max_len)
    r = oe.dfs(a.source, a.sink)
    gi = e.dfs(a.source, *a.sink)
    if (gi is None) != r.is_null:
        print("DIVERGES: product returns %s, oracle returns %s" % ("null" if gi is None else "a graph", "null" if r.is_null else "a graph"))
        return 1
    if gi is None:
        print("identical: both return null")
        return 0
    pv, ov, pe, oe_ = gi.vertex_tuples(), r.vertices(), gi.edge_tuples(), r.edges()
    rc = 0
    for what, p, o in (("vertex", pv, ov), ("edge", pe, oe_)):
        d = first_diff(p, o)
        if d is None:
            print("%s list identical (%d entries, insertion order)" % (what, len(p)))
            continue
        rc = 1
        print("DIVERGES at %s %d of %d (product) / %d (oracle):" % (what, d, len(p), len(o)))
        for i in range(max(0, d - 10), d):
            print("   %6d  %s" % (i, p[i]))
        print(" > product %s" % (p[d],) if d < len(p) else " > product: <end>")
        print(" > oracle  %s" % (o[d],) if d < len(o) else " > oracle:  <end>")
    print("k-mers traversed: product %d, oracle %d" % (e.dfs_kmers_traversed, oe.kmers_traversed()))
    return rc


if __name__ == "__main__":
    sys.exit(main())
