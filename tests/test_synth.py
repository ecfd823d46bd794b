"""The synthetic-workload generator (tools/synth.cpp) writes bench inputs in the reference's file
formats.  Its graph must equal what TempGraphAssembler builds from the same sequences and its
closed-form links must equal what TempLinksAssembler derives from the tiled reads (both via the oracle)."""
import numpy as np
import pytest


@pytest.mark.parametrize("k,seed", [(21, 1), (31, 2), (47, 3)])
def test_synth_matches_fixture_assemblers(orc, tmp_path, k, seed):
    from tools import synth
    R, S = 90, 4
    prefix = str(tmp_path / "syn")
    st = synth.generate(prefix, 24000, k, n_chrom=3, n_indels=12, n_dnm=8, n_tandem=3, n_repeat_families=8,
                        repeat_copies=4, repeat_len=(k // 2, 3 * k), read_len=R, read_stride=S, n_seeds=300, seed=seed, threads=2)
    g = orc.Graph(prefix + ".ctx", tuned=True)
    assert (g.k, g.C, g.N) == (k, 3, st["n_records"])
    assert [g.sample_name(c) for c in range(3)] == ["child", "mom", "dad"]
    child = open(prefix + ".child.txt").read().split()
    # child colour of the graph == TempGraphAssembler over the child chromosomes
    ref_path = str(tmp_path / "child_only.ctx")
    orc.build_graph(ref_path, [("child", child)], k)
    rg = orc.Graph(ref_path, tuned=True)
    ref = {}
    for i in range(rg.N):
        f = rg.record_string(i).split()
        ref[f[0]] = (int(f[1]), f[2])
    seen = 0
    for i in range(g.N):
        f = g.record_string(i).split()
        if int(f[1]) > 0:
            assert ref[f[0]] == (int(f[1]), f[4]), f
            seen += 1
        else:
            assert f[4] == "........"
    assert seen == rg.N
    # links == TempLinksAssembler over the tiled reads
    reads = [c[a:a + R] for c in child for a in range(0, len(c) - R + 1, S)]
    lp = str(tmp_path / "ref.ctp.gz")
    orc.build_links(g, lp, "child", reads)
    exp = {kmer: sorted(js) for kmer, js in orc.Links(lp).records()}
    got = {kmer: sorted(js) for kmer, js in orc.Links(prefix + ".ctp.gz").records()}
    assert got == exp
    assert st["n_links"] == sum(len(v) for v in exp.values()) and st["n_link_kmers"] == len(exp)
    # seeds are child k-mers; the de novo ones are absent from both parents
    seeds = np.fromfile(prefix + ".seeds", dtype=np.uint8).reshape(-1, k)
    idx = g.find_batch(seeds, tuned=True)
    assert (idx >= 0).all() and len(seeds) == 300
    for i in range(st["n_novel_seeds"]):
        f = g.record_string(int(idx[i])).split()
        assert int(f[1]) > 0 and int(f[2]) == 0 and int(f[3]) == 0
