"""The wave-cooperative code on the CPU: the host simulation with 64 lanes per wavefront in LOCK STEP (csrc/rt.h: one fibre per lane, a
wavefront primitive is a barrier at which every live lane deposits its operand) and the device's 16 link-store elements per lane in
simulated LDS.  What the one-lane simulation of tests/test_hostsim_parity.py cannot reach runs here as it does on the device: lscoop.h
(whole-wavefront adds / choices, the 16-lane group forms, the gathered junction records), wave-cooperative table regrowth (strand.h),
path expansion and contigs from stored paths (walk.cpp: prefix sums over 64 stored entries), the request bucketing of the sharded
regime (image.cpp), k_dfs with 64 searches per wavefront.  Same cases, same oracle."""
import os

import pytest

from tests import parity_cases as pc

# The lock-step simulation costs a fibre switch per lane and primitive: the whole module (both wavefront widths, every case below) takes
# about 25 minutes on 8 cores.  CI runs the cases marked `quick` at 64 lanes (about 3 minutes); LDBG_WAVEFRONT_ALL=1 runs everything
# (done once per round: profiles/r03_hostsim_wavefront.log).
ALL = os.environ.get("LDBG_WAVEFRONT_ALL") == "1"
slow = pytest.mark.skipif(not ALL, reason="the full lock-step suite runs with LDBG_WAVEFRONT_ALL=1")


@pytest.fixture(scope="module", params=[64, 16] if ALL else [64])
def lib(request):
    from tests import hostsim
    return hostsim.load_wavefront(request.param)


@pytest.mark.parametrize("k,seed,links", [(9, 2, True), (47, 5, True)] + [pytest.param(*x, marks=slow) for x in [(31, 4, True), (33, 7, True), (21, 3, False), (64, 9, False)]])
def test_random_walks(orc, lib, tmp_path, k, seed, links): pc.case_random_walks(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("seed", [0] + [pytest.param(x, marks=slow) for x in (1, 2, 3)])
def test_dense_cycles(orc, lib, tmp_path, seed): pc.case_dense_cycles(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("seed", [0] + [pytest.param(x, marks=slow) for x in (1, 2, 3)])
def test_run_steps(orc, lib, tmp_path, seed): pc.case_run_steps(orc, lib, tmp_path, seed)


@slow
def test_big_link_stores(orc, lib, tmp_path): pc.case_big_link_stores(orc, lib, tmp_path)      # (6.5 minutes in lock step: hundreds of live links per walk)
@slow
def test_long_walks(orc, lib, tmp_path): pc.case_long_walks(orc, lib, tmp_path)
def test_ref_cycles_without_and_with_links(orc, lib, tmp_path): pc.test_ref_cycles_without_and_with_links(orc, lib, tmp_path)
def test_ref_link_guided_walk(orc, lib, tmp_path): pc.test_ref_link_guided_walk(orc, lib, tmp_path)
@slow
def test_hash_collision(orc, lib, tmp_path): pc.case_hash_collision(orc, lib, tmp_path)


@pytest.mark.parametrize("seed", [pytest.param(0, marks=slow), pytest.param(109, marks=slow)])
def test_dfs_run_steps(orc, lib, tmp_path, seed): pc.case_dfs_run_steps(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("seed", [0, pytest.param(1, marks=slow)])
def test_dfs_dense(orc, lib, tmp_path, seed): pc.case_dfs_dense(orc, lib, tmp_path, seed)


@slow
@pytest.mark.parametrize("k,seed,links", [(31, 4, True)])
def test_dfs_rules(orc, lib, tmp_path, k, seed, links): pc.case_dfs_rules(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("k,seed,links", [(31, 2, True)])
def test_partition(orc, lib, tmp_path, k, seed, links): pc.case_partition(orc, lib, tmp_path, k, seed, links)
