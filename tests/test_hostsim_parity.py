"""CPU-only run of the parity cases against the TEST-ONLY host simulation of the kernels
(tests/hostsim): checks the kernel and host logic of libldbg without a GPU.  The real parity
gate is tests/test_gpu_parity.py (-m gpu), which runs the same cases through the HIP library."""
import pytest

from tests import parity_cases as pc


@pytest.fixture(scope="module")
def lib():
    from tests import hostsim
    return hostsim.load()


def test_fixture_graph(orc, lib, tmp_path): pc.case_fixture_graph(orc, lib, tmp_path)


@pytest.mark.parametrize("k,ncol", [(5, 1), (21, 2), (31, 3), (32, 1), (33, 2), (47, 3), (63, 3), (64, 1), (65, 2), (95, 1)])
def test_random_find(orc, lib, tmp_path, k, ncol): pc.case_random_find(orc, lib, tmp_path, k, ncol)


@pytest.mark.parametrize("k", [32, 64, 96])
def test_all_bits_kmers(orc, lib, tmp_path, k): pc.case_all_bits_kmers(orc, lib, tmp_path, k)


def test_q1_tiny(orc, lib, tmp_path): pc.case_q1_tiny(orc, lib, tmp_path)
def test_unsorted_rejected(orc, lib, tmp_path): pc.case_unsorted_rejected(orc, lib, tmp_path)
def test_record_count_guard(orc, lib, tmp_path, monkeypatch): pc.case_record_count_guard(orc, lib, tmp_path, monkeypatch)
def test_ref_short_contig_reconstruction(orc, lib, tmp_path): pc.test_ref_short_contig_reconstruction(orc, lib, tmp_path)
def test_ref_recruitment(orc, lib, tmp_path): pc.test_ref_recruitment(orc, lib, tmp_path)
def test_ref_cycles_without_and_with_links(orc, lib, tmp_path): pc.test_ref_cycles_without_and_with_links(orc, lib, tmp_path)
def test_ref_iterate_fwd_rev(orc, lib, tmp_path): pc.test_ref_iterate_fwd_rev(orc, lib, tmp_path)
def test_ref_go_forward_and_backward(orc, lib, tmp_path): pc.test_ref_go_forward_and_backward(orc, lib, tmp_path)
def test_ref_link_guided_walk(orc, lib, tmp_path): pc.test_ref_link_guided_walk(orc, lib, tmp_path)


@pytest.mark.parametrize("k,seed,links", [(9, 1, False), (9, 2, True), (21, 3, False), (31, 4, True), (47, 5, True), (63, 6, False), (33, 7, True), (32, 8, False), (64, 9, False), (65, 10, False)])
def test_random_walks(orc, lib, tmp_path, k, seed, links): pc.case_random_walks(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("seed", range(8))
def test_dense_cycles(orc, lib, tmp_path, seed): pc.case_dense_cycles(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("seed", range(6))
def test_run_steps(orc, lib, tmp_path, seed): pc.case_run_steps(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("seed", [0, 109, 231])      # (231: a branch that opens inside a stretch an ancestor crossed — found by tests/test_soak_hostsim.py)
def test_dfs_run_steps(orc, lib, tmp_path, seed): pc.case_dfs_run_steps(orc, lib, tmp_path, seed)


def test_long_walks(orc, lib, tmp_path): pc.case_long_walks(orc, lib, tmp_path)


def test_link_formats(orc, lib, tmp_path): pc.case_link_formats(orc, lib, tmp_path)


def test_sort(orc, lib, tmp_path): pc.case_sort(orc, lib, tmp_path)
def test_sort_rewrites_header(orc, lib, tmp_path): pc.case_sort_rewrites_header(orc, lib, tmp_path)
def test_join(orc, lib, tmp_path): pc.case_join(orc, lib, tmp_path)


def test_collection(orc, lib, tmp_path): pc.case_collection(orc, lib, tmp_path)


def test_ref_collection(orc, lib, tmp_path): pc.test_ref_collection(orc, lib, tmp_path)


def test_big_link_stores(orc, lib, tmp_path): pc.case_big_link_stores(orc, lib, tmp_path)


def test_hash_collision(orc, lib, tmp_path): pc.case_hash_collision(orc, lib, tmp_path)


@pytest.mark.parametrize("k,seed,links", [(9, 2, False), (21, 3, False), (31, 4, True), (47, 5, True)])
def test_dfs_rules(orc, lib, tmp_path, k, seed, links): pc.case_dfs_rules(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("seed", range(4))
def test_dfs_dense(orc, lib, tmp_path, seed): pc.case_dfs_dense(orc, lib, tmp_path, seed)


def test_ref_dfs_with_sinks(orc, lib, tmp_path): pc.test_ref_dfs_with_sinks(orc, lib, tmp_path)
def test_ref_multiple_traversal_colors(orc, lib, tmp_path): pc.test_ref_multiple_traversal_colors(orc, lib, tmp_path)


def test_dfs_packed_results(orc, lib, tmp_path): pc.case_dfs_packed_results(orc, lib, tmp_path)


def test_dfs_second_launch_without_the_index(orc, lib, tmp_path, monkeypatch):
    """a search the run steps hand back sends its chunk round again without the run index: forced here, the results must not change"""
    monkeypatch.setenv("LDBG_DFS_FORCE_RETRY", "1")
    pc.case_dfs_run_steps(orc, lib, tmp_path, 0)
    pc.case_dfs_dense(orc, lib, tmp_path, 1)


def test_factory_validation(orc, lib, tmp_path): pc.case_factory_validation(orc, lib, tmp_path)


def test_ref_fill_gaps(orc, lib, tmp_path): pc.test_ref_fill_gaps(orc, lib, tmp_path)


@pytest.mark.parametrize("seed", range(3))
def test_fill_gaps_random(orc, lib, tmp_path, seed): pc.case_fill_gaps_random(orc, lib, tmp_path, seed)


@pytest.mark.parametrize("k,seed,links", [(21, 1, False), (31, 2, True), (47, 3, True)])
def test_partition(orc, lib, tmp_path, k, seed, links): pc.case_partition(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("k,seed,links", [(21, 1, False), (31, 2, True)])
def test_findtips(orc, lib, tmp_path, k, seed, links): pc.case_findtips(orc, lib, tmp_path, k, seed, links)


@pytest.mark.parametrize("k,seed,links", [(21, 1, False), (31, 2, True), (47, 3, True), (32, 4, False)])
def test_facade(orc, lib, tmp_path, k, seed, links): pc.case_facade(orc, lib, tmp_path, k, seed, links)


def test_batch_splitting_on_a_small_device(orc, lib, tmp_path, monkeypatch):
    """3 MB of "device memory": the path / table pools run dry, the host splits the batch and re-runs — results stay exact"""
    monkeypatch.setenv("LDBG_HOSTSIM_MEM_MB", "3")
    pc.case_long_walks(orc, lib, tmp_path)
    pc.case_dfs_dense(orc, lib, tmp_path, 1)
    pc.case_partition(orc, lib, tmp_path, 31, 2, True)


def test_dfs_step_limit(orc, lib, tmp_path, monkeypatch): pc.case_dfs_step_limit(orc, lib, tmp_path, monkeypatch)


def test_close_in_any_order(orc, lib, tmp_path): pc.case_close_in_any_order(orc, lib, tmp_path)


def test_small_visited_tables(orc, lib, tmp_path, monkeypatch):
    """visited tables that start at 64 entries: probe rounds wrap round the end of a table and tables regrow many times"""
    monkeypatch.setenv("LDBG_VT_INITIAL", "64")
    pc.case_long_walks(orc, lib, tmp_path)
    pc.case_dense_cycles(orc, lib, tmp_path, 2)
    pc.case_random_walks(orc, lib, tmp_path, 31, 5, True, n=600)
    pc.case_dfs_dense(orc, lib, tmp_path, 3)
    pc.case_dfs_rules(orc, lib, tmp_path, 31, 2, True)


def test_lowercase_queries(orc, lib, tmp_path): pc.case_lowercase_queries(orc, lib, tmp_path)
