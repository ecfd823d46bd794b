"""host-side helpers of the product package that need neither a GPU nor the host simulation"""


def test_java_hashmap_order_matches_an_emulated_hashmap():
    """corticall_amd.partition.java_hashmap_order (final table size, bucket, insertion order) against a HashMap emulated put by put
    with its resizes (tests/parity_cases.py: JavaHashMap), at sizes around the resize thresholds"""
    import random
    import numpy as np
    from corticall_amd.partition import java_hashmap_order, java_bytes_hash
    from tests.parity_cases import JavaHashMap
    rng = random.Random(5)
    for n in (1, 2, 11, 12, 13, 24, 25, 48, 49, 96, 97, 200, 1000, 3073):
        hashes = [rng.getrandbits(32) if rng.random() < 0.8 else rng.choice([0, 1, 16, 17, 0x10000, 0x10001]) for _ in range(n)]
        m = JavaHashMap()
        for i, h in enumerate(hashes):
            m.put(i, h, False)
        assert list(java_hashmap_order(np.array(hashes, dtype=np.uint32))) == m.keys(), n
    # Arrays.hashCode(byte[]) of ASCII k-mers: "abc" -> 126145 (the value the reference's JVM gives)
    assert int(java_bytes_hash(np.frombuffer(b"abc", dtype=np.uint8).reshape(1, 3))[0]) == 126145


def test_engine_pool_deals_items_out_in_order():
    """EnginePool.map: items dealt out to the engines in turn, one thread per engine, results in item order, a worker's exception raised in
    the caller (no device needed: the engines are stand-ins)"""
    from corticall_amd.traversal import EnginePool

    class F:
        made = 0

        def make(self):
            F.made += 1
            return "engine%d" % F.made
    pool = EnginePool(F(), 3)
    assert pool.engines == ["engine1", "engine2", "engine3"]
    out = pool.map(lambda e, x: (e, x * x), range(8))
    assert [v for _, v in out] == [x * x for x in range(8)]
    assert [e for e, _ in out] == ["engine%d" % (1 + j % 3) for j in range(8)]
    assert pool.map(lambda e, x: x, []) == [] and pool.map(lambda e, x: (e, x), [5]) == [("engine1", 5)]

    def boom(e, x):
        if x == 4:
            raise ValueError("four")
        return x
    import pytest
    with pytest.raises(ValueError):
        pool.map(boom, range(6))
