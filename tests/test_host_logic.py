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
