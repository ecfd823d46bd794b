"""Finds k-mers whose java.util.Arrays.hashCode equals that of their reverse complement (the Q6 quirk of
CanonicalKmer.isFlipped) by meet-in-the-middle over the per-position terms of
h(s) - h(rc(s)) mod 2^32, and writes them to tests/golden/hash_collisions.txt.  Pure arithmetic; needs nothing
from the reference."""
import sys

import numpy as np

CODE = np.array([65, 67, 71, 84], dtype=np.uint64)
M = np.uint64(0xFFFFFFFF)


def pow31(n):
    return pow(31, n, 1 << 32)


def terms(k):
    """t[j][b] = contribution of base b at position j to h(s) - h(rc(s)) (mod 2^32)"""
    t = np.zeros((k, 4), dtype=np.uint64)
    for j in range(k):
        for b in range(4):
            t[j, b] = (int(CODE[b]) * pow31(k - 1 - j) - int(CODE[3 - b]) * pow31(j)) % (1 << 32)
    return t


def half_sums(t, positions):
    s = np.zeros(1, dtype=np.uint64)
    for j in positions:
        s = ((s[:, None] + t[j][None, :]) & M).reshape(-1)      # base of position j is the fastest digit so far
    return s


def find(k, rng, want=2, mitm=18):
    t = terms(k)
    out = []
    while len(out) < want:
        fixed = rng.integers(0, 4, size=k - mitm)
        base = int(sum(int(t[j, fixed[j]]) for j in range(k - mitm)) % (1 << 32))
        pa, pb = list(range(k - mitm, k - mitm // 2)), list(range(k - mitm // 2, k))
        sa, sb = half_sums(t, pa), half_sums(t, pb)
        need = (np.uint64(1 << 32) - ((sa + np.uint64(base)) & M)) & M
        order = np.argsort(sb)
        pos = np.searchsorted(sb[order], need)
        pos[pos >= len(sb)] = 0
        hit = np.nonzero(sb[order][pos] == need)[0]
        for ia in hit[:want]:
            ib = int(order[pos[ia]])

            def digits(v, n):   # the LAST position of the half is the fastest digit
                d = []
                for _ in range(n):
                    d.append(v & 3)
                    v >>= 2
                return d[::-1]
            s = list(fixed) + digits(int(ia), len(pa)) + digits(ib, len(pb))
            kmer = "".join("ACGT"[b] for b in s)
            out.append(kmer)
    return out[:want]


def jhash(s):
    h = 1
    for c in s.encode():
        h = (31 * h + c) & 0xFFFFFFFF
    return h


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


if __name__ == "__main__":
    rng = np.random.default_rng(12345)
    lines = []
    for k in (21, 31, 47, 63):
        for kmer in find(k, rng):
            assert jhash(kmer) == jhash(rc(kmer)) and kmer != rc(kmer), kmer
            lines.append(kmer)
            print(k, kmer, jhash(kmer))
    out = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/hash_collisions.txt"
    open(out, "w").write("\n".join(lines) + "\n")
