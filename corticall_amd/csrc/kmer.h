// 2-bit packed k-mer arithmetic shared by host and device code (product side).
// Representation ("packed words"): W = ceil(k/32) uint64, word 0 most significant, bases right
// aligned, A=0 C=1 G=2 T=3 — the value McCortex stores in a .ctx record.  Unsigned lexicographic
// order of the words equals the byte-wise ASCII order the reference sorts and searches by
// (CortexByteKmer.compareTo, J/utils/kmer/CortexByteKmer.java:41-49).
#pragma once
#include <stdint.h>
#include <algorithm>
#include <thread>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LDBG_HD __host__ __device__ __forceinline__
#else
#define LDBG_HD inline
#endif

namespace ldbg {

template <int W>
struct Kmer {
    uint64_t w[W];
};

template <int W>
LDBG_HD bool kmer_eq(const Kmer<W>& a, const Kmer<W>& b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < W; i++) e &= a.w[i] == b.w[i];
    return e;
}
// -1 / 0 / +1, unsigned lexicographic from word 0
template <int W>
LDBG_HD int kmer_cmp(const Kmer<W>& a, const Kmer<W>& b) {
#pragma unroll
    for (int i = 0; i < W; i++) {
        if (a.w[i] < b.w[i]) return -1;
        if (a.w[i] > b.w[i]) return 1;
    }
    return 0;
}

// reverse the order of the 32 two-bit groups of a word
LDBG_HD uint64_t rev2(uint64_t x) {
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
}

// SequenceUtils.reverseComplement on packed words (J/utils/sequence/SequenceUtils.java:127-135)
template <int W>
LDBG_HD Kmer<W> kmer_revcomp(const Kmer<W>& a, int k) {
    Kmer<W> r;
#pragma unroll
    for (int i = 0; i < W; i++) r.w[i] = rev2(~a.w[W - 1 - i]);
    const int s = 64 * W - 2 * k;   // right-align: shift the 64W-bit string right by s (0 <= s < 64)
    if (s > 0) {
#pragma unroll
        for (int i = W - 1; i > 0; i--) r.w[i] = (r.w[i] >> s) | (r.w[i - 1] << (64 - s));
        r.w[0] >>= s;
    }
    return r;
}

// alphanumericallyLowestOrientation (SequenceUtils.java:206-225): min(kmer, revcomp), ties -> input
template <int W>
LDBG_HD Kmer<W> kmer_canonical(const Kmer<W>& a, int k, bool* flipped_by_compare) {
    Kmer<W> rc = kmer_revcomp<W>(a, k);
    bool f = kmer_cmp<W>(rc, a) < 0;
    *flipped_by_compare = f;
    return f ? rc : a;
}

// word `idx` without a dynamically indexed array access (keeps Kmer<W> in registers on the GPU:
// a runtime subscript would force every struct holding a k-mer into scratch / LDS)
template <int W>
LDBG_HD uint64_t kmer_word(const Kmer<W>& a, int idx) {
    uint64_t v = a.w[0];
#pragma unroll
    for (int i = 1; i < W; i++) v = (idx == i) ? a.w[i] : v;
    return v;
}
template <int W>
LDBG_HD void kmer_or_word(Kmer<W>& a, int idx, uint64_t bits) {
#pragma unroll
    for (int i = 0; i < W; i++) a.w[i] |= (idx == i) ? bits : 0ULL;
}

// base i (0 = first / leftmost base) as 0..3
template <int W>
LDBG_HD unsigned kmer_base(const Kmer<W>& a, int k, int i) {
    int bit = 2 * (k - 1 - i);
    return (unsigned)((kmer_word<W>(a, W - 1 - (bit >> 6)) >> (bit & 63)) & 3ULL);
}

// successor: drop the first base, append b   (TraversalUtils.getAllNextKmers, sk[1:]+e)
template <int W>
LDBG_HD Kmer<W> kmer_next(const Kmer<W>& a, int k, unsigned b) {
    Kmer<W> r;
#pragma unroll
    for (int i = 0; i < W - 1; i++) r.w[i] = (a.w[i] << 2) | (a.w[i + 1] >> 62);
    r.w[W - 1] = (a.w[W - 1] << 2) | (uint64_t)b;
    const int top = 2 * k - 64 * (W - 1);   // bits used in word 0 (1..64)
    if (top < 64) r.w[0] &= ((1ULL << top) - 1ULL);
    return r;
}
// predecessor: prepend b, drop the last base   (getAllPrevKmers, e+sk[:-1])
template <int W>
LDBG_HD Kmer<W> kmer_prev(const Kmer<W>& a, int k, unsigned b) {
    Kmer<W> r;
#pragma unroll
    for (int i = W - 1; i > 0; i--) r.w[i] = (a.w[i] >> 2) | (a.w[i - 1] << 62);
    r.w[0] = a.w[0] >> 2;
    int bit = 2 * (k - 1);
    kmer_or_word<W>(r, W - 1 - (bit >> 6), (uint64_t)b << (bit & 63));
    return r;
}

// java.util.Arrays.hashCode(byte[]) of the ASCII form (CanonicalKmer.java:16,23,33 — quirk Q6)
template <int W>
LDBG_HD uint32_t kmer_java_hash(const Kmer<W>& a, int k) {
    uint32_t h = 1;
    for (int i = 0; i < k; i++) {
        unsigned b = kmer_base<W>(a, k, i);   // kmer_word: select chain, no scratch
        // 'A'=65 'C'=67 'G'=71 'T'=84
        uint32_t ch = b == 0 ? 65u : (b == 1 ? 67u : (b == 2 ? 71u : 84u));
        h = 31u * h + ch;
    }
    return h;
}

// The two Java hashes of a k-mer and of its reverse complement, modulo 32, without walking the bases:
// Arrays.hashCode(bytes) = 31^k + sum c_i 31^(k-1-i), and 31 == -1 (mod 32), so the hash mod 32 is an
// alternating sum of the character codes by distance from the end.  Used as a 31/32 quick reject before
// the full kmer_java_hash comparison of quirk Q6.
template <int W>
LDBG_HD void kmer_java_hash_mod32(const Kmer<W>& a, int k, uint32_t* h_self, uint32_t* h_rc) {
    // n[v][par]: occurrences of base v at even / odd distance from the END of the k-mer
    int n[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
    for (int i = 0; i < W; i++) {
        const int fields = (i == 0) ? (k - 32 * (W - 1)) : 32;          // valid 2-bit fields in this word
        const uint64_t valid = fields >= 32 ? 0x5555555555555555ULL : ((1ULL << (2 * fields)) - 1ULL) & 0x5555555555555555ULL;
        const uint64_t x = a.w[i];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const uint64_t pat = (uint64_t)v * 0x5555555555555555ULL;
            const uint64_t t = x ^ pat;
            const uint64_t z = ~(t | (t >> 1)) & valid;                  // 1 at the low bit of every field equal to v
            n[v][0] += __builtin_popcountll(z & 0x1111111111111111ULL); // even distance: fields 0,2,4,...
            n[v][1] += __builtin_popcountll(z & 0x4444444444444444ULL);
        }
    }
    const int code[4] = {65, 67, 71, 84};
    const int sign_k = (k & 1) ? -1 : 1;
    int hs = sign_k, hr = sign_k;
    const int q = (k - 1) & 1;      // a base at distance d from the end sits at distance k-1-d in the reverse complement
#pragma unroll
    for (int v = 0; v < 4; v++) {
        hs += code[v] * (n[v][0] - n[v][1]);
        hr += code[3 - v] * (n[v][q] - n[v][1 - q]);
    }
    *h_self = (uint32_t)hs & 31u;
    *h_rc = (uint32_t)hr & 31u;
}

// top 2p bits (first p bases) as an integer, p <= 16, p <= k
template <int W>
LDBG_HD uint32_t kmer_prefix(const Kmer<W>& a, int k, int p) {
    int sh = 2 * (k - p);   // shift right by sh over the 64W-bit string
    int wi = W - 1 - (sh >> 6);
    int b = sh & 63;
    uint64_t v = kmer_word<W>(a, wi) >> b;
    if (b != 0 && wi > 0) v |= kmer_word<W>(a, wi - 1) << (64 - b);
    return (uint32_t)(v & ((1ULL << (2 * p)) - 1ULL));
}

// ---- host-only ASCII conversion -------------------------------------------------------------
// Two rules, as in the reference.  ENCODING a k-mer (CortexRecord.encodeBinaryKmer -> charToBinaryNucleotide, CortexRecord.java:347-360) takes
// either case: ascii_to_words_ci, behind ldbg_kmer_encode only.  LOOKING a k-mer UP does not: findRecord(String) compares the query's bytes
// with the records' upper-case k-mers (CortexGraph.java:272-317), so "acgt…" has no record, exactly like a string with an N (quirk Q4) —
// every query path (find, walk and dfs seeds, sinks, seek, assemble, neighbours) goes through ascii_to_words / ascii_batch_to_words /
// walk.cpp: k_seed_words, which accept ACGT only.
inline bool ascii_to_words_impl(const char* s, int k, uint64_t* w, int W, bool any_case) {
    for (int i = 0; i < W; i++) w[i] = 0;
    for (int i = 0; i < k; i++) {
        uint64_t v;
        switch (any_case ? (s[i] & ~0x20) : s[i]) {
            case 'A': v = 0; break;
            case 'C': v = 1; break;
            case 'G': v = 2; break;
            case 'T': v = 3; break;
            default: return false;
        }
        int bit = 2 * (k - 1 - i);
        w[W - 1 - (bit >> 6)] |= v << (bit & 63);
    }
    return true;
}
inline bool ascii_to_words(const char* s, int k, uint64_t* w, int W) { return ascii_to_words_impl(s, k, w, W, false); }
inline bool ascii_to_words_ci(const char* s, int k, uint64_t* w, int W) { return ascii_to_words_impl(s, k, w, W, true); }
// the same for a batch of n k-mers of k bytes each (seeds of a walk / dfs batch): table lookup, one shift-or per base, a few
// threads for large batches.  valid[q] = 1 where string q is a k-mer over ACGT, 0 otherwise (findRecord then misses, quirk Q4):
// validity travels beside the words — at k = 32, 64, 96, 128 every bit pattern of the words is a k-mer, none is left for a mark.
inline void ascii_batch_to_words(const char* s, int64_t n, int k, int W, uint64_t* words, uint8_t* valid) {
    static const struct Lut {
        uint8_t v[256];
        Lut() { for (int i = 0; i < 256; i++) v[i] = 0x80; v['A'] = 0; v['C'] = 1; v['G'] = 2; v['T'] = 3; }       // (upper case only: a lookup, see above)
    } lut;
    const int nw = (k + 31) / 32, lead = W - nw;         // words that carry bases; leading all-zero words
    auto run = [&](int64_t lo, int64_t hi) {
        for (int64_t q = lo; q < hi; q++) {
            const uint8_t* c = (const uint8_t*)s + q * k;
            uint64_t* w = words + q * W;
            unsigned bad = 0;
            int i = 0;
            for (int wi = 0; wi < W; wi++) {
                const int cnt = wi < lead ? 0 : (wi == lead ? k - 32 * (nw - 1) : 32);
                uint64_t acc = 0;
                for (int j = 0; j < cnt; j++) { const uint8_t v = lut.v[c[i++]]; bad |= v; acc = (acc << 2) | (uint64_t)(v & 3u); }
                w[wi] = acc;
            }
            if (bad & 0x80u) for (int wi = 0; wi < W; wi++) w[wi] = 0ull;
            valid[q] = (bad & 0x80u) ? 0 : 1;
        }
    };
    const int nt = n >= 8192 ? (int)std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nt <= 1) { run(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(run, n * t / nt, n * (t + 1) / nt);
    for (auto& t : th) t.join();
}
inline void words_to_ascii(const uint64_t* w, int k, int W, char* out) {
    for (int i = 0; i < k; i++) {
        int bit = 2 * (k - 1 - i);
        out[i] = "ACGT"[(w[W - 1 - (bit >> 6)] >> (bit & 63)) & 3ULL];
    }
}

}  // namespace ldbg
