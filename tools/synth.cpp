// Synthetic LdBG workload generator for bench.py and the at-scale parity properties.
// NOT part of the product path and NOT the oracle: it only writes input files in the reference's
// on-disk formats — a sorted Cortex .ctx v6 graph (docs/ctx_spec.md), a .ctp.gz v4 link file
// (CortexLinksIterable.java:69-226) and a seed list.
//
// Model (deterministic, splitmix64): parent P1 = random genome in `n_chrom` chromosomes with
// interspersed repeat families and tandem repeats; parent P2 = P1 with SNVs and small indels; child =
// per-chromosome crossover mosaic of P1/P2 plus de novo mutations.  Colours: 0 child, 1 mom (P1),
// 2 dad (P2).  Coverage = number of occurrences of the k-mer in the colour's sequences, edges = union of
// flanking bases (the construction TempGraphAssembler.java:60-99 performs).  Links for the child are what
// TempLinksAssembler.java:29-105 derives from error-free reads of length `read_len` tiled every
// `read_stride` bases over both strands of the child, computed in closed form per anchor k-mer.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <string>
#include <thread>
#include <vector>

typedef unsigned __int128 u128;

extern "C" {
struct SynthParams {
    int64_t genome_len;
    int32_t k, n_chrom, colours, with_links;
    uint64_t seed;
    double gc, snv_rate;
    int32_t n_indels, n_dnm, n_tandem, n_repeat_families, repeat_copies, repeat_len_min, repeat_len_max;
    int32_t read_len, read_stride, n_seeds, threads;
};
struct SynthStats {
    int64_t n_records, n_link_kmers, n_links, n_seeds, n_novel_seeds, child_len;
};
int ldbg_synth_generate(const SynthParams* p, const char* out_prefix, SynthStats* st);
}

namespace {

struct Rng {
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};
typedef std::vector<uint8_t> Seq;   // base codes 0..3

uint8_t rand_base(Rng& r, double gc) { bool g = r.uniform() < gc; bool hi = r.next() & 1; return g ? (hi ? 1 : 2) : (hi ? 0 : 3); }

struct __attribute__((packed)) Occ { u128 key; uint8_t colour; uint8_t edge; };      // 18 bytes: a 1 Gb genome in 3 colours is 3 x 10^9 of them, twice over while they are bucketed
struct Rec { u128 key; uint32_t cov[3]; uint8_t edges[3]; };

inline u128 kmask(int k) { return k == 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }

// all k-mer occurrences of one sequence in one colour
void extract(const Seq& s, int k, int colour, std::vector<Occ>& out) {
    const int64_t n = (int64_t)s.size();
    if (n < k) return;
    const u128 mask = kmask(k);
    u128 fw = 0, rc = 0;
    for (int i = 0; i < k - 1; i++) { fw = (fw << 2) | s[i]; rc = (rc >> 2) | ((u128)(3 - s[i]) << (2 * k - 2)); }
    for (int64_t i = 0; i + k <= n; i++) {
        uint8_t b = s[i + k - 1];
        fw = ((fw << 2) | b) & mask;
        rc = (rc >> 2) | ((u128)(3 - b) << (2 * k - 2));
        bool flipped = rc < fw;
        int pb = i > 0 ? s[i - 1] : -1, nb = i + k < n ? s[i + k] : -1;
        uint8_t e = 0;
        // in-edge base X <-> bit 7-X ; out-edge base X <-> bit X  (CortexRecord.java:117-140)
        if (!flipped) { if (pb >= 0) e |= 1u << (7 - pb); if (nb >= 0) e |= 1u << nb; }
        else { if (nb >= 0) e |= 1u << (7 - (3 - nb)); if (pb >= 0) e |= 1u << (3 - pb); }
        out.push_back({flipped ? rc : fw, (uint8_t)colour, e});
    }
}

void write_u32(std::string& o, uint32_t v) { o.append((const char*)&v, 4); }
void write_u64(std::string& o, uint64_t v) { o.append((const char*)&v, 8); }

std::string ctx_header(int k, int C, const char* const* names) {
    std::string o = "CORTEX";
    int W = (k + 31) / 32;
    write_u32(o, 6); write_u32(o, k); write_u32(o, W); write_u32(o, C);
    for (int c = 0; c < C; c++) write_u32(o, 0);
    for (int c = 0; c < C; c++) write_u64(o, 0);
    for (int c = 0; c < C; c++) { write_u32(o, (uint32_t)strlen(names[c])); o += names[c]; }
    static const unsigned char err[16] = {0, 0xd8, 0xa3, 0x70, 0x3d, 0x0a, 0xd7, 0xa3, 0xf8, 0x3f, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; c++) o.append((const char*)err, 16);
    for (int c = 0; c < C; c++) { o.append(4, '\0'); write_u32(o, 0); write_u32(o, 0); write_u32(o, 0); }
    o += "CORTEX";
    return o;
}

std::string kmer_ascii(u128 key, int k) {
    std::string s(k, 'A');
    for (int i = 0; i < k; i++) s[i] = "ACGT"[(int)((key >> (2 * (k - 1 - i))) & 3)];
    return s;
}
u128 kmer_of(const Seq& s, int64_t pos, int k) { u128 v = 0; for (int i = 0; i < k; i++) v = (v << 2) | s[pos + i]; return v; }
u128 revcomp(u128 v, int k) { u128 r = 0; for (int i = 0; i < k; i++) { r = (r << 2) | (3 - (int)(v & 3)); v >>= 2; } return r; }

int64_t find_rec(const std::vector<Rec>& recs, u128 key) {
    int64_t lo = 0, hi = (int64_t)recs.size() - 1;
    while (lo <= hi) { int64_t m = (lo + hi) / 2; if (recs[m].key == key) return m; if (recs[m].key < key) lo = m + 1; else hi = m - 1; }
    return -1;
}
int popc4(int x) { return (x & 1) + ((x >> 1) & 1) + ((x >> 2) & 1) + ((x >> 3) & 1); }

struct LinkOut { u128 key; uint8_t is_fw; std::string junc; };

}  // namespace

extern "C" int ldbg_synth_generate(const SynthParams* pp, const char* out_prefix, SynthStats* st) {
    const SynthParams& p = *pp;
    const int k = p.k, C = p.colours;
    if (k < 3 || k > 64 || (C != 1 && C != 3)) return 1;
    Rng rng{p.seed};
    const int threads = std::max(1, p.threads);

    // ---- P1
    std::vector<Seq> p1(p.n_chrom);
    const int64_t clen = std::max<int64_t>(p.genome_len / p.n_chrom, 4 * k);
    for (auto& c : p1) { c.resize(clen); for (auto& b : c) b = rand_base(rng, p.gc); }
    for (int f = 0; f < p.n_repeat_families; f++) {
        int64_t L = p.repeat_len_min + (int64_t)rng.below(std::max(1, p.repeat_len_max - p.repeat_len_min + 1));
        L = std::min<int64_t>(L, clen / 4);
        Seq rep(L);
        for (auto& b : rep) b = rand_base(rng, p.gc);
        for (int c = 0; c < p.repeat_copies; c++) {
            Seq& ch = p1[rng.below(p.n_chrom)];
            int64_t pos = (int64_t)rng.below(clen - L);
            if (rng.next() & 1) for (int64_t i = 0; i < L; i++) ch[pos + i] = rep[i];
            else for (int64_t i = 0; i < L; i++) ch[pos + i] = 3 - rep[L - 1 - i];
        }
    }
    for (int t = 0; t < p.n_tandem; t++) {
        int64_t unit = 5 + (int64_t)rng.below(26), copies = 3 + (int64_t)rng.below(8);
        Seq& ch = p1[rng.below(p.n_chrom)];
        int64_t pos = (int64_t)rng.below(clen - unit * copies - 1);
        for (int64_t i = unit; i < unit * copies; i++) ch[pos + i] = ch[pos + i % unit];
    }
    // ---- P2 (SNVs + indels) with a coordinate map P1 -> P2
    std::vector<Seq> p2(p.n_chrom);
    std::vector<std::vector<int32_t>> map12(p.n_chrom);
    if (C == 3) {
        for (int c = 0; c < p.n_chrom; c++) {
            const Seq& a = p1[c];
            Seq& b = p2[c];
            auto& m = map12[c];
            m.resize(a.size());
            int indels_here = p.n_indels / p.n_chrom + (c < p.n_indels % p.n_chrom ? 1 : 0);
            double indel_rate = (double)indels_here / (double)a.size();
            for (int64_t i = 0; i < (int64_t)a.size(); i++) {
                m[i] = (int32_t)b.size();
                double u = rng.uniform();
                if (u < p.snv_rate) b.push_back((uint8_t)((a[i] + 1 + rng.below(3)) & 3));
                else if (u < p.snv_rate + indel_rate) {
                    if (rng.next() & 1) { b.push_back(a[i]); int n = 1 + (int)rng.below(10); for (int j = 0; j < n; j++) b.push_back(rand_base(rng, p.gc)); }
                    else {                                   // deletion
                        int64_t n = (int64_t)rng.below(10);
                        for (int64_t j = 1; j <= n && i + j < (int64_t)a.size(); j++) m[i + j] = (int32_t)b.size();
                        i += n;
                    }
                } else b.push_back(a[i]);
            }
        }
    }
    // ---- child: crossover mosaic + DNMs
    std::vector<Seq> child(p.n_chrom);
    struct Dnm { int chrom; int64_t pos; };
    std::vector<Dnm> dnms;
    for (int c = 0; c < p.n_chrom; c++) {
        if (C == 3) {
            int64_t x = (int64_t)(p1[c].size() / 4 + rng.below(p1[c].size() / 2));
            bool first_p1 = rng.next() & 1;
            int64_t x2 = map12[c][x];
            if (first_p1) { child[c].assign(p1[c].begin(), p1[c].begin() + x); child[c].insert(child[c].end(), p2[c].begin() + x2, p2[c].end()); }
            else { child[c].assign(p2[c].begin(), p2[c].begin() + x2); child[c].insert(child[c].end(), p1[c].begin() + x, p1[c].end()); }
        } else child[c] = p1[c];
    }
    if (C == 3) {
        for (int d = 0; d < p.n_dnm; d++) {
            int c = (int)rng.below(p.n_chrom);
            Seq& s = child[c];
            int64_t pos = k + (int64_t)rng.below(s.size() - 2 * k - 16);
            double u = rng.uniform();
            if (u < 0.7) s[pos] = (uint8_t)((s[pos] + 1 + rng.below(3)) & 3);                                  // SNV
            else if (u < 0.8) { int n = 1 + (int)rng.below(8); Seq ins(n); for (auto& b : ins) b = rand_base(rng, p.gc); s.insert(s.begin() + pos, ins.begin(), ins.end()); }
            else if (u < 0.9) { int n = 1 + (int)rng.below(8); s.erase(s.begin() + pos, s.begin() + pos + n); }
            else { for (int j = 0; j < 3; j++) s[pos + j] = (uint8_t)((s[pos + j] + 1 + rng.below(3)) & 3); } // MNP
            dnms.push_back({c, pos});
        }
    }
    int64_t child_len = 0;
    for (auto& s : child) child_len += (int64_t)s.size();

    if (child_len <= 5000000) {   // small runs: keep the child chromosomes so tests can re-derive reads
        std::string path = std::string(out_prefix) + ".child.txt";
        FILE* f = fopen(path.c_str(), "wb");
        if (f) {
            for (auto& s : child) { std::string a(s.size(), 'A'); for (size_t i = 0; i < s.size(); i++) a[i] = "ACGT"[s[i]]; fwrite(a.data(), 1, a.size(), f); fputc('\n', f); }
            fclose(f);
        }
    }

    // ---- k-mer occurrences, sorted, reduced to records
    std::vector<std::pair<const Seq*, int>> jobs;
    for (auto& s : child) jobs.push_back({&s, 0});
    if (C == 3) { for (auto& s : p1) jobs.push_back({&s, 1}); for (auto& s : p2) jobs.push_back({&s, 2}); }
    std::vector<std::vector<Occ>> parts(jobs.size());
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back([&] { size_t j; while ((j = next++) < jobs.size()) { parts[j].reserve(jobs[j].first->size()); extract(*jobs[j].first, k, jobs[j].second, parts[j]); } });
        for (auto& x : th) x.join();
    }
    // bucket by the first 4 bases, sort buckets in parallel
    const int NB = 256;
    const int shift = 2 * k - 8;
    std::vector<int64_t> bcount(NB + 1, 0);
    for (auto& v : parts) for (auto& o : v) bcount[(int)(o.key >> shift) + 1]++;
    for (int b = 0; b < NB; b++) bcount[b + 1] += bcount[b];
    std::vector<Occ> occ((size_t)bcount[NB]);
    {
        std::vector<int64_t> cur(bcount.begin(), bcount.end() - 1);
        for (auto& v : parts) { for (auto& o : v) occ[(size_t)cur[(int)(o.key >> shift)]++] = o; std::vector<Occ>().swap(v); }
    }
    {
        std::atomic<int> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back([&] { int b; while ((b = next++) < NB) std::sort(occ.begin() + bcount[b], occ.begin() + bcount[b + 1], [](const Occ& a, const Occ& c) { return a.key < c.key; }); });
        for (auto& x : th) x.join();
    }
    std::vector<Rec> recs;
    recs.reserve(occ.size() / 2);
    for (size_t i = 0; i < occ.size();) {
        Rec r; r.key = occ[i].key; memset(r.cov, 0, sizeof r.cov); memset(r.edges, 0, sizeof r.edges);
        size_t j = i;
        for (; j < occ.size() && occ[j].key == r.key; j++) { r.cov[occ[j].colour]++; r.edges[occ[j].colour] |= occ[j].edge; }
        recs.push_back(r);
        i = j;
    }
    std::vector<Occ>().swap(occ);

    // ---- write .ctx
    const int W = (k + 31) / 32;
    {
        static const char* names3[3] = {"child", "mom", "dad"};
        static const char* names1[1] = {"sample"};
        std::string path = std::string(out_prefix) + ".ctx";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return 2;
        std::string h = ctx_header(k, C, C == 3 ? names3 : names1);
        fwrite(h.data(), 1, h.size(), f);
        const size_t rs = 8 * W + 5 * C;
        std::vector<uint8_t> buf;
        buf.reserve(rs * 65536);
        for (size_t i = 0; i < recs.size(); i++) {
            const Rec& r = recs[i];
            uint64_t w[2] = {W == 2 ? (uint64_t)(r.key >> 64) : (uint64_t)r.key, (uint64_t)r.key};
            for (int x = 0; x < W; x++) { const uint8_t* b = (const uint8_t*)&w[x]; buf.insert(buf.end(), b, b + 8); }
            for (int c = 0; c < C; c++) { const uint8_t* b = (const uint8_t*)&r.cov[c]; buf.insert(buf.end(), b, b + 4); }
            for (int c = 0; c < C; c++) buf.push_back(r.edges[c]);
            if (buf.size() >= rs * 65536 || i + 1 == recs.size()) {
                if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); return 5; }      // (disk full: say so instead of leaving a short file)
                buf.clear();
            }
        }
        if (fclose(f) != 0) return 5;
    }

    // ---- child links
    int64_t n_link_kmers = 0, n_links = 0;
    if (p.with_links) {
        const int R = p.read_len, S = std::max(1, p.read_stride);
        std::vector<std::vector<LinkOut>> per_chrom(child.size());
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back([&] {
            size_t ci;
            while ((ci = next++) < child.size()) {
                const Seq& s = child[ci];
                const int64_t len = (int64_t)s.size(), nk = len - k + 1;
                if (nk < 2 || len < R) continue;
                // in/out degree of every k-mer occurrence in its forward orientation, child colour
                std::vector<uint8_t> din(nk), dout(nk);
                for (int64_t q = 0; q < nk; q++) {
                    u128 fw = kmer_of(s, q, k), rc = revcomp(fw, k);
                    bool fl = rc < fw;
                    int64_t ri = find_rec(recs, fl ? rc : fw);
                    uint8_t e = recs[ri].edges[0];
                    int lo = e & 0xf, hi = e >> 4;
                    dout[q] = (uint8_t)popc4(!fl ? lo : hi);
                    din[q] = (uint8_t)popc4(!fl ? hi : lo);
                }
                for (int strand = 0; strand < 2; strand++) {
                    // strand 1: the reverse complement sequence; k-mer q' there is revcomp of forward k-mer nk-1-q'
                    auto base = [&](int64_t x) -> uint8_t { return strand == 0 ? s[x] : (uint8_t)(3 - s[len - 1 - x]); };
                    auto d_out = [&](int64_t q) -> int { return strand == 0 ? dout[q] : din[nk - 1 - q]; };
                    auto d_in = [&](int64_t q) -> int { return strand == 0 ? din[q] : dout[nk - 1 - q]; };
                    // read starts on this strand: forward reads start at multiples of S; their reverse complements
                    // start at len - R - a on the other strand
                    auto has_start = [&](int64_t lo, int64_t hi) -> bool {   // a valid read start in [lo, hi]?
                        lo = std::max<int64_t>(lo, 0); hi = std::min<int64_t>(hi, len - R);
                        if (lo > hi) return false;
                        if (strand == 0) return (hi / S) * S >= lo;
                        // a' = len - R - a, a = m*S  ->  a' ≡ (len - R) mod S
                        int64_t r0 = (len - R) % S;
                        int64_t v = hi - ((hi - r0) % S + S) % S;
                        return v >= lo;
                    };
                    std::vector<int64_t> forks;
                    for (int64_t f = 0; f + 1 < nk; f++) if (d_out(f) > 1) forks.push_back(f);
                    const int64_t span = R - k - 1;   // a read starting at a covers forks f <= a + span
                    size_t fi = 0;
                    for (int64_t q = 0; q + 1 < nk; q++) {
                        if (d_in(q + 1) <= 1) continue;
                        while (fi < forks.size() && forks[fi] < q) fi++;
                        std::string junc;
                        for (size_t m = fi; m < forks.size() && forks[m] <= q + span; m++) {
                            junc.push_back("ACGT"[base(forks[m] + k)]);
                            int64_t next_f = m + 1 < forks.size() ? forks[m + 1] : (int64_t)1 << 60;
                            // a read contains q (a <= q), reaches fork m (a >= f_m - span) but not fork m+1 (a < f_{m+1} - span)
                            if (has_start(forks[m] - span, std::min(q, next_f - 1 - span))) {
                                u128 fw = 0;
                                for (int i = 0; i < k; i++) fw = (fw << 2) | base(q + i);
                                u128 rc = revcomp(fw, k);
                                bool fl = rc < fw;
                                per_chrom[ci].push_back({fl ? rc : fw, (uint8_t)(fl ? 0 : 1), junc});
                            }
                        }
                    }
                }
            }
        });
        for (auto& x : th) x.join();
        std::vector<LinkOut> all;
        for (auto& v : per_chrom) { all.insert(all.end(), v.begin(), v.end()); std::vector<LinkOut>().swap(v); }
        std::sort(all.begin(), all.end(), [](const LinkOut& a, const LinkOut& b) {
            if (a.key != b.key) return a.key < b.key;
            if (a.is_fw != b.is_fw) return a.is_fw > b.is_fw;
            return a.junc < b.junc;
        });
        all.erase(std::unique(all.begin(), all.end(), [](const LinkOut& a, const LinkOut& b) { return a.key == b.key && a.is_fw == b.is_fw && a.junc == b.junc; }), all.end());
        n_links = (int64_t)all.size();
        for (size_t i = 0; i < all.size(); i++) if (i == 0 || all[i].key != all[i - 1].key) n_link_kmers++;
        std::string path = std::string(out_prefix) + ".ctp.gz";
        gzFile gz = gzopen(path.c_str(), "wb1");
        if (!gz) return 3;
        char hdr[2048];
        snprintf(hdr, sizeof hdr,
                 "{\n  \"file_format\": \"ctp\",\n  \"format_version\": 4,\n  \"file_key\": 0,\n  \"graph\": {\n    \"num_colours\": 1,\n"
                 "    \"kmer_size\": %d,\n    \"num_kmers_in_graph\": %lld,\n    \"colours\": [{\n      \"colour\": 0,\n      \"sample\": \"%s\",\n"
                 "      \"total_sequence\": 0,\n      \"cleaned_tips\": false,\n      \"cleaned_unitigs\": false\n    }]\n  },\n  \"paths\": {\n"
                 "    \"num_kmers_with_paths\": %lld,\n    \"num_paths\": %lld,\n    \"path_bytes\": %lld\n  }\n}\n\n",
                 k, (long long)recs.size(), C == 3 ? "child" : "sample", (long long)n_link_kmers, (long long)n_links, (long long)n_links);
        gzwrite(gz, hdr, (unsigned)strlen(hdr));
        std::string out;
        for (size_t i = 0; i < all.size();) {
            size_t j = i;
            while (j < all.size() && all[j].key == all[i].key) j++;
            out += kmer_ascii(all[i].key, k) + " " + std::to_string(j - i) + "\n";
            for (size_t x = i; x < j; x++) out += std::string(all[x].is_fw ? "F " : "R ") + std::to_string(all[x].junc.size()) + " 1 " + all[x].junc + "\n";
            if (out.size() > (1 << 20)) { gzwrite(gz, out.data(), (unsigned)out.size()); out.clear(); }
            i = j;
        }
        out += "\n";
        gzwrite(gz, out.data(), (unsigned)out.size());
        gzclose(gz);
    }

    // ---- seeds: k-mers spanning de novo mutations that are child-only ("novel"), padded with random child k-mers
    std::vector<u128> seeds;
    int64_t n_novel = 0;
    for (auto& d : dnms) {
        const Seq& s = child[d.chrom];
        for (int64_t q = std::max<int64_t>(0, d.pos - k + 1); q <= d.pos && q + k <= (int64_t)s.size() && (int64_t)seeds.size() < p.n_seeds; q++) {
            u128 fw = kmer_of(s, q, k), rc = revcomp(fw, k);
            int64_t ri = find_rec(recs, std::min(fw, rc));
            if (ri >= 0 && recs[ri].cov[1] == 0 && recs[ri].cov[2] == 0) { seeds.push_back(fw); n_novel++; }
        }
    }
    while ((int64_t)seeds.size() < p.n_seeds) {
        const Seq& s = child[rng.below(child.size())];
        int64_t q = (int64_t)rng.below(s.size() - k);
        u128 fw = kmer_of(s, q, k);
        seeds.push_back((rng.next() & 1) ? fw : revcomp(fw, k));
    }
    {
        std::string path = std::string(out_prefix) + ".seeds";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return 4;
        for (auto v : seeds) { std::string a = kmer_ascii(v, k); fwrite(a.data(), 1, a.size(), f); }
        fclose(f);
    }
    if (st) { st->n_records = (int64_t)recs.size(); st->n_link_kmers = n_link_kmers; st->n_links = n_links; st->n_seeds = (int64_t)seeds.size(); st->n_novel_seeds = n_novel; st->child_len = child_len; }
    return 0;
}
