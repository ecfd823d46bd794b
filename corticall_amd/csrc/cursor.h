#pragma once
#include "engine_host.h"

namespace ldbg {

// host handle of the device-resident cursor of one engine (TraversalEngine.java:241-339)
class CursorHost {
public:
    explicit CursorHost(Engine& e);
    ~CursorHost();
    void seek(const char* kmer);
    bool has(bool fwd);
    void step(bool fwd, char* kmer_out, int64_t* rec_out);
    // assemble(seed) (TraversalEngine.java:112-145): vertices in contig order (packed k-mer words, record index or -1)
    void assemble(const char* seed, int64_t capacity, int64_t* len, uint64_t* words, int64_t* rec);

private:
    struct Impl;
    Engine& eng_;
    Impl* impl_;
    void peek(bool* has_next, bool* has_prev, uint32_t* status, uint64_t* out_words, int64_t* out_rec);
    static void check_status(uint32_t st);
    int64_t cur_record();
};

}  // namespace ldbg
