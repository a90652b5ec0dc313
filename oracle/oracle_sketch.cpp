// oracle/oracle_sketch.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
//
// CPU restatement of DipGenie's (w,k)-minimizer sketching:
//   hash128_to_64_            /root/reference/src/solver.cpp:16-24
//   MurmurHash3_x64_128       /root/reference/src/MurmurHash3.cpp:255-332 (Appleby, public domain;
//                             restated from the published algorithm)
//   reverse_strand_           /root/reference/src/misc.cpp:103-115
//   Solver::index_kmers       /root/reference/src/solver.cpp:277-363 (window loop :302-361)
//   Solver::compute_hashes    /root/reference/src/solver.cpp:366-412
//   Sp_R / kmer_count         /root/reference/src/solver.cpp:526-546, 711-732
// The window loop keeps the reference's string-based semantics on purpose (ASCII order, ties ->
// newest, emit on hash change), so non-ACGT input behaves identically.
#include "oracle.h"
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <vector>

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33; return k;
}

extern "C" void orc_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2]) {
    const uint8_t *data = (const uint8_t *)key;
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (int i = 0; i < nblocks; ++i) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
    case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
    case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
    case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
    case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
    case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
    case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
    case 9:  k2 ^= (uint64_t)tail[8];
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
    case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
    case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
    case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
    case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
    case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
    case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
    case 1:  k1 ^= (uint64_t)tail[0];
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

// solver.cpp:16-24
extern "C" uint64_t orc_hash_kmer(const char *s, int len) {
    uint64_t h[2];
    orc_murmur3_x64_128(s, len, 0, h);
    return h[0] ^ h[1];
}

// misc.cpp:103-115
static std::string revcomp(const std::string &seq) {
    std::string r;
    r.reserve(seq.size());
    for (int i = (int)seq.size() - 1; i >= 0; --i) {
        char c = seq[i];
        if (c == 'A' || c == 'a') r += 'T';
        else if (c == 'T' || c == 't') r += 'A';
        else if (c == 'C' || c == 'c') r += 'G';
        else if (c == 'G' || c == 'g') r += 'C';
        else r += c;
    }
    return r;
}

// The shared window loop of solver.cpp:302-361 (index_kmers) and :374-409 (compute_hashes).
template <class Emit>
static void window_loop(std::string s, int k, int w, Emit emit) {
    std::transform(s.begin(), s.end(), s.begin(), ::toupper);      // :288 / :368
    const int64_t n = (int64_t)s.size();
    if (n < (int64_t)w + k - 1) return;                            // :291 / :372
    uint64_t prev_hash = UINT64_MAX;                               // :302 / :374
    std::deque<std::pair<std::string, int64_t>> dq;
    for (int64_t i = 0; i <= n - k; ++i) {
        std::string fwd = s.substr(i, k);
        std::string rev = revcomp(fwd);
        std::string mn = std::min(fwd, rev);                       // :313
        while (!dq.empty() && dq.back().first >= mn) dq.pop_back();  // :316 ties -> newest
        dq.emplace_back(mn, i);
        if (!dq.empty() && dq.front().second <= i - w) dq.pop_front();   // :324
        if (i >= w - 1) {                                          // :329
            // NB the reference hashes std::string(best_kmer.c_str()): a NUL byte would truncate.
            std::string best(dq.front().first.c_str());
            uint64_t h = orc_hash_kmer(best.data(), (int)best.size());
            if (h != prev_hash) { prev_hash = h; emit(h, dq.front().second); }
        }
    }
}

extern "C" int64_t orc_minimizers(const char *seq, int64_t len, int k, int w,
                                  uint64_t *hash, int64_t *pos, int64_t cap) {
    int64_t n = 0;
    window_loop(std::string(seq, (size_t)len), k, w, [&](uint64_t h, int64_t p) {
        if (n < cap) { if (hash) hash[n] = h; if (pos) pos[n] = p; }
        ++n;
    });
    return n;
}

extern "C" int64_t orc_compute_hashes(const char *read, int64_t len, int k, int w, uint64_t *out, int64_t cap) {
    std::set<uint64_t> S;
    window_loop(std::string(read, (size_t)len), k, w, [&](uint64_t h, int64_t) { S.insert(h); });
    int64_t n = 0;
    for (uint64_t h : S) { if (n < cap) out[n] = h; ++n; }
    return n;
}

// solver.cpp:526-546 (Sp_R keyed by hash, value = #reads containing it) == kmer_count (:711-732)
extern "C" int orc_sketch_reads(const char *bases, const int64_t *read_off, int64_t n_reads, int k, int w,
                                uint64_t **hash, int32_t **cnt, int64_t *n_distinct) {
    std::map<uint64_t, int32_t> sp;
    for (int64_t r = 0; r < n_reads; ++r) {
        std::set<uint64_t> S;
        window_loop(std::string(bases + read_off[r], (size_t)(read_off[r + 1] - read_off[r])), k, w,
                    [&](uint64_t h, int64_t) { S.insert(h); });
        for (uint64_t h : S) sp[h]++;
    }
    *n_distinct = (int64_t)sp.size();
    *hash = (uint64_t *)malloc(sizeof(uint64_t) * (sp.size() + 1));
    *cnt = (int32_t *)malloc(sizeof(int32_t) * (sp.size() + 1));
    int64_t i = 0;
    for (auto &kv : sp) { (*hash)[i] = kv.first; (*cnt)[i] = kv.second; ++i; }
    return 0;
}

extern "C" void orc_free(void *p) { free(p); }
