// libdipgenie_hip.so -- (w,k)-minimizer sketching for MI355X (gfx950).
//
// Replaces the window loops of Solver::index_kmers / Solver::compute_hashes and the Sp_R /
// kmer_count maps (/root/reference/src/solver.cpp:302-361, 374-409, 526-546, 711-732).
// Semantics kept exactly: sequences are upper-cased; the canonical k-mer is the
// ASCII-lexicographic min of the k-mer and its reverse complement (non-ACGT bytes pass through the
// complement and order as ASCII); the window minimum takes the NEWEST on ties; a minimizer is
// emitted when its hash differs from the previous window's; hash = h1^h2 of MurmurHash3_x64_128 over
// the k ASCII bytes, seed 0 (restated from the published algorithm).
//
// Kernel shape: one wave (64 lanes) per tile of <=128 windows, 4 tiles per 256-thread workgroup.  The kernel is bound by its
// instruction count (one wave per 150-bp read), so every phase is written for few wave instructions: bases are staged four per
// lane (dword load, SWAR upper-casing / 2-bit codes / not-ACGT flags, OR-ed into a 2-bit stream and an invalid plane in LDS);
// every lane builds the canonical 2-bit codes of its k-mers (pure ACGT, k<=32, where 2-bit order == ASCII order) from the
// stream; window minima by doubling, in registers for tiles of <=128 k-mers; only window minima whose position changed are
// hashed, the workgroup's four tiles together by one wave.  K-mers containing other bytes fall to an in-kernel bytewise
// comparison -- same result, no host fallback.  Reads: every minimizer goes straight into the bucket of its hash range and the
// spectrum (Sp_R: distinct hashes, reads per hash) comes from one LDS table per bucket (dg_sketch_spectrum.hip); the generic
// route (sparse output, compaction, stable radix sort by hash + (hash,read) run flags + reduce_by_key, rocPRIM) is kept.
// Haplotypes: sparse output + compaction, order = sequence order.
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dg_sketch.hpp"

namespace dgi {

constexpr int TW = 128;                 // windows per tile (2 per lane)
constexpr int PLANE_WORDS = 12;         // 64-bit words per base bit plane: (TW + 255 + 255) / 64 + 2
#define PLANE_OFF(k, w) ((8 * (TW + (w)) + 4 * (TW + 1) + (TW + (w)) + (TW + (w) + (k)) + 7) & ~7)
// behind the planes: the window-minimum tables -- mc[TW + w] u64, ac[TW + 1] u64, mp[TW + w] u16, ap[TW + 1] u16
// behind the planes: the 2-bit base stream (base t at bits [2t, 2t + 1]), 2 words per 64 bases
#define STREAM_OFF(k, w) (PLANE_OFF(k, w) + 8 * 3 * PLANE_WORDS)
#define TABLE_OFF(k, w) (STREAM_OFF(k, w) + 8 * (2 * PLANE_WORDS + 1))
// orders this wave's LDS traffic across lanes (LDS is in order per wave; this keeps the compiler from moving accesses)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct Tile {
    int64_t seq_start;                  // offset of the sequence in the bases buffer
    int32_t seq_len;
    int32_t win0;                       // first window of the tile (window i covers k-mers i .. i+w-1)
    int32_t nwin;
    int32_t seq_id;
};

void sketch_state_free(SketchState *s) {
    if (!s) return;
    for (auto &e : s->ev) if (e) (void)hipEventDestroy(e);
    if (s->h_status) (void)hipHostFree(s->h_status);
    delete s;
}

// ------------------------------------------------------------------ MurmurHash3_x64_128 (h1^h2)
__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k;
}
// G(t) returns byte t of the key
template <class G>
__device__ __forceinline__ uint64_t murmur3_fold(G byte_at, int len) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = 0, h2 = 0;            // seed 0
    const int nblocks = len >> 4;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t k1 = 0, k2 = 0;
        for (int t = 0; t < 8; ++t) k1 |= (uint64_t)byte_at(16 * b + t) << (8 * t);
        for (int t = 0; t < 8; ++t) k2 |= (uint64_t)byte_at(16 * b + 8 + t) << (8 * t);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const int tail = nblocks << 4, rem = len & 15;
    uint64_t k1 = 0, k2 = 0;
    for (int t = 8; t < rem; ++t) k2 |= (uint64_t)byte_at(tail + t) << (8 * (t - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int t = 0; t < rem && t < 8; ++t) k1 |= (uint64_t)byte_at(tail + t) << (8 * t);
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 ^ h2;                     // solver.cpp:23
}

__device__ __forceinline__ uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
__device__ __forceinline__ uint8_t comp(uint8_t c) {    // misc.cpp:103-115 on upper-cased input
    return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
}
__device__ __forceinline__ int code2(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

// byte t of the canonical k-mer starting at LDS offset p (orientation o: 1 = reverse complement)
__device__ __forceinline__ uint8_t canon_byte(const uint8_t *s, int p, int k, int o, int t) {
    return o ? comp(s[p + k - 1 - t]) : s[p + t];
}
__device__ __forceinline__ int cmp_canon(const uint8_t *s, int k, int p, int op, int q, int oq) {
    for (int t = 0; t < k; ++t) {
        const uint8_t a = canon_byte(s, p, k, op, t), b = canon_byte(s, q, k, oq, t);
        if (a != b) return a < b ? -1 : 1;
    }
    return 0;
}

// spreads the low 32 bits of x to the even bit positions
__device__ __forceinline__ uint64_t spread32(uint64_t x) {
    x &= 0xFFFFFFFFULL;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}
// bits [q, q + n) of the bit string m[0] | m[1] << 64 | ... (n <= 32)
__device__ __forceinline__ uint32_t bits_at(const uint64_t *m, int q, int n) {
    const int wd = q >> 6, sh = q & 63;
    uint64_t v = m[wd] >> sh;
    if (sh) v |= m[wd + 1] << (64 - sh);
    return (uint32_t)(v & ((n >= 32) ? 0xFFFFFFFFULL : ((1ULL << n) - 1ULL)));
}
// 2k bits starting at bit 2q of the base stream: base q + t of the sequence at bits [2t, 2t + 1]
__device__ __forceinline__ uint64_t stream_at(const uint64_t *st, int q, int k) {
    const int wd = q >> 5, sh = (q & 31) << 1;
    uint64_t v = st[wd] >> sh;
    if (sh) v |= st[wd + 1] << (64 - sh);
    return k >= 32 ? v : (v & ((1ULL << (2 * k)) - 1ULL));
}
// reverses the order of the k 2-bit digits of x
__device__ __forceinline__ uint64_t reverse_digits(uint64_t x, int k) {
    uint64_t b = __brevll(x) >> (64 - 2 * k);                         // bits reversed: digits reversed, each digit's two bits swapped
    return ((b & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((b & 0x5555555555555555ULL) << 1);
}
// eight 2-bit codes (digit j of d = byte j of the result) -> eight ASCII bytes "ACGT"[code]
__device__ __forceinline__ uint64_t ascii8(uint32_t d) {
    uint64_t y = d & 0xFFFFu;
    y = (y | (y << 24)) & 0x000000FF000000FFULL;
    y = (y | (y << 12)) & 0x000F000F000F000FULL;
    y = (y | (y << 6)) & 0x0303030303030303ULL;
    const uint64_t h = (y >> 1) & 0x0101010101010101ULL;               // code >= 2
    return 0x4141414141414141ULL + 2 * y + 2 * h + 0x0B * (h & y);     // A 41, C 43, G 47 (41+4+2), T 54 (41+6+2+0B)
}
// MurmurHash3_x64_128 (h1^h2, seed 0) of the k ASCII bytes of a pure-ACGT k-mer given as its 2-bit code (first base most
// significant), k <= 32: the 8-byte words are generated from the code, no per-byte loop
__device__ __forceinline__ uint64_t murmur3_code(uint64_t code, int k) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    const uint64_t rd = reverse_digits(code, k);                       // base t at digit t
    auto word = [&](int first) -> uint64_t {                           // bytes first .. first+7 of the key (bytes past k are 0)
        const int left = k - first;
        if (left <= 0) return 0ULL;
        uint64_t a = ascii8((uint32_t)(rd >> (2 * first)));
        return left >= 8 ? a : (a & ((1ULL << (8 * left)) - 1ULL));
    };
    uint64_t h1 = 0, h2 = 0;
    const int nblocks = k >> 4;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t k1 = word(16 * b), k2 = word(16 * b + 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const int tail = nblocks << 4, rem = k & 15;
    if (rem > 8) { uint64_t k2 = word(tail + 8); k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    if (rem > 0) { uint64_t k1 = word(tail); k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 ^ h2;
}
// byte t of a k-mer given as 2-bit code (first base most significant)
__device__ __forceinline__ uint8_t code_byte(uint64_t code, int k, int t) {
    return (uint8_t)(0x54474341u >> (8 * (int)((code >> (2 * (k - 1 - t))) & 3ULL)));   // "ACGT"
}

// ------------------------------------------------------------------ tile kernel
// MODE 0: count emissions per tile; MODE 1: write hashes (+ aux: seq_id for reads, position for haplotypes) at tile_base;
// MODE 2: both in one pass -- tile_base holds SPARSE offsets (prefix of windows per tile, an upper bound
// of the emissions), a compaction kernel closes the gaps afterwards.  Two passes meant hashing everything twice.
// MODE 3 (reads, the one in use): every minimizer goes straight into the bucket of its hash range (dg_sketch_spectrum.hip):
// one returning atomic on the bucket's fill counter + two 8/4-byte stores per minimizer, hidden behind the kernel's ALU work.
// bucket_mode (MODE 2 / 3, reads): a hash is emitted once per tile (the Sp_R semantics are per read, solver.cpp:526-546, so
// a second emission of the same hash by the same read never counts) and reads of more than one tile carry bit 31 in their id.
template <int MODE, bool AUX_IS_POS>
__global__ __launch_bounds__(256) void sketch_tile_kernel(const char *__restrict__ bases, const Tile *__restrict__ tiles,
                                                          int64_t n_tiles, int k, int w, int64_t *__restrict__ tile_cnt,
                                                          const int64_t *__restrict__ tile_base, uint64_t *__restrict__ out_hash,
                                                          int64_t *__restrict__ out_aux, int lds_per_wave, int bucket_mode, BucketEmit be) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile_id = (int64_t)blockIdx.x * 4 + wave;
    const bool active = tile_id < n_tiles;              // inactive waves (the grid's last workgroup) only take part in the two barriers around the hashing
    Tile T{0, 0, 0, 0, 0};
    if (active) T = tiles[tile_id];
    const int has_prev = T.win0 > 0 ? 1 : 0;
    const int km0 = T.win0 - has_prev;                  // first k-mer (sequence coordinate) needed
    const int nkm = active ? T.nwin + has_prev + w - 1 : 0;   // k-mers needed
    const int nb = active ? nkm + k - 1 : 0;            // bases needed
    unsigned char *base = smem + (size_t)wave * lds_per_wave;
    uint64_t *code = (uint64_t *)base;                                 // [TW + w]
    int32_t *wpos = (int32_t *)(code + (TW + w));                      // [TW + 1] argmin k-mer (tile-local) per window
    uint8_t *info = (uint8_t *)(wpos + (TW + 1));                      // [TW + w] bit0 valid, bit1 orientation
    uint8_t *sq = info + (TW + w);                                     // [TW + w + k]

    const char *src = bases + T.seq_start + km0;
    // Staging, four bases per lane: one dword load, then SWAR -- upper-casing, 2-bit codes (A<C<G<T as in ASCII), "not ACGT" flags.
    // The four codes packed to a byte are OR-ed into the 32-base word of the 2-bit stream they belong to, the flags (a nibble) into the
    // invalid plane: one LDS atomic each.  (Before: a byte load, an LDS round trip and three ballots per base, and a 60-instruction bit
    // interleave per 32-base word -- a third of the kernel's instructions.)
    uint64_t *plane = (uint64_t *)(base + PLANE_OFF(k, w));           // [3][PLANE_WORDS]: (unused), (unused), not-ACGT
    uint64_t *stream = (uint64_t *)(base + STREAM_OFF(k, w));         // [2 * PLANE_WORDS + 1]: 2-bit codes, base t at bits [2t, 2t + 1]
    if (lane < 3 * PLANE_WORDS) plane[lane] = 0;
    if (lane < 2 * PLANE_WORDS + 1) stream[lane] = 0;
    WAVE_SYNC();                                                       // (the four waves of a workgroup work on tiles of their own: wave-level ordering is all the staging needs)
    bool tile_inv = false;
    {
        uint32_t *stream32 = (uint32_t *)stream, *inv32 = (uint32_t *)(plane + 2 * PLANE_WORDS);
        struct __attribute__((packed)) U32 { uint32_t v; };
        auto nz = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };   // bit 7 of every non-zero byte
        for (int t0 = 4 * lane; t0 < nb; t0 += 256) {
            uint32_t b = 0;
            if (t0 + 4 <= nb) b = ((const U32 *)(src + t0))->v;
            else for (int q = 0; t0 + q < nb; ++q) b |= (uint32_t)(uint8_t)src[t0 + q] << (8 * q);
            const uint32_t x7 = b & 0x7F7F7F7Fu;
            const uint32_t lower = (x7 + 0x1F1F1F1Fu) & ~(x7 + 0x05050505u) & ~b & 0x80808080u;    // 'a' + 0x1F = 0x80 = 'z' + 1 + 0x05
            const uint32_t u = b - (lower >> 2);
            for (int q = 0; q < 4 && t0 + q < nb; ++q) sq[t0 + q] = (uint8_t)(u >> (8 * q));
            const uint32_t bad = (nz(u ^ 0x41414141u) & nz(u ^ 0x43434343u) & nz(u ^ 0x47474747u) & nz(u ^ 0x54545454u)) >> 7;   // 1 per byte that is not A, C, G, T
            const uint32_t x = (u >> 1) & 0x03030303u;                 // A 0, C 1, T 2, G 3
            const uint32_t code = (x ^ ((x >> 1) & 0x01010101u)) & ~(bad * 3u);
            const uint32_t pack8 = ((code * 0x01041040u) >> 24) & 0xFFu, inv4 = ((bad * 0x01020408u) >> 24) & 0xFu;
            atomicOr(&stream32[t0 >> 4], pack8 << ((t0 & 15) << 1));
            if (inv4) atomicOr(&inv32[t0 >> 5], inv4 << (t0 & 31));
            tile_inv |= inv4 != 0;
        }
    }
    const bool any_inv = __ballot(tile_inv) != 0;                      // (most tiles are pure ACGT: their k-mers skip the look at the invalid plane)
    WAVE_SYNC();
#ifdef DG_TILE_DEBUG
    if (((bucket_mode >> 8) & 15) == 2) return;
#endif
    bool lane_inv = false;
    for (int q = lane; q < nkm; q += 64) {
        bool valid = (k <= 32) && (!any_inv || bits_at(plane + 2 * PLANE_WORDS, q, k) == 0);
        lane_inv |= !valid;
        int o;
        if (valid) {
            // base t of the k-mer at digit t = the reverse complement's digits once complemented (base k-1-t' complemented
            // is its digit t'... read from the top); the forward code wants base 0 most significant: digits reversed
            const uint64_t x = stream_at(stream, q, k);
            const uint64_t msk = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
            const uint64_t r = ~x & msk;
            const uint64_t f = reverse_digits(x, k);
            o = r < f ? 1 : 0;
            code[q] = o ? r : f;
        } else {
            o = cmp_canon(sq, k, q, 1, q, 0) < 0 ? 1 : 0;   // rc < fwd, bytewise
            code[q] = 0;
        }
        info[q] = (uint8_t)((valid ? 1 : 0) | (o << 1));
    }
    WAVE_SYNC();
#ifdef DG_TILE_DEBUG
    if (((bucket_mode >> 8) & 15) == 3) return;
#endif

    // window minima (ties -> newest, solver.cpp:316)
    const int nw_all = active ? T.nwin + has_prev : 0;
    uint64_t *mc = (uint64_t *)(base + TABLE_OFF(k, w));               // [TW + w]
    uint64_t *ac = mc + (TW + w);                                      // [TW + 1]
    uint16_t *mp = (uint16_t *)(ac + (TW + 1));                        // [TW + w]
    uint16_t *ap = mp + (TW + w);                                      // [TW + 1]
    if (__ballot(lane_inv) == 0 && nkm <= 128) {
        // every k-mer pure ACGT and at most two per lane (a 150-bp read): the doubling below with the table in REGISTERS -- lane l
        // keeps k-mers l and l + 64, an entry's partner h places up comes through ds_bpermute.  No LDS table, no wave syncs: the LDS
        // form was the largest phase of the kernel (0.50 of 1.47 ms on the config-4 reads).
        uint64_t c0 = lane < nkm ? code[lane] : 0, c1 = lane + 64 < nkm ? code[lane + 64] : 0;
        uint32_t p0 = (uint32_t)lane, p1 = (uint32_t)lane + 64u;
        uint64_t a0 = 0, a1 = 0;                                        // window minima: windows l and l + 64
        uint32_t q0 = 0, q1 = 0;
        auto fetch = [&](int dist, uint64_t &fc0, uint32_t &fp0, uint64_t &fc1, uint32_t &fp1, int &row_of0) {
            // entries (lane + dist) of the table as seen from row 0 (fc0 / fp0) and from row 1 (fc1 / fp1: row 1 only, else invalid)
            const int sl = ((lane + dist) & 63) << 2;
            const uint32_t x0l = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)(uint32_t)c0), x0h = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)(uint32_t)(c0 >> 32));
            const uint32_t x1l = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)(uint32_t)c1), x1h = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)(uint32_t)(c1 >> 32));
            const uint32_t y0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)p0), y1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)p1);
            row_of0 = (lane + dist) >> 6;                               // 0: the partner of a row-0 entry is in row 0, 1: in row 1, 2: beyond
            const bool r1 = row_of0 == 1;
            fc0 = r1 ? ((uint64_t)x1h << 32 | x1l) : ((uint64_t)x0h << 32 | x0l); fp0 = r1 ? y1 : y0;
            fc1 = (uint64_t)x1h << 32 | x1l; fp1 = y1;
        };
        int off = 0;
        bool first = true;
        for (int L = 1; L <= w; L <<= 1) {
            if (L > 1) {
                const int h = L >> 1;
                uint64_t fc0, fc1; uint32_t fp0, fp1; int ro;
                fetch(h, fc0, fp0, fc1, fp1, ro);
                const bool ok0 = lane + h < nkm && ro < 2, ok1 = ro == 0 && lane + 64 + h < nkm;
                if (ok0 && fc0 <= c0) { c0 = fc0; p0 = fp0; }
                if (ok1 && fc1 <= c1) { c1 = fc1; p1 = fp1; }
            }
            if (w & L) {
                uint64_t fc0 = c0, fc1 = c1; uint32_t fp0 = p0, fp1 = p1; int ro = 0;
                if (off) fetch(off, fc0, fp0, fc1, fp1, ro);
                // (windows exist only where all their k-mers do: wi + off < nkm for every wi < nw_all)
                if (lane < nw_all && (first || fc0 <= a0)) { a0 = fc0; q0 = fp0; }
                if (lane + 64 < nw_all && (first || fc1 <= a1)) { a1 = fc1; q1 = fp1; }
                off += L;
                first = false;
            }
        }
        if (lane < nw_all) wpos[lane] = (int32_t)q0;
        if (lane + 64 < nw_all) wpos[lane + 64] = (int32_t)q1;
    } else if (__ballot(lane_inv) == 0) {
        // every k-mer of the tile is pure ACGT: minima by doubling.  m_L[q] = (smallest code, its newest position) over
        // k-mers [q, q + L), built in place for L = 1, 2, 4, ... (m_2L[q] = m_L[q] (+) m_L[q + L], "later wins on <="); a
        // window of w k-mers is the concatenation of one range per set bit of w, lowest bit first.  ~8 LDS round trips per
        // k-mer instead of w of them.
        for (int q = lane; q < nkm; q += 64) { mc[q] = code[q]; mp[q] = (uint16_t)q; }
        WAVE_SYNC();
        int off = 0;
        bool first = true;
        for (int L = 1; L <= w; L <<= 1) {
            if (L > 1) {
                const int h = L >> 1;
                for (int q0 = 0; q0 < nkm; q0 += 64) {                 // ascending: a round only reads entries no round has rewritten yet
                    const int q = q0 + lane;
                    const bool ok = q + h < nkm;
                    uint64_t c1 = 0, c2 = 0;
                    uint16_t p2 = 0;
                    if (ok) { c1 = mc[q]; c2 = mc[q + h]; p2 = mp[q + h]; }
                    WAVE_SYNC();                                      // all loads of the round before its stores
                    if (ok && c2 <= c1) { mc[q] = c2; mp[q] = p2; }
                    WAVE_SYNC();
                }
            }
            if (w & L) {
                for (int wi = lane; wi < nw_all; wi += 64) {
                    const uint64_t c2 = mc[wi + off];
                    const uint16_t p2 = mp[wi + off];
                    if (first || c2 <= ac[wi]) { ac[wi] = c2; ap[wi] = p2; }
                }
                off += L;
                first = false;
                WAVE_SYNC();
            }
        }
        for (int wi = lane; wi < nw_all; wi += 64) wpos[wi] = (int32_t)ap[wi];
    } else {
        for (int wi = lane; wi < nw_all; wi += 64) {                   // tiles holding N / IUPAC bytes: plain scan, bytewise where needed
            int best = wi;
            bool allv = true;
            for (int t = 0; t < w; ++t) allv = allv && (info[wi + t] & 1);
            if (allv) {
                uint64_t bc = code[wi];
                for (int t = 1; t < w; ++t) { const uint64_t c = code[wi + t]; if (c <= bc) { bc = c; best = wi + t; } }
            } else {
                for (int t = 1; t < w; ++t)
                    if (cmp_canon(sq, k, wi + t, (info[wi + t] >> 1) & 1, best, (info[best] >> 1) & 1) <= 0) best = wi + t;
            }
            wpos[wi] = best;
        }
    }
    WAVE_SYNC();
#ifdef DG_TILE_DEBUG
    if (((bucket_mode >> 8) & 15) == 4) return;
#endif

    // emission (solver.cpp:329-335 / 401-407): a minimizer is emitted where its hash differs from the previous window's.
    // Windows sharing their argmin form runs; every run's k-mer is hashed ONCE (one lane per run), then run j is emitted iff
    // its hash differs from run j - 1's (run 0 of a tile that continues a sequence is the previous tile's last window: context
    // only; run 0 of a sequence's first tile compares with prev_hash = UINT64_MAX).
    uint16_t *run_w = mp;                                              // first window of every run
    uint64_t *run_h = mc;                                              // its hash
    int n_runs = 0;
    for (int q0 = 0; q0 < nw_all; q0 += 64) {
        const int wi = q0 + lane;
        const bool head = wi < nw_all && (wi == 0 || wpos[wi] != wpos[wi - 1]);
        const unsigned long long m = __ballot(head);
        if (head) run_w[n_runs + __popcll(m & ((1ULL << lane) - 1ULL))] = (uint16_t)wi;
        n_runs += __popcll(m);
    }
    // Hashing is the costliest phase per instruction (~300 for MurmurHash3 of one k-mer) and the emptiest: a 150-bp read has ~8 runs,
    // 8 lanes of 64.  The kernel is bound by its instruction count, so the runs of the workgroup's four tiles are hashed TOGETHER by
    // one wave (their records are in LDS already; ~32 lanes busy, the code runs once instead of four times).
    int *hdr = (int *)(smem + 4 * (size_t)lds_per_wave);               // [0..3] runs per wave; [4..7] read id | multi-tile << 31; [8..11] has_prev
    if (lane == 0) {
        hdr[wave] = n_runs;
        hdr[4 + wave] = (int)((uint32_t)T.seq_id | ((int64_t)T.seq_len - k - w + 2 > TW ? 1u << 31 : 0u));
        hdr[8 + wave] = has_prev;
    }
    __syncthreads();
    const bool hasher = wave == (int)(blockIdx.x & 3);                 // (rotating: a workgroup's wave i sits on SIMD i -- always wave 0 would load one SIMD of four)
    if (MODE == 3 && !hasher) return;                                  // bucket form: the hashing wave also emits for the whole workgroup (below)
    if (hasher) {
        const int p1 = hdr[0], p2 = p1 + hdr[1], p3 = p2 + hdr[2], tot = p3 + hdr[3];
        for (int g0 = 0; g0 < tot; g0 += 64) {
            const int g = g0 + lane;
            if (g < tot) {
                const int v = (g >= p1 ? 1 : 0) + (g >= p2 ? 1 : 0) + (g >= p3 ? 1 : 0);
                const int j = g - (v == 0 ? 0 : (v == 1 ? p1 : (v == 2 ? p2 : p3)));
                unsigned char *bv = smem + (size_t)v * lds_per_wave;
                const uint64_t *code_v = (const uint64_t *)bv;
                const int32_t *wpos_v = (const int32_t *)(code_v + (TW + w));
                const uint8_t *info_v = (const uint8_t *)(wpos_v + (TW + 1)), *sq_v = info_v + (TW + w);
                uint64_t *mc_v = (uint64_t *)(bv + TABLE_OFF(k, w));
                const uint16_t *mp_v = (const uint16_t *)(mc_v + (TW + w) + (TW + 1));
                const int p = wpos_v[mp_v[j]];
                const int op = (info_v[p] >> 1) & 1;
                const bool vp = info_v[p] & 1;
                const uint64_t cp = code_v[p];
                if (vp) {
                    mc_v[j] = murmur3_code(cp, k);                     // pure ACGT: the key's words come straight from the code
                } else {
                    auto bp = [&](int t) -> uint8_t { return canon_byte(sq_v, p, k, op, t); };
                    mc_v[j] = murmur3_fold(bp, k);
                }
            }
        }
        if (MODE == 3) {
            // ... and the emission of all four tiles: one pass of the emission code per workgroup instead of four (it was a fifth of
            // the kernel), the returning atomics of ~32 minimizers in one instruction.  Same rules as below: run j of a tile is
            // emitted iff its hash differs from run j - 1's (context run 0 of a continuing tile: never) and from every earlier run's.
            WAVE_SYNC();
            for (int g0 = 0; g0 < tot; g0 += 64) {
                const int g = g0 + lane;
                if (g < tot) {
                    const int v = (g >= p1 ? 1 : 0) + (g >= p2 ? 1 : 0) + (g >= p3 ? 1 : 0);
                    const int j = g - (v == 0 ? 0 : (v == 1 ? p1 : (v == 2 ? p2 : p3)));
                    const uint64_t *mc_v = (const uint64_t *)(smem + (size_t)v * lds_per_wave + TABLE_OFF(k, w));
                    const int hp = hdr[8 + v];
                    const uint64_t H = mc_v[j];
                    bool emit = j >= hp && H != (j == 0 ? UINT64_MAX : mc_v[j - 1]);
                    if (emit)                                          // (a first hash equal to the initial prev_hash is never emitted: it shadows nothing, solver.cpp:329)
                        for (int a = (!hp && mc_v[0] == UINT64_MAX) ? 1 : 0; a < j - 1; ++a) if (mc_v[a] == H) { emit = false; break; }
                    if (emit) {
                        const uint32_t b = (uint32_t)(H >> (64 - be.bbits));
                        const uint32_t pos = atomicAdd(&be.fill[(size_t)b * FILL_PAD], 1u);   // one counter per 64-byte line: atomics on one line serialise at the memory side
                        if (pos < be.stride) {
                            const size_t at = (size_t)b * be.stride + pos;
                            be.bk_hash[at] = H;
                            be.bk_read[at] = (uint32_t)hdr[4 + v];
                        } else {                                       // the bucket is full (a hash held by > ~10 k reads): the shared spill list
                            const uint32_t sp = atomicAdd(be.spill_n, 1u);
                            if (sp < be.spill_cap) { be.spill_hash[sp] = H; be.spill_read[sp] = (uint32_t)hdr[4 + v]; }
                        }
                    }
                }
            }
            return;
        }
    }
    __syncthreads();
    if (!active) return;
#ifdef DG_TILE_DEBUG
    if (((bucket_mode >> 8) & 15) == 5) return;
#endif
    int64_t wbase = (MODE == 1 || MODE == 2) ? tile_base[tile_id] : 0;
    int64_t total = 0;
    const bool multi_tile = (int64_t)T.seq_len - k - w + 2 > TW;
    const bool dedupe = !AUX_IS_POS && bucket_mode;
    // a first hash equal to the initial prev_hash is never emitted (solver.cpp:329): it does not shadow later runs
    const int a_first = (dedupe && !has_prev && run_h[0] == UINT64_MAX) ? 1 : 0;
    for (int j0 = has_prev; j0 < n_runs; j0 += 64) {
        const int j = j0 + lane;
        bool emit = false;
        uint64_t H = 0;
        int p = 0;
        if (j < n_runs) {
            H = run_h[j];
            p = wpos[run_w[j]];
            emit = H != (j == 0 ? UINT64_MAX : run_h[j - 1]);
        }
        if (dedupe) {                                                  // run j repeats the hash of an earlier run of this tile
            // the kernel is issue-bound (~4 k cycles of its SIMD per tile), so this loop is scalar-controlled: two broadcast
            // reads per round, compares against SGPR bounds
            const int last = __builtin_amdgcn_readfirstlane(min(n_runs, j0 + 64) - 1), first = __builtin_amdgcn_readfirstlane(a_first);
            bool dup = false;
            int a = first;
            for (; a + 1 < last; a += 2) {
                const uint64_t x0 = run_h[a], x1 = run_h[a + 1];
                dup |= ((x0 == H) & (a < j - 1)) | ((x1 == H) & (a + 1 < j - 1));
            }
            if (a < last) dup |= (run_h[a] == H) & (a < j - 1);
            emit &= !dup;
        }
        const unsigned long long m = __ballot(emit);
        if ((MODE == 1 || MODE == 2) && emit) {
            const int64_t slot = wbase + total + __popcll(m & ((1ULL << lane) - 1ULL));
            out_hash[slot] = H;
            out_aux[slot] = AUX_IS_POS ? (int64_t)(km0 + p) : ((int64_t)T.seq_id | ((bucket_mode && multi_tile) ? (int64_t)1 << 31 : 0));
        }
        total += __popcll(m);
    }
    if ((MODE == 0 || MODE == 2) && lane == 0) tile_cnt[tile_id] = total;
}

// closes the gaps of the single-pass output: one wave per tile copies its cnt entries from the sparse to the dense offset
__global__ __launch_bounds__(256) void compact_tiles_kernel(const int64_t *__restrict__ sparse_base, const int64_t *__restrict__ dense_base,
                                                            const int64_t *__restrict__ cnt, int64_t n_tiles, const uint64_t *__restrict__ sh,
                                                            const int64_t *__restrict__ sa, uint64_t *__restrict__ dh, int64_t *__restrict__ da) {
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int lane = threadIdx.x & 63;
    const int64_t sb = sparse_base[tile], db = dense_base[tile], n = cnt[tile];
    for (int64_t t = lane; t < n; t += 64) { dh[db + t] = sh[sb + t]; da[db + t] = sa[sb + t]; }
}

// Tile descriptors are built on the device from the sequence offsets (a million reads: no 8 MB offset download, no host
// loop, no 32 MB descriptor upload per call).  seq_count: per sequence its tiles and windows; after exclusive scans of
// both, seq_fill writes the descriptors and the sparse output offsets (a tile emits at most one minimizer per window).
struct SeqCount { int64_t tiles, wins, multi; };   // per sequence (multi: 1 if it has several tiles); after the exclusive scan: first tile / first window (entry n_seq = totals)
struct SeqCountPlus { __host__ __device__ SeqCount operator()(const SeqCount &a, const SeqCount &b) const { return SeqCount{a.tiles + b.tiles, a.wins + b.wins, a.multi + b.multi}; } };
__global__ void seq_count_kernel(const int64_t *__restrict__ off, int64_t n_seq, int k, int w, SeqCount *__restrict__ cnt) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seq) return;
    int64_t nw = 0;
    if (s < n_seq) nw = max((off[s + 1] - off[s]) - k - w + 2, (int64_t)0);      // solver.cpp:291 / 372: nothing if len < w+k-1
    cnt[s] = SeqCount{(nw + TW - 1) / TW, nw, nw > TW ? 1 : 0};
}
__global__ __launch_bounds__(256) void seq_fill_kernel(const int64_t *__restrict__ off, int64_t n_seq, int k, int w, const SeqCount *__restrict__ first,
                                                       Tile *__restrict__ tiles, int64_t *__restrict__ sparse) {
    // short sequences (reads: one tile each): one thread per sequence; a long one (a haplotype) is strided by the whole grid
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if (n_seq == 1) {
        const int64_t len = off[1] - off[0], nw = first[1].wins, nt = first[1].tiles;
        for (int64_t t = tid; t < nt; t += nth) {
            tiles[t] = Tile{off[0], (int32_t)len, (int32_t)(t * TW), (int32_t)min((int64_t)TW, nw - t * TW), 0};
            sparse[t] = t * TW;
        }
        if (tid == 0) sparse[nt] = nw;
        return;
    }
    for (int64_t s = tid; s <= n_seq; s += nth) {
        const SeqCount f = first[s];
        if (s == n_seq) { sparse[f.tiles] = f.wins; break; }
        const int64_t len = off[s + 1] - off[s], nw = first[s + 1].wins - f.wins, t0 = f.tiles;
        for (int64_t q = 0; q * TW < nw; ++q) {
            tiles[t0 + q] = Tile{off[s], (int32_t)len, (int32_t)(q * TW), (int32_t)min((int64_t)TW, nw - q * TW), (int32_t)s};
            sparse[t0 + q] = f.wins + q * TW;
        }
    }
}

__global__ void pair_flag_kernel(const uint64_t *__restrict__ hash, const int64_t *__restrict__ seq, int64_t n, int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || hash[i] != hash[i - 1] || seq[i] != seq[i - 1]) ? 1 : 0;   // one per distinct (hash, read)
}

__global__ void hash_kmers_kernel(const char *__restrict__ kmers, int64_t n, int k, uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char *s = (const unsigned char *)kmers + i * k;
    auto b = [&](int t) -> uint8_t { return s[t]; };
    out[i] = murmur3_fold(b, k);
}

__global__ void dict_count_kernel(const uint64_t *__restrict__ dict, int64_t n_dict, const uint64_t *__restrict__ hash,
                                  const int32_t *__restrict__ cnt, int64_t n, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dict) return;
    const uint64_t key = dict[i];
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (hash[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < n && hash[lo] == key) out[i] += cnt[lo];
}

// ---- read-sharded scoring (one rank per GPU): hash-range partition, global ranks, multiplicity histogram ----
// owner of a hash when the uint64 space is cut into `world` equal ranges (hashes are MurmurHash3 outputs: uniform)
__host__ __device__ __forceinline__ int hash_owner(uint64_t h, int world) { return (int)(((h >> 32) * (uint64_t)world) >> 32); }

// split[r] = first index of the sorted hash list whose owner is >= r  (r = 0 .. world; split[world] = n)
__global__ void partition_kernel(const uint64_t *__restrict__ hash, int64_t n, int world, int64_t *__restrict__ split) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r > world) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (hash_owner(hash[mid], world) < r) lo = mid + 1; else hi = mid; }
    split[r] = lo;
}

// rank1[i] += base + idx + 1 for every dictionary hash found in this rank's range of the merged spectrum (owner only:
// a sum over ranks leaves global id + 1, 0 = not a read minimizer)
__global__ void dict_rank_kernel(const uint64_t *__restrict__ dict, int64_t n_dict, const uint64_t *__restrict__ hash, int64_t n, int64_t base,
                                 int64_t *__restrict__ rank1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dict) return;
    const uint64_t key = dict[i];
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (hash[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < n && hash[lo] == key) rank1[i] += base + lo + 1;
}

// hist[min(count, n_bins - 1)] += 1 per spectrum entry (solver.cpp:745-755 Hist_kmer, sharded).  Nearly all the mass sits
// on a handful of small multiplicities: counts below 8 are tallied per lane in registers and reduced over the wave (one
// global atomic per wave and bin), the rest goes through a workgroup-private LDS histogram.
constexpr int HIST_LDS = 4096;
__global__ __launch_bounds__(256) void mult_hist_kernel(const int32_t *__restrict__ cnt, int64_t n, int n_bins, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int lh[HIST_LDS];
    for (int q = threadIdx.x; q < HIST_LDS; q += 256) lh[q] = 0;
    __syncthreads();
    unsigned int low[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = min(max(cnt[i], 0), n_bins - 1);
        if (m < 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) low[q] += m == q ? 1u : 0u;
        } else if (m < HIST_LDS) atomicAdd(&lh[m], 1u);
        else atomicAdd(&hist[m], 1ULL);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        unsigned int v = low[q];
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_down(v, sft);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&lh[q], v);           // per workgroup first: a global word takes ~88 atomics per microsecond
    }
    __syncthreads();
    for (int q = threadIdx.x; q < HIST_LDS && q < n_bins; q += 256)
        if (lh[q]) atomicAdd(&hist[q], (unsigned long long)lh[q]);
}

// dict_count_kernel and dict_rank_kernel on the same list (one rank: the local spectrum is the global one): one binary search
__global__ void dict_count_rank_kernel(const uint64_t *__restrict__ dict, int64_t n_dict, const uint64_t *__restrict__ hash,
                                       const int32_t *__restrict__ cnt, int64_t n, int64_t base, int32_t *__restrict__ out, int64_t *__restrict__ rank1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dict) return;
    const uint64_t key = dict[i];
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (hash[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < n && hash[lo] == key) { out[i] += cnt[lo]; rank1[i] += base + lo + 1; }
}

// ------------------------------------------------------------------ host side
#ifdef DG_TILE_DEBUG                     // measurement build (tools/tile_phases.sh): the tile kernel returns after phase $DG_TILE_DEBUG
#define DG_TILE_DEBUG_BITS (getenv("DG_TILE_DEBUG") ? atoi(getenv("DG_TILE_DEBUG")) << 8 : 0)
#else
#define DG_TILE_DEBUG_BITS 0
#endif
static SketchState &state(dg_ctx *c) {
    if (!c->sk) c->sk = new SketchState();
    return *c->sk;
}

static size_t lds_per_wave(int k, int w) {
    size_t b = (size_t)TABLE_OFF(k, w) + 8 * (size_t)(TW + w) + 8 * (size_t)(TW + 1) + 2 * (size_t)(TW + w) + 2 * (size_t)(TW + 1) + 8;
    return (b + 15) & ~(size_t)15;
}

// Tile descriptors for device-resident bases and sequence offsets (one synchronisation: the tile and window totals size
// the launch and the arrays).
static int prepare_tiles(dg_ctx *c, const int64_t *off_dev, int64_t n_seq, int k, int w, int64_t *n_tiles, int64_t *n_windows, int64_t *n_multi = nullptr) {
    SketchState &S = state(c);
    hipStream_t s = c->stream;
    *n_tiles = 0; *n_windows = 0;
    if (n_seq <= 0) return DG_OK;
    // per-sequence tile / window counts and their exclusive scan (entry n_seq = totals): one scan of the pair
    if (int rc = S.d_seq_tiles.ensure(sizeof(SeqCount) * (size_t)(n_seq + 1))) return rc;
    if (int rc = S.d_seq_tile0.ensure(sizeof(SeqCount) * (size_t)(n_seq + 1))) return rc;
    hipLaunchKernelGGL(seq_count_kernel, dim3((unsigned)((n_seq + 1 + 255) / 256)), dim3(256), 0, s, off_dev, n_seq, k, w, S.d_seq_tiles.as<SeqCount>());
    size_t tb = 0;
    DG_HIP(rocprim::exclusive_scan(nullptr, tb, S.d_seq_tiles.as<SeqCount>(), S.d_seq_tile0.as<SeqCount>(), SeqCount{0, 0, 0}, (size_t)(n_seq + 1), SeqCountPlus(), s));
    if (int rc = S.d_tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::exclusive_scan(S.d_tmp.p, tb, S.d_seq_tiles.as<SeqCount>(), S.d_seq_tile0.as<SeqCount>(), SeqCount{0, 0, 0}, (size_t)(n_seq + 1), SeqCountPlus(), s));
    int64_t tot[3] = {0, 0, 0};
    DG_HIP(hipMemcpyAsync(tot, S.d_seq_tile0.as<SeqCount>() + n_seq, 24, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    if (n_multi) *n_multi = tot[2];
    const int64_t nt = tot[0], n_win = tot[1];
    if (nt == 0) return DG_OK;
    if (int rc = S.d_tiles.ensure(sizeof(Tile) * nt)) return rc;
    if (int rc = S.d_tile_cnt.ensure(8 * (nt + 1))) return rc;
    if (int rc = S.d_tile_base.ensure(8 * (nt + 1))) return rc;
    if (int rc = S.d_tile_sparse.ensure(8 * (nt + 1))) return rc;
    hipLaunchKernelGGL(seq_fill_kernel, dim3((unsigned)std::min<int64_t>(((n_seq == 1 ? nt : n_seq + 1) + 255) / 256, 4096)), dim3(256), 0, s, off_dev, n_seq, k, w,
                       S.d_seq_tile0.as<SeqCount>(), S.d_tiles.as<Tile>(), S.d_tile_sparse.as<int64_t>());
    DG_HIP(hipGetLastError());
    *n_tiles = nt; *n_windows = n_win;
    return DG_OK;
}

// The tile kernel, sparse form: on return tile t's minimizers are the d_tile_cnt[t] entries at d_tile_sparse[t] of d_hash2 /
// d_aux2 (a tile emits at most one per window, so the slots never collide).
template <bool AUX_IS_POS>
static int launch_tiles_sparse(dg_ctx *c, const char *bases_dev, int64_t nt, int64_t n_win, int k, int w, int bucket_mode) {
    SketchState &S = state(c);
    if (nt == 0) return DG_OK;
    if (int rc = S.d_hash2.ensure(8 * (size_t)std::max<int64_t>(n_win, 1))) return rc;     // sparse output (dead before the sort reuses them)
    if (int rc = S.d_aux2.ensure(8 * (size_t)std::max<int64_t>(n_win, 1))) return rc;
    const size_t lpw = lds_per_wave(k, w);
    hipLaunchKernelGGL((sketch_tile_kernel<2, AUX_IS_POS>), dim3((unsigned)((nt + 3) / 4)), dim3(256), 4 * lpw + 16, c->stream, bases_dev, S.d_tiles.as<Tile>(), nt, k, w,
                       S.d_tile_cnt.as<int64_t>(), S.d_tile_sparse.as<int64_t>(), S.d_hash2.as<uint64_t>(), S.d_aux2.as<int64_t>(), (int)lpw, bucket_mode, BucketEmit{});
    DG_HIP(hipGetLastError());
    return DG_OK;
}

// The tile kernel, bucket form (reads): minimizers go straight into the buckets `be` describes
static int launch_tiles_buckets(dg_ctx *c, const char *bases_dev, int64_t nt, int k, int w, const BucketEmit &be) {
    SketchState &S = state(c);
    if (nt == 0) return DG_OK;
    const size_t lpw = lds_per_wave(k, w);
    hipLaunchKernelGGL((sketch_tile_kernel<3, false>), dim3((unsigned)((nt + 3) / 4)), dim3(256), 4 * lpw + 16, c->stream, bases_dev, S.d_tiles.as<Tile>(), nt, k, w,
                       (int64_t *)nullptr, (const int64_t *)nullptr, (uint64_t *)nullptr, (int64_t *)nullptr, (int)lpw, 1 | DG_TILE_DEBUG_BITS, be);
    DG_HIP(hipGetLastError());
    return DG_OK;
}

// closes the gaps of the sparse output: on return d_hash / d_aux hold the n_emit minimizers in sequence order
static int compact_tiles(dg_ctx *c, int64_t nt, int64_t *n_emit) {
    SketchState &S = state(c);
    hipStream_t s = c->stream;
    *n_emit = 0;
    if (nt == 0) return DG_OK;
    // exclusive scan of counts (as int64) -> tile_base; total at [nt]
    DG_HIP(hipMemsetAsync((char *)S.d_tile_cnt.p + 8 * nt, 0, 8, s));
    const int64_t *in = S.d_tile_cnt.as<int64_t>();
    size_t tb = 0;
    DG_HIP(rocprim::exclusive_scan(nullptr, tb, in, S.d_tile_base.as<int64_t>(), (int64_t)0, (size_t)(nt + 1), rocprim::plus<int64_t>(), s));
    if (int rc = S.d_tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::exclusive_scan(S.d_tmp.p, tb, in, S.d_tile_base.as<int64_t>(), (int64_t)0, (size_t)(nt + 1), rocprim::plus<int64_t>(), s));
    int64_t total = 0;
    DG_HIP(hipMemcpyAsync(&total, S.d_tile_base.as<int64_t>() + nt, 8, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    *n_emit = total;
    if (total == 0) return DG_OK;
    if (int rc = S.d_hash.ensure(8 * total)) return rc;
    if (int rc = S.d_aux.ensure(8 * total)) return rc;
    hipLaunchKernelGGL(compact_tiles_kernel, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, s, S.d_tile_sparse.as<int64_t>(), S.d_tile_base.as<int64_t>(), S.d_tile_cnt.as<int64_t>(),
                       nt, S.d_hash2.as<uint64_t>(), S.d_aux2.as<int64_t>(), S.d_hash.as<uint64_t>(), S.d_aux.as<int64_t>());
    DG_HIP(hipGetLastError());
    return DG_OK;
}

template <bool AUX_IS_POS>
static int run_tiles(dg_ctx *c, const char *bases_dev, const int64_t *off_dev, int64_t n_seq, int k, int w, int64_t *n_emit) {
    int64_t nt = 0, n_win = 0;
    *n_emit = 0;
    if (int rc = prepare_tiles(c, off_dev, n_seq, k, w, &nt, &n_win)) return rc;
    if (int rc = launch_tiles_sparse<AUX_IS_POS>(c, bases_dev, nt, n_win, k, w, 0)) return rc;
    return compact_tiles(c, nt, n_emit);
}

// (hash, read) pairs in d_hash/d_aux -> sorted distinct hashes + #reads in d_uniq/d_cnt; returns n_distinct
static int spectrum_from_pairs(dg_ctx *c, int64_t n, int64_t *n_distinct) {
    SketchState &S = state(c);
    hipStream_t s = c->stream;
    *n_distinct = 0;
    if (n == 0) return DG_OK;
    if (int rc = S.d_hash2.ensure(8 * n)) return rc;
    if (int rc = S.d_aux2.ensure(8 * n)) return rc;
    size_t tb = 0;
    DG_HIP(rocprim::radix_sort_pairs(nullptr, tb, S.d_hash.as<uint64_t>(), S.d_hash2.as<uint64_t>(), S.d_aux.as<int64_t>(),
                                     S.d_aux2.as<int64_t>(), (size_t)n, 0, 64, s));
    if (int rc = S.d_tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::radix_sort_pairs(S.d_tmp.p, tb, S.d_hash.as<uint64_t>(), S.d_hash2.as<uint64_t>(), S.d_aux.as<int64_t>(),
                                     S.d_aux2.as<int64_t>(), (size_t)n, 0, 64, s));
    if (int rc = S.d_flag.ensure(4 * n)) return rc;
    hipLaunchKernelGGL(pair_flag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S.d_hash2.as<uint64_t>(), S.d_aux2.as<int64_t>(), n,
                       S.d_flag.as<int32_t>());
    if (int rc = S.d_uniq.ensure(8 * n)) return rc;
    if (int rc = S.d_cnt.ensure(4 * n)) return rc;
    if (int rc = S.d_n.ensure(8)) return rc;
    DG_HIP(rocprim::reduce_by_key(nullptr, tb, S.d_hash2.as<uint64_t>(), S.d_flag.as<int32_t>(), (size_t)n, S.d_uniq.as<uint64_t>(),
                                  S.d_cnt.as<int32_t>(), S.d_n.as<unsigned long long>(), rocprim::plus<int32_t>(),
                                  rocprim::equal_to<uint64_t>(), s));
    if (int rc = S.d_tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::reduce_by_key(S.d_tmp.p, tb, S.d_hash2.as<uint64_t>(), S.d_flag.as<int32_t>(), (size_t)n, S.d_uniq.as<uint64_t>(),
                                  S.d_cnt.as<int32_t>(), S.d_n.as<unsigned long long>(), rocprim::plus<int32_t>(),
                                  rocprim::equal_to<uint64_t>(), s));
    unsigned long long nd = 0;
    DG_HIP(hipMemcpyAsync(&nd, S.d_n.p, 8, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    *n_distinct = (int64_t)nd;
    return DG_OK;
}

static int check_kw(int k, int w) {
    if (k < 1 || k > 255 || w < 1 || w > 255) { set_error("k and w must be in 1..255 (k=%d w=%d)", k, w); return DG_ERR_ARG; }
    return DG_OK;
}

static int events(SketchState &S) {
    for (auto &e : S.ev) if (!e) DG_HIP(hipEventCreate(&e));
    return DG_OK;
}

// out_hash == nullptr: the spectrum stays in d_uniq / d_cnt (host API); otherwise it is written to the caller's device buffers.
// Route: minimizers straight into hash-range buckets (MODE 3) -> if a bucket runs over its stride, the exact two-pass
// placement from the sparse output -> if that is not possible either, the generic sort of all pairs.
static int sketch_reads_device(dg_ctx *c, const char *bases_dev, const int64_t *off_dev, int64_t n_reads, int k, int w, uint64_t *out_hash,
                               int32_t *out_cnt, int64_t cap, int64_t *n_distinct) {
    SketchState &S = state(c);
    hipStream_t s = c->stream;
    if (int rc = events(S)) return rc;
    DG_HIP(hipEventRecord(S.ev[0], s));
    int64_t n_emit = 0, nt = 0, n_win = 0;
    *n_distinct = 0;
    S.stat_overflow = 0; S.stat_buckets = 0; S.stat_path = 2;
    int64_t n_multi = 0;
    if (int rc = prepare_tiles(c, off_dev, n_reads, k, w, &nt, &n_win, &n_multi)) return rc;
    BucketPlan plan;
    bucket_plan(S, n_reads, nt, n_win, n_multi, w, &plan);
    bool done = nt == 0, ev1 = false;
    if (!done && plan.ok && S.opt_mode == 0 && !S.sticky_exact) {
        BucketEmit be;
        if (int rc = bucket_fast_begin(c, S, plan, &be)) return rc;
        if (int rc = launch_tiles_buckets(c, bases_dev, nt, k, w, be)) return rc;
        DG_HIP(hipEventRecord(S.ev[1], s)); ev1 = true;
        int outcome = 0;
        if (int rc = bucket_finish(c, S, plan, true, out_hash, out_cnt, cap, n_distinct, &n_emit, &outcome)) return rc;
        if (outcome == 0) { done = true; S.stat_path = 0; }
        else if (outcome == 1) S.sticky_exact = true;                  // this ctx's read sets have hashes too frequent for the stride
        else plan.ok = false;
    }
    if (!done && plan.ok && S.opt_mode != 1) {
        if (int rc = launch_tiles_sparse<false>(c, bases_dev, nt, n_win, k, w, 1)) return rc;
        if (!ev1) { DG_HIP(hipEventRecord(S.ev[1], s)); ev1 = true; }
        if (int rc = bucket_exact_scatter(c, S, plan, nt)) return rc;
        int outcome = 0;
        if (int rc = bucket_finish(c, S, plan, false, out_hash, out_cnt, cap, n_distinct, &n_emit, &outcome)) return rc;
        if (outcome == 0) { done = true; S.stat_path = 1; }
    }
    if (!done) {                                                       // generic path: stable 64-bit sort + run flags + reduce by key
        if (int rc = launch_tiles_sparse<false>(c, bases_dev, nt, n_win, k, w, 0)) return rc;
        if (!ev1) { DG_HIP(hipEventRecord(S.ev[1], s)); ev1 = true; }
        if (int rc = compact_tiles(c, nt, &n_emit)) return rc;
        if (int rc = spectrum_from_pairs(c, n_emit, n_distinct)) return rc;
        if (out_hash && *n_distinct && *n_distinct <= cap) {
            DG_HIP(hipMemcpyAsync(out_hash, S.d_uniq.p, 8 * (size_t)*n_distinct, hipMemcpyDeviceToDevice, s));
            DG_HIP(hipMemcpyAsync(out_cnt, S.d_cnt.p, 4 * (size_t)*n_distinct, hipMemcpyDeviceToDevice, s));
        }
    }
    if (!ev1) DG_HIP(hipEventRecord(S.ev[1], s));
    DG_HIP(hipEventRecord(S.ev[2], s));
    DG_HIP(hipStreamSynchronize(s));
    DG_HIP(hipEventElapsedTime(&S.timing.kernel_ms, S.ev[0], S.ev[1]));
    DG_HIP(hipEventElapsedTime(&S.timing.sort_ms, S.ev[1], S.ev[2]));
    DG_HIP(hipEventElapsedTime(&S.timing.total_ms, S.ev[0], S.ev[2]));
    S.timing.n_emitted = n_emit;
    return DG_OK;
}

}  // namespace dgi

using namespace dgi;

extern "C" int dg_sketch_reads(dg_ctx *c, const char *bases, const int64_t *read_off, int64_t n_reads, int k, int w,
                               uint64_t **hash, int32_t **cnt, int64_t *n_distinct) {
    if (int rc = bind(c)) return rc;
    if (int rc = check_kw(k, w)) return rc;
    if (!read_off || !hash || !cnt || !n_distinct || n_reads < 0) { set_error("dg_sketch_reads: bad arguments"); return DG_ERR_ARG; }
    SketchState &S = state(c);
    const int64_t nb = n_reads ? read_off[n_reads] : 0;
    if (int rc = S.d_bases.ensure((size_t)nb + 16)) return rc;
    if (nb) DG_HIP(hipMemcpyAsync(S.d_bases.p, bases, (size_t)nb, hipMemcpyHostToDevice, c->stream));
    if (int rc = S.d_off.ensure(8 * (size_t)(n_reads + 1))) return rc;
    if (n_reads) DG_HIP(hipMemcpyAsync(S.d_off.p, read_off, 8 * (size_t)(n_reads + 1), hipMemcpyHostToDevice, c->stream));
    int64_t nd = 0;
    if (int rc = sketch_reads_device(c, S.d_bases.as<char>(), S.d_off.as<int64_t>(), n_reads, k, w, nullptr, nullptr, 0, &nd)) return rc;
    *hash = (uint64_t *)malloc(8 * (size_t)(nd + 1));
    *cnt = (int32_t *)malloc(4 * (size_t)(nd + 1));
    if (!*hash || !*cnt) { set_error("host malloc failed"); return DG_ERR_OOM; }
    if (nd) {
        DG_HIP(hipMemcpyAsync(*hash, S.d_uniq.p, 8 * (size_t)nd, hipMemcpyDeviceToHost, c->stream));
        DG_HIP(hipMemcpyAsync(*cnt, S.d_cnt.p, 4 * (size_t)nd, hipMemcpyDeviceToHost, c->stream));
        DG_HIP(hipStreamSynchronize(c->stream));
    }
    *n_distinct = nd;
    return DG_OK;
}

extern "C" int dg_sketch_reads_dev(dg_ctx *c, const char *bases_dev, const int64_t *read_off_dev, int64_t n_reads, int64_t n_bases,
                                   int k, int w, uint64_t *hash_dev, int32_t *count_dev, int64_t cap, int64_t *n_distinct) {
    if (int rc = bind(c)) return rc;
    if (int rc = check_kw(k, w)) return rc;
    if (!bases_dev || !read_off_dev || !n_distinct || n_reads < 0) { set_error("dg_sketch_reads_dev: bad arguments"); return DG_ERR_ARG; }
    (void)n_bases;
    int64_t nd = 0;
    if (!hash_dev || !count_dev || cap < 0) { set_error("dg_sketch_reads_dev: bad output arguments"); return DG_ERR_ARG; }
    if (int rc = sketch_reads_device(c, bases_dev, read_off_dev, n_reads, k, w, hash_dev, count_dev, cap, &nd)) return rc;
    if (nd > cap) { set_error("dg_sketch_reads_dev: %lld distinct hashes exceed capacity %lld", (long long)nd, (long long)cap); return DG_ERR_ARG; }
    *n_distinct = nd;
    return DG_OK;
}

extern "C" int dg_sketch_set_option(dg_ctx *c, const char *name, int64_t value) {
    if (int rc = bind(c)) return rc;
    if (!name) { set_error("dg_sketch_set_option: null name"); return DG_ERR_ARG; }
    SketchState &S = state(c);
    const std::string n(name);
    if (n == "spectrum_mode") { if (value < 0 || value > 2) { set_error("spectrum_mode must be 0, 1 or 2"); return DG_ERR_ARG; } S.opt_mode = (int)value; S.sticky_exact = false; }
    else if (n == "bucket_bits") { if (value < 0 || value > 15) { set_error("bucket_bits must be 0 (automatic) .. 15"); return DG_ERR_ARG; } S.opt_bucket_bits = (int)value; }
    else if (n == "bucket_stride") { if (value < 0 || value > (1 << 20)) { set_error("bucket_stride out of range"); return DG_ERR_ARG; } S.opt_stride = (int)value; S.sticky_exact = false; }
    else if (n == "host_buckets") { if (value < 0 || value > 256) { set_error("host_buckets must be 0 (default 256) .. 256"); return DG_ERR_ARG; } S.opt_host_buckets = (int)value; }
    else if (n == "spill_cap") { if (value < -1 || value > ((int64_t)1 << 28)) { set_error("spill_cap must be -1 (none) .. 2^28, 0 = default"); return DG_ERR_ARG; } S.opt_spill_cap = value; S.sticky_exact = false; }
    else if (n == "residual_cap") { if (value < -1 || value > 1024) { set_error("residual_cap must be -1 (none) .. 1024, 0 = default"); return DG_ERR_ARG; } S.opt_residual_cap = (int)value; }
    else { set_error("dg_sketch_set_option: unknown option '%s'", name); return DG_ERR_ARG; }
    return DG_OK;
}

extern "C" int dg_sketch_get_stat(dg_ctx *c, const char *name, int64_t *value) {
    if (!c || !c->sk || !name || !value) { set_error("dg_sketch_get_stat: no state"); return DG_ERR_STATE; }
    const SketchState &S = *c->sk;
    const std::string n(name);
    if (n == "spectrum_path") *value = S.stat_path;                    // 0 buckets filled by the tile kernel, 1 buckets placed from the sparse output, 2 generic sort
    else if (n == "buckets") *value = S.stat_buckets;
    else if (n == "overflow_buckets") *value = S.stat_overflow;
    else if (n == "spilled_pairs") *value = S.stat_spilled;
    else { set_error("dg_sketch_get_stat: unknown name '%s'", name); return DG_ERR_ARG; }
    return DG_OK;
}

// minimizer list of one haplotype, left on the device: (hash, position of the winning k-mer) in sequence order.  The
// pointers stay valid until the next sketch call on this ctx.  kernel_ms is recorded in the ctx's sketch timing.
namespace dgi {
int sketch_haplotype_dev(dg_ctx *c, const char *seq, int64_t len, int k, int w, const uint64_t **hash_dev, const int64_t **pos_dev, int64_t *n) {
    if (int rc = check_kw(k, w)) return rc;
    if (len < 0) { set_error("sketch_haplotype: negative length"); return DG_ERR_ARG; }
    if (len >= ((int64_t)1 << 31)) { set_error("sequence longer than 2^31 bases is not supported"); return DG_ERR_UNSUPPORTED; }
    SketchState &S = state(c);
    if (int rc = events(S)) return rc;
    if (int rc = S.d_bases.ensure((size_t)len + 16)) return rc;
    if (len) DG_HIP(hipMemcpyAsync(S.d_bases.p, seq, (size_t)len, hipMemcpyHostToDevice, c->stream));
    const int64_t off[2] = {0, len};
    if (int rc = S.d_off.ensure(16)) return rc;
    DG_HIP(hipMemcpyAsync(S.d_off.p, off, 16, hipMemcpyHostToDevice, c->stream));
    DG_HIP(hipEventRecord(S.ev[0], c->stream));
    int64_t ne = 0;
    if (int rc = run_tiles<true>(c, S.d_bases.as<char>(), S.d_off.as<int64_t>(), 1, k, w, &ne)) return rc;
    DG_HIP(hipEventRecord(S.ev[1], c->stream));
    DG_HIP(hipStreamSynchronize(c->stream));
    DG_HIP(hipEventElapsedTime(&S.timing.kernel_ms, S.ev[0], S.ev[1]));
    S.timing.sort_ms = 0; S.timing.total_ms = S.timing.kernel_ms; S.timing.n_emitted = ne;
    *hash_dev = S.d_hash.as<uint64_t>(); *pos_dev = S.d_aux.as<int64_t>(); *n = ne;
    return DG_OK;
}
}  // namespace dgi

extern "C" int dg_sketch_haplotype(dg_ctx *c, const char *seq, int64_t len, int k, int w, uint64_t **hash, int64_t **pos, int64_t *n) {
    if (int rc = bind(c)) return rc;
    if (!hash || !pos || !n) { set_error("dg_sketch_haplotype: bad arguments"); return DG_ERR_ARG; }
    const uint64_t *hd = nullptr; const int64_t *pd = nullptr; int64_t ne = 0;
    if (int rc = sketch_haplotype_dev(c, seq, len, k, w, &hd, &pd, &ne)) return rc;
    *hash = (uint64_t *)malloc(8 * (size_t)(ne + 1));
    *pos = (int64_t *)malloc(8 * (size_t)(ne + 1));
    if (!*hash || !*pos) { set_error("host malloc failed"); return DG_ERR_OOM; }
    if (ne) {
        DG_HIP(hipMemcpyAsync(*hash, hd, 8 * (size_t)ne, hipMemcpyDeviceToHost, c->stream));
        DG_HIP(hipMemcpyAsync(*pos, pd, 8 * (size_t)ne, hipMemcpyDeviceToHost, c->stream));
        DG_HIP(hipStreamSynchronize(c->stream));
    }
    *n = ne;
    return DG_OK;
}

extern "C" int dg_hash_kmers(dg_ctx *c, const char *kmers, int64_t n, int k, uint64_t *out) {
    if (int rc = bind(c)) return rc;
    if (k < 1 || k > 255 || n < 0 || !out) { set_error("dg_hash_kmers: bad arguments"); return DG_ERR_ARG; }
    if (n == 0) return DG_OK;
    SketchState &S = state(c);
    if (int rc = S.d_kmers.ensure((size_t)n * k)) return rc;
    if (int rc = S.d_out.ensure(8 * (size_t)n)) return rc;
    DG_HIP(hipMemcpyAsync(S.d_kmers.p, kmers, (size_t)n * k, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(hash_kmers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, S.d_kmers.as<char>(), n, k, S.d_out.as<uint64_t>());
    DG_HIP(hipMemcpyAsync(out, S.d_out.p, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    DG_HIP(hipStreamSynchronize(c->stream));
    return DG_OK;
}

extern "C" int dg_sketch_get_timing(dg_ctx *c, dg_sketch_timing *t) {
    if (!c || !c->sk || !t) { set_error("dg_sketch_get_timing: no state"); return DG_ERR_STATE; }
    *t = c->sk->timing;
    return DG_OK;
}

extern "C" int dg_sketch_count_dictionary_dev(dg_ctx *c, const uint64_t *dict_dev, int64_t n_dict, const uint64_t *hash_dev,
                                              const int32_t *count_dev, int64_t n, int32_t *counts_dev) {
    if (int rc = bind(c)) return rc;
    if (n_dict <= 0) return DG_OK;
    hipLaunchKernelGGL(dict_count_kernel, dim3((unsigned)((n_dict + 255) / 256)), dim3(256), 0, c->stream, dict_dev, n_dict, hash_dev, count_dev, n, counts_dev);
    DG_HIP(hipGetLastError());
    return DG_OK;
}

extern "C" int dg_sketch_merge_runs_dev(dg_ctx *c, const uint64_t *hash_dev, const int32_t *count_dev, int64_t n, uint64_t *out_hash_dev,
                                        int32_t *out_count_dev, int64_t cap, int64_t *n_out) {
    if (int rc = bind(c)) return rc;
    if (!n_out) { set_error("dg_sketch_merge_runs_dev: null n_out"); return DG_ERR_ARG; }
    *n_out = 0;
    if (n == 0) return DG_OK;
    SketchState &S = state(c);
    hipStream_t s = c->stream;
    if (int rc = S.d_hash2.ensure(8 * n)) return rc;
    if (int rc = S.d_flag.ensure(4 * n)) return rc;
    size_t tb = 0;
    DG_HIP(rocprim::radix_sort_pairs(nullptr, tb, hash_dev, S.d_hash2.as<uint64_t>(), count_dev, S.d_flag.as<int32_t>(), (size_t)n, 0, 64, s));
    if (int rc = S.d_tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::radix_sort_pairs(S.d_tmp.p, tb, hash_dev, S.d_hash2.as<uint64_t>(), count_dev, S.d_flag.as<int32_t>(), (size_t)n, 0, 64, s));
    if (int rc = S.d_uniq.ensure(8 * n)) return rc;
    if (int rc = S.d_cnt.ensure(4 * n)) return rc;
    if (int rc = S.d_n.ensure(8)) return rc;
    DG_HIP(rocprim::reduce_by_key(nullptr, tb, S.d_hash2.as<uint64_t>(), S.d_flag.as<int32_t>(), (size_t)n, S.d_uniq.as<uint64_t>(),
                                  S.d_cnt.as<int32_t>(), S.d_n.as<unsigned long long>(), rocprim::plus<int32_t>(), rocprim::equal_to<uint64_t>(), s));
    if (int rc = S.d_tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::reduce_by_key(S.d_tmp.p, tb, S.d_hash2.as<uint64_t>(), S.d_flag.as<int32_t>(), (size_t)n, S.d_uniq.as<uint64_t>(),
                                  S.d_cnt.as<int32_t>(), S.d_n.as<unsigned long long>(), rocprim::plus<int32_t>(), rocprim::equal_to<uint64_t>(), s));
    unsigned long long nd = 0;
    DG_HIP(hipMemcpyAsync(&nd, S.d_n.p, 8, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    if (nd > 0) {                                          // padding of a fixed-size exchange: (0xFFFF...F, 0) entries collapse into one last entry of count 0
        unsigned long long last_h = 0; int32_t last_c = 1;
        DG_HIP(hipMemcpyAsync(&last_h, S.d_uniq.as<uint64_t>() + (nd - 1), 8, hipMemcpyDeviceToHost, s));
        DG_HIP(hipMemcpyAsync(&last_c, S.d_cnt.as<int32_t>() + (nd - 1), 4, hipMemcpyDeviceToHost, s));
        DG_HIP(hipStreamSynchronize(s));
        if (last_h == ~0ULL && last_c == 0) --nd;
    }
    if ((int64_t)nd > cap) { set_error("merge: %llu distinct exceed capacity %lld", nd, (long long)cap); return DG_ERR_ARG; }
    DG_HIP(hipMemcpyAsync(out_hash_dev, S.d_uniq.p, 8 * (size_t)nd, hipMemcpyDeviceToDevice, s));
    DG_HIP(hipMemcpyAsync(out_count_dev, S.d_cnt.p, 4 * (size_t)nd, hipMemcpyDeviceToDevice, s));
    DG_HIP(hipStreamSynchronize(s));
    *n_out = (int64_t)nd;
    return DG_OK;
}

extern "C" int dg_sketch_partition_dev(dg_ctx *c, const uint64_t *hash_dev, int64_t n, int world, int64_t *split_dev) {
    if (int rc = bind(c)) return rc;
    if (world < 1 || world > 4096 || !split_dev || n < 0) { set_error("dg_sketch_partition_dev: bad arguments"); return DG_ERR_ARG; }
    hipLaunchKernelGGL(partition_kernel, dim3((unsigned)((world + 1 + 63) / 64)), dim3(64), 0, c->stream, hash_dev, n, world, split_dev);
    DG_HIP(hipGetLastError());
    return DG_OK;
}

extern "C" int dg_sketch_rank_dictionary_dev(dg_ctx *c, const uint64_t *dict_dev, int64_t n_dict, const uint64_t *hash_dev, int64_t n, int64_t base,
                                             int64_t *rank1_dev) {
    if (int rc = bind(c)) return rc;
    if (n_dict <= 0) return DG_OK;
    hipLaunchKernelGGL(dict_rank_kernel, dim3((unsigned)((n_dict + 255) / 256)), dim3(256), 0, c->stream, dict_dev, n_dict, hash_dev, n, base, rank1_dev);
    DG_HIP(hipGetLastError());
    return DG_OK;
}

extern "C" int dg_sketch_count_rank_dictionary_dev(dg_ctx *c, const uint64_t *dict_dev, int64_t n_dict, const uint64_t *hash_dev,
                                                   const int32_t *count_dev, int64_t n, int64_t base, int32_t *counts_dev, int64_t *rank1_dev) {
    if (int rc = bind(c)) return rc;
    if (n_dict <= 0) return DG_OK;
    hipLaunchKernelGGL(dict_count_rank_kernel, dim3((unsigned)((n_dict + 255) / 256)), dim3(256), 0, c->stream, dict_dev, n_dict, hash_dev, count_dev, n, base,
                       counts_dev, rank1_dev);
    DG_HIP(hipGetLastError());
    return DG_OK;
}

extern "C" int dg_sketch_histogram_dev(dg_ctx *c, const int32_t *count_dev, int64_t n, int n_bins, uint64_t *hist_dev) {
    if (int rc = bind(c)) return rc;
    if (n_bins < 2 || !hist_dev) { set_error("dg_sketch_histogram_dev: bad arguments"); return DG_ERR_ARG; }
    if (n <= 0) return DG_OK;
    hipLaunchKernelGGL(mult_hist_kernel, dim3((unsigned)std::min<int64_t>((n + 8191) / 8192, 256)), dim3(256), 0, c->stream, count_dev, n, n_bins, (unsigned long long *)hist_dev);
    DG_HIP(hipGetLastError());
    return DG_OK;
}
