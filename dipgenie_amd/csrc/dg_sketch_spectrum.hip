// libdipgenie_hip.so -- the read spectrum (Sp_R) from the tile kernel's minimizers, bucketed by hash range (gfx950).
//
// What it computes is the reference's Sp_R / kmer_count maps (/root/reference/src/solver.cpp:526-546, 711-732): every
// distinct minimizer hash of the read set with the number of DISTINCT reads that hold it.  dg_sketch.hip's generic path gets
// there with a 64-bit stable radix sort of all (hash, read) pairs (8 onesweep passes over 16-byte pairs) + a run flag +
// reduce-by-key.  MurmurHash3 output is uniform, so this path treats the hash as an address instead of sorting it:
//
//   buckets   the top `bbits` bits of the hash choose one of B buckets (~2-3 k pairs each).  The tile kernel drops every
//             minimizer straight into its bucket (slot = fill[bucket]++, B * stride slots; sketch_tile_kernel MODE 3), already
//             free of repeats of a hash inside one tile.  A full bucket (a hash held by > ~10 k reads) sends its further pairs to one
//             shared spill list, which its table workgroup looks through; only if that list runs over too is the pass repeated with
//             exact placement from the sparse tile output (hist -> scan -> basefill -> scatter below): any bucket size fits then.
//   table     one workgroup per bucket.  The next `sbits` bits address a table in LDS with TWO hashes per entry, the
//             smallest and the largest of the sub-bucket (64-bit LDS atomic min / max), and a count for each: B * 2^sbits
//             entries for ~10^6 distinct hashes, so a third hash in an entry is rare (~1 % of the hashes).  Pairs are never
//             stored or sorted: two passes over the bucket, a handful of LDS atomics per pair, whatever the multiplicity of
//             a hash.  Third hashes, and pairs of reads longer than one tile (the same (hash, read) may come from two
//             tiles), go to a small residual list that is resolved quadratically (<= 1024 entries; more: the host finishes
//             that bucket's segment with rocPRIM).
//   output    entries in table order ARE the sorted distinct hashes of the bucket (min, third hashes by rank, max); an
//             exclusive scan over the table gives their places; dscan + gather concatenate the buckets (hash ranges).
//
// The result is a pure function of the multiset of pairs, so it equals the generic path's bit for bit.
#include <cstring>
#include <algorithm>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dg_sketch.hpp"

namespace dgi {

namespace {

constexpr int BT = 1024;                // lanes of a hist / scatter / table workgroup
constexpr int STRIDE = 12288;           // slots per bucket when the tile kernel fills the buckets
constexpr int RCAP = 1024;              // residual entries per bucket (one per lane)
constexpr int SBITS_MIN = 12, SETCAP = 16384;   // hash-set slots of the multi-tile repeat filter at 2^12 table entries (64 of the tables' 112 KB)
constexpr int SBITS = 12;               // table entries per bucket = 2^SBITS (112 KB of LDS: one workgroup per CU)
constexpr int SPILL_CAP = 1 << 20;        // pairs the shared spill list of full buckets holds (12 MB)
constexpr int OVF_MAX = 256;            // buckets finished by the host one by one; more -> generic path
constexpr int G_MAX = 256;              // hist / scatter workgroups (one per CU)

// exclusive scan of one value per lane over the workgroup (blockDim.x a multiple of 64, <= 1024); scr: >= 17 entries of T
template <class T>
__device__ __forceinline__ T block_excl_scan(T v, T *scr, T *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();                                                   // scr may still be read from a previous call
    if (lane == 63) scr[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        T x = lane < nw ? scr[lane] : (T)0, xi = x;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const T u = __shfl_up(xi, o);
            if (lane >= o) xi += u;
        }
        if (lane < nw) scr[lane] = xi - x;
        if (lane == nw - 1) scr[16] = xi;
    }
    __syncthreads();
    *total = scr[16];
    return scr[wave] + inc - v;
}

// the tiles of workgroup g: an equal share of the tile list
__device__ __forceinline__ void tile_share(int64_t n_tiles, int g, int G, int64_t *t0, int64_t *t1) {
    const int64_t per = (n_tiles + G - 1) / G;
    *t0 = min(n_tiles, (int64_t)g * per);
    *t1 = min(n_tiles, *t0 + per);
}

// row g of the matrix = this workgroup's pairs per bucket; bucket totals by atomics (G * B adds, no return value)
__global__ __launch_bounds__(BT) void bk_hist_kernel(const int64_t *__restrict__ sparse, const int64_t *__restrict__ cnt, int64_t n_tiles,
                                                     const uint64_t *__restrict__ sh, int bbits, uint32_t *__restrict__ matrix,
                                                     uint32_t *__restrict__ bucket_cnt) {
    extern __shared__ uint32_t hist[];
    const int B = 1 << bbits;
    for (int b = threadIdx.x; b < B; b += BT) hist[b] = 0;
    __syncthreads();
    int64_t t0, t1;
    tile_share(n_tiles, blockIdx.x, gridDim.x, &t0, &t1);
    const int sub = threadIdx.x & 7;
    for (int64_t t = t0 + (threadIdx.x >> 3); t < t1; t += BT / 8) {      // 8 lanes per tile
        const int64_t sb = sparse[t], n = cnt[t];
        for (int64_t i = sub; i < n; i += 8) atomicAdd(&hist[sh[sb + i] >> (64 - bbits)], 1u);
    }
    __syncthreads();
    uint32_t *row = matrix + (size_t)blockIdx.x * B;
    for (int b = threadIdx.x; b < B; b += BT) {
        const uint32_t v = hist[b];
        row[b] = v;
        if (v) atomicAdd(&bucket_cnt[b], v);
    }
}

// exclusive scan of n (<= 32 * 1024) counters by one workgroup: out[0..n], out[n] = total
__global__ __launch_bounds__(BT) void bk_scan_kernel(const uint32_t *__restrict__ in, int n, uint32_t *__restrict__ out) {
    __shared__ uint32_t scr[17];
    const int per = (n + BT - 1) / BT, i0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (int i = i0; i < min(n, i0 + per); ++i) sum += in[i];
    uint32_t total, run = block_excl_scan(sum, scr, &total);
    for (int i = i0; i < min(n, i0 + per); ++i) { const uint32_t v = in[i]; out[i] = run; run += v; }
    if (threadIdx.x == 0) out[n] = total;
}

// matrix[g][b] := bucket_start[b] + pairs of bucket b held by workgroups < g
__global__ void bk_basefill_kernel(uint32_t *__restrict__ matrix, const uint32_t *__restrict__ bucket_start, int B, int G) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t run = bucket_start[b];
    for (int g = 0; g < G; ++g) { const uint32_t v = matrix[(size_t)g * B + b]; matrix[(size_t)g * B + b] = run; run += v; }
}

__global__ __launch_bounds__(BT) void bk_scatter_kernel(const int64_t *__restrict__ sparse, const int64_t *__restrict__ cnt, int64_t n_tiles,
                                                        const uint64_t *__restrict__ sh, const int64_t *__restrict__ sa, int bbits,
                                                        const uint32_t *__restrict__ matrix, uint64_t *__restrict__ bk_hash,
                                                        uint32_t *__restrict__ bk_read) {
    extern __shared__ uint32_t cur[];
    const int B = 1 << bbits;
    const uint32_t *row = matrix + (size_t)blockIdx.x * B;
    for (int b = threadIdx.x; b < B; b += BT) cur[b] = row[b];
    __syncthreads();
    int64_t t0, t1;
    tile_share(n_tiles, blockIdx.x, gridDim.x, &t0, &t1);
    const int sub = threadIdx.x & 7;
    for (int64_t t = t0 + (threadIdx.x >> 3); t < t1; t += BT / 8) {
        const int64_t sb = sparse[t], n = cnt[t];
        for (int64_t i = sub; i < n; i += 8) {
            const uint64_t h = sh[sb + i];
            const uint32_t pos = atomicAdd(&cur[h >> (64 - bbits)], 1u);
            bk_hash[pos] = h;
            bk_read[pos] = (uint32_t)sa[sb + i];
        }
    }
}

// One bucket: its sorted distinct hashes and distinct-read counts, written over the bucket's own first entries (every pair
// has been read by then).  fast: bucket b = fill[b] pairs at b * stride; otherwise pairs [start[b], start[b + 1]).
// LDS: mn, mx u64[nS] | cm, cM, nd u32[nS] | residual entries (low u64, sub u32, read u32)[RCAP] | flags u8[RCAP] | scratch
struct Residual { uint64_t low; uint32_t sub, read; };
__global__ __launch_bounds__(BT) void bk_table_kernel(uint64_t *bk_hash, uint32_t *bk_read, const uint32_t *__restrict__ start,
                                                      const uint32_t *__restrict__ fill, uint32_t stride, int bbits, int sbits, uint32_t residual_cap,
                                                      int has_multi, uint32_t *__restrict__ dcount, uint32_t *__restrict__ ovf, const uint32_t *__restrict__ spill_n,
                                                      const uint64_t *spill_hash, uint32_t *spill_read, uint32_t spill_cap) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int nS = 1 << sbits, tid = threadIdx.x, b = blockIdx.x;
    unsigned long long *mn = (unsigned long long *)lds_raw, *mx = mn + nS;
    uint32_t *cm = (uint32_t *)(mx + nS), *cM = cm + nS, *nd = cM + nS;
    Residual *res = (Residual *)(nd + nS);
    uint8_t *rfl = (uint8_t *)(res + RCAP);
    uint32_t *scr = (uint32_t *)(rfl + RCAP);                          // [0..16] scan, [20] residual count, [21] hash-set entries
    size_t s;
    uint32_t n, n_sp = 0;                                               // pairs in the bucket's own slots; entries of the spill list to look through
    if (stride) {
        s = (size_t)b * stride; n = fill[(size_t)b * FILL_PAD];
        if (n > stride) {                                               // the rest of the bucket is in the shared spill list, among other buckets' pairs
            n = stride; n_sp = spill_n[0];
            if (n_sp > spill_cap) { if (tid == 0) { dcount[b] = 0; atomicAdd(&ovf[1], 1u); } return; }   // the list ran over too: exact placement
        }
    } else { s = start[b]; n = start[b + 1] - start[b]; }
    if (n == 0) { if (tid == 0) dcount[b] = 0; return; }
    const int lowbits = 64 - bbits - sbits;
    const unsigned long long lowmask = (1ULL << lowbits) - 1ULL;
    const uint32_t n_all = n + n_sp;
    // pair i of the bucket: its own slots first, then the spill list (entries of other buckets: false)
    auto pair = [&](uint32_t i, unsigned long long &h, uint32_t &r) -> bool {
        if (i < n) { h = bk_hash[s + i]; r = bk_read[s + i]; return true; }
        h = spill_hash[i - n];
        if ((uint32_t)(h >> (64 - bbits)) != (uint32_t)b) return false;
        r = spill_read[i - n];
        return true;
    };
    auto give_up = [&](bool to_host) {                                  // nothing of the bucket has been overwritten
        if (tid == 0) {
            dcount[b] = 0;
            if (to_host && !n_sp) { const uint32_t o = atomicAdd(&ovf[0], 1u); if (o < OVF_MAX) ovf[2 + o] = (uint32_t)b; }   // the host finishes this segment
            else atomicAdd(&ovf[1], 1u);                               // (a bucket with spilled pairs is not one segment: exact placement)
        }
    };
    if (has_multi) {
        // Pairs of reads longer than one tile (bit 31 of the read id): the same (hash, read) may have come from two tiles.  An
        // LDS hash set of pair indices keeps one of each and clears its bit; the repeats keep theirs and are skipped below.
        // (The set lives where the tables go afterwards.)
        uint32_t *set = (uint32_t *)lds_raw;
        const uint32_t cap = SETCAP << (sbits - SBITS_MIN), EMPTY = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < cap; i += BT) set[i] = EMPTY;
        if (tid == 0) scr[21] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n_all; i += BT) {
            unsigned long long h; uint32_t r;
            if (!pair(i, h, r) || !(r >> 31)) continue;
            unsigned long long m = (h ^ ((unsigned long long)(r & 0x7FFFFFFFu) * 0x9E3779B97F4A7C15ULL));
            m ^= m >> 29; m *= 0xBF58476D1CE4E5B9ULL; m ^= m >> 32;
            uint32_t slot = (uint32_t)m & (cap - 1);
            for (uint32_t probe = 0; probe < cap; ++probe) {
                const uint32_t old = atomicCAS(&set[slot], EMPTY, i);
                if (old == EMPTY) { if (i < n) bk_read[s + i] = r & 0x7FFFFFFFu; else spill_read[i - n] = r & 0x7FFFFFFFu; atomicAdd(&scr[21], 1u); break; }
                unsigned long long ho; uint32_t ro;
                pair(old, ho, ro);                                      // (its owner may be clearing bit 31 right now: masked, so either value compares the same)
                if (ho == h && ((ro ^ r) & 0x7FFFFFFFu) == 0) break;
                slot = (slot + 1) & (cap - 1);
            }
        }
        __syncthreads();
        if (scr[21] > cap / 2 + cap / 4) { give_up(true); return; }     // the set got too full to trust the probe bound
    }
    for (int i = tid; i < nS; i += BT) { mn[i] = ~0ULL; mx[i] = 0; cm[i] = 0; cM[i] = 0; nd[i] = 0; }
    if (tid == 0) scr[20] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n_all; i += BT) {
        unsigned long long h; uint32_t r;
        if (!pair(i, h, r)) continue;
        const unsigned long long low = h & lowmask;
        const uint32_t sub = (uint32_t)(h >> lowbits) & (uint32_t)(nS - 1);
        atomicMin(&mn[sub], low);
        atomicMax(&mx[sub], low);
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_all; i += BT) {
        unsigned long long h; uint32_t r;
        if (!pair(i, h, r) || (r >> 31)) continue;                     // (bit 31 still set: a repeat of a (hash, read) counted elsewhere)
        const unsigned long long low = h & lowmask;
        const uint32_t sub = (uint32_t)(h >> lowbits) & (uint32_t)(nS - 1);
        if (low == mn[sub]) atomicAdd(&cm[sub], 1u);
        else if (low == mx[sub]) atomicAdd(&cM[sub], 1u);
        else {
            const uint32_t x = atomicAdd(&scr[20], 1u);
            if (x < residual_cap) res[x] = Residual{low, sub, r};
        }
    }
    __syncthreads();
    const uint32_t R = scr[20];
    if (R > residual_cap) { give_up(true); return; }
    // residual entry of this lane = a third hash of its table entry: the first of its hash speaks for it (count, rank)
    Residual me{0, 0, 0};
    uint32_t third_cnt = 0, third_rank = 0;
    bool third = false;
    if (R) {
        if ((uint32_t)tid < R) {
            me = res[tid];
            bool first = true;
            for (uint32_t j = 0; j < R; ++j)
                if (res[j].sub == me.sub && res[j].low == me.low) { if (j < (uint32_t)tid) first = false; ++third_cnt; }
            if (first) { third = true; atomicAdd(&nd[me.sub], 1u); }
            rfl[tid] = first ? 1 : 0;
        }
        __syncthreads();
        if (third)
            for (uint32_t j = 0; j < R; ++j) if (rfl[j] && res[j].sub == me.sub && res[j].low < me.low) ++third_rank;
    }
    // places: entry `sub` holds (non-empty) + (max differs from min) + third hashes; exclusive scan over the table
    const int per = nS / BT, i0 = tid * per;                           // nS >= BT
    uint32_t d[8], sum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        d[q] = 0;
        if (q < per) { const int sub = i0 + q; if (mn[sub] != ~0ULL) d[q] = 1 + (mx[sub] != mn[sub] ? 1 : 0) + nd[sub]; }
        sum += d[q];
    }
    uint32_t total, run = block_excl_scan(sum, scr, &total);
    const unsigned long long top = (unsigned long long)b << (64 - bbits);
    __syncthreads();                                                   // every nd[] has been read
    if (stride && total > stride) { give_up(false); return; }          // more distinct hashes than the bucket has slots for its results
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q < per) {
        const int sub = i0 + q;
        nd[sub] = run;
        if (d[q]) {
            const unsigned long long hs = top | ((unsigned long long)sub << lowbits);
            bk_hash[s + run] = hs | mn[sub];
            ((int32_t *)bk_read)[s + run] = (int32_t)cm[sub];
            if (mx[sub] != mn[sub]) { bk_hash[s + run + d[q] - 1] = hs | mx[sub]; ((int32_t *)bk_read)[s + run + d[q] - 1] = (int32_t)cM[sub]; }
        }
        run += d[q];
    }
    __syncthreads();
    if (third) {
        const size_t at = s + nd[me.sub] + 1 + third_rank;
        bk_hash[at] = top | ((unsigned long long)me.sub << lowbits) | me.low;
        ((int32_t *)bk_read)[at] = (int32_t)third_cnt;
    }
    if (tid == 0) dcount[b] = total;
}

// exclusive scan of the buckets' distinct counts + the run's status = {pairs, distinct hashes, buckets left to the host,
// buckets that need exact placement, pairs sent to the spill list}
__global__ __launch_bounds__(BT) void bk_dscan_kernel(const uint32_t *__restrict__ dcount, const uint32_t *__restrict__ fill, const uint32_t *__restrict__ start,
                                                      int B, uint32_t *__restrict__ dstart, const uint32_t *__restrict__ ovf, const uint32_t *__restrict__ spill_n, int64_t *__restrict__ status) {
    __shared__ uint32_t scr[17];
    __shared__ unsigned long long pairs;
    if (threadIdx.x == 0) pairs = 0;
    const int per = (B + BT - 1) / BT, i0 = threadIdx.x * per;
    uint32_t sum = 0;
    unsigned long long np = 0;
    for (int i = i0; i < min(B, i0 + per); ++i) { sum += dcount[i]; if (fill) np += fill[(size_t)i * FILL_PAD]; }
    uint32_t total, run = block_excl_scan(sum, scr, &total);
    for (int i = i0; i < min(B, i0 + per); ++i) { const uint32_t v = dcount[i]; dstart[i] = run; run += v; }
    if (fill && np) atomicAdd(&pairs, np);
    __syncthreads();
    if (threadIdx.x == 0) {
        dstart[B] = total;
        status[0] = fill ? (int64_t)pairs : (int64_t)start[B];
        status[1] = total;
        status[2] = ovf[0];
        status[3] = ovf[1];
        status[4] = spill_n ? (int64_t)spill_n[0] : 0;
    }
}

// bucket-local results -> dense output
__global__ __launch_bounds__(256) void bk_gather_kernel(const uint32_t *__restrict__ start, uint32_t stride, const uint32_t *__restrict__ dcount,
                                                        const uint32_t *__restrict__ dstart, const uint64_t *__restrict__ bk_hash,
                                                        const uint32_t *__restrict__ bk_read, uint64_t *__restrict__ out_hash, int32_t *__restrict__ out_cnt, int64_t cap) {
    const int b = blockIdx.x;
    const size_t s = stride ? (size_t)b * stride : (size_t)start[b];
    const uint32_t d0 = dstart[b], D = dcount[b];
    for (uint32_t j = threadIdx.x; j < D; j += 256)
        if ((int64_t)(d0 + j) < cap) { out_hash[d0 + j] = bk_hash[s + j]; out_cnt[d0 + j] = (int32_t)bk_read[s + j]; }
}

__global__ void bk_pair_flag_kernel(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ read, int64_t n, int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || hash[i] != hash[i - 1] || ((read[i] ^ read[i - 1]) & 0x7FFFFFFFu)) ? 1 : 0;   // bit 31: multi-tile mark, not part of the id
}
__global__ void bk_set_dcount_kernel(uint32_t *dcount, int b, const unsigned long long *n) { dcount[b] = (uint32_t)*n; }

int bits_for(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

size_t table_lds_bytes(int sbits) { return (size_t)(1 << sbits) * (8 + 8 + 4 + 4 + 4) + sizeof(Residual) * RCAP + RCAP + 4 * 24; }

// one bucket [s, s + n) left to the host: stable sort by read, then by hash, run flags, reduce by key -- results over the
// bucket's own first entries, like the table kernel's
int finish_segment(dg_ctx *c, SketchState &S, int b, size_t s, size_t n) {
    hipStream_t st = c->stream;
    if (int rc = S.d_hash.ensure(8 * n)) return rc;                    // hashes carried by the read sort
    if (int rc = S.d_aux.ensure(8 * n)) return rc;                     // [0, 4n) reads sorted, [4n, 8n) reads after the hash sort
    if (int rc = S.d_uniq.ensure(8 * n)) return rc;                    // hashes sorted
    if (int rc = S.d_flag.ensure(4 * n)) return rc;
    if (int rc = S.d_n.ensure(8)) return rc;
    uint64_t *bh = S.d_bk_hash.as<uint64_t>() + s;
    uint32_t *br = S.d_bk_read.as<uint32_t>() + s;
    uint32_t *r1 = S.d_aux.as<uint32_t>(), *r2 = r1 + n;
    uint64_t *h1 = S.d_hash.as<uint64_t>(), *h2 = S.d_uniq.as<uint64_t>();
    size_t tb = 0, tb2 = 0, tb3 = 0;
    DG_HIP(rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t *)br, r1, (const uint64_t *)bh, h1, n, 0, 31, st));
    DG_HIP(rocprim::radix_sort_pairs(nullptr, tb2, (const uint64_t *)h1, h2, (const uint32_t *)r1, r2, n, 0, 64, st));
    DG_HIP(rocprim::reduce_by_key(nullptr, tb3, h2, S.d_flag.as<int32_t>(), n, bh, (int32_t *)br, S.d_n.as<unsigned long long>(), rocprim::plus<int32_t>(),
                                  rocprim::equal_to<uint64_t>(), st));
    if (int rc = S.d_tmp.ensure(std::max(tb, std::max(tb2, tb3)))) return rc;
    DG_HIP(rocprim::radix_sort_pairs(S.d_tmp.p, tb, (const uint32_t *)br, r1, (const uint64_t *)bh, h1, n, 0, 31, st));
    DG_HIP(rocprim::radix_sort_pairs(S.d_tmp.p, tb2, (const uint64_t *)h1, h2, (const uint32_t *)r1, r2, n, 0, 64, st));
    hipLaunchKernelGGL(bk_pair_flag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h2, r2, (int64_t)n, S.d_flag.as<int32_t>());
    DG_HIP(rocprim::reduce_by_key(S.d_tmp.p, tb3, h2, S.d_flag.as<int32_t>(), n, bh, (int32_t *)br, S.d_n.as<unsigned long long>(), rocprim::plus<int32_t>(),
                                  rocprim::equal_to<uint64_t>(), st));
    hipLaunchKernelGGL(bk_set_dcount_kernel, dim3(1), dim3(1), 0, st, S.d_bk_dcount.as<uint32_t>(), b, S.d_n.as<unsigned long long>());
    DG_HIP(hipGetLastError());
    return DG_OK;
}

int common_buffers(dg_ctx *c, SketchState &S, const BucketPlan &plan) {
    if (!S.attr_set) {
        DG_HIP(hipFuncSetAttribute((const void *)bk_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)table_lds_bytes(SBITS)));
        DG_HIP(hipFuncSetAttribute((const void *)bk_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 15));
        DG_HIP(hipFuncSetAttribute((const void *)bk_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 15));
        S.attr_set = true;
    }
    if (!S.h_status) DG_HIP(hipHostMalloc((void **)&S.h_status, 64, hipHostMallocDefault));
    if (int rc = S.d_bk_dcount.ensure(4 * (size_t)plan.B)) return rc;
    if (int rc = S.d_bk_dstart.ensure(4 * (size_t)(plan.B + 1))) return rc;
    if (int rc = S.d_bk_ovf.ensure(4 * (size_t)(OVF_MAX + 2))) return rc;
    if (int rc = S.d_bk_status.ensure(64)) return rc;
    return DG_OK;
}

}  // namespace

void bucket_plan(const SketchState &S, int64_t n_reads, int64_t nt, int64_t n_win, int64_t n_multi, int w, BucketPlan *plan) {
    *plan = BucketPlan{};
    plan->has_multi = n_multi > 0;
    if (S.opt_mode == 1 || nt <= 0 || n_win >= ((int64_t)1 << 31) || n_reads >= ((int64_t)1 << 31)) return;
    // ~2.5 k pairs per bucket at the emission density of random sequence (2 / (w + 1) per window)
    const int64_t n_est = std::max<int64_t>(nt, 2 * n_win / (w + 1));
    int bbits = S.opt_bucket_bits > 0 ? S.opt_bucket_bits : bits_for((uint64_t)((n_est + 2559) / 2560 - 1));
    plan->bbits = std::max(1, std::min(15, bbits));
    plan->sbits = SBITS;
    plan->B = 1 << plan->bbits;
    plan->G = (int)std::min<int64_t>(G_MAX, (nt + 127) / 128);
    plan->stride = S.opt_stride > 0 ? (uint32_t)S.opt_stride : (uint32_t)STRIDE;
    plan->residual_cap = S.opt_residual_cap < 0 ? 0u : S.opt_residual_cap > 0 ? (uint32_t)S.opt_residual_cap : (uint32_t)RCAP;
    plan->spill_cap = S.opt_spill_cap < 0 ? 0u : S.opt_spill_cap > 0 ? (uint32_t)S.opt_spill_cap : (uint32_t)SPILL_CAP;
    plan->ok = true;
}

int bucket_fast_begin(dg_ctx *c, SketchState &S, const BucketPlan &plan, BucketEmit *be) {
    if (int rc = common_buffers(c, S, plan)) return rc;
    const size_t slots = (size_t)plan.B * plan.stride;
    if (int rc = S.d_bk_hash.ensure(8 * slots)) return rc;
    if (int rc = S.d_bk_read.ensure(4 * slots)) return rc;
    // fill counters (a line each) | the table kernel's overflow record (2 + OVF_MAX words) | the spill list's counter (a line of its own)
    if (int rc = S.d_bk_fill.ensure(4 * ((size_t)plan.B * FILL_PAD + OVF_MAX + 48))) return rc;
    DG_HIP(hipMemsetAsync(S.d_bk_fill.p, 0, 4 * ((size_t)plan.B * FILL_PAD + OVF_MAX + 48), c->stream));   // the one memset of the pass
    if (int rc = S.d_spill_hash.ensure(8 * (size_t)std::max<uint32_t>(plan.spill_cap, 1))) return rc;
    if (int rc = S.d_spill_read.ensure(4 * (size_t)std::max<uint32_t>(plan.spill_cap, 1))) return rc;
    uint32_t *fill = S.d_bk_fill.as<uint32_t>();
    *be = BucketEmit{fill, S.d_bk_hash.as<uint64_t>(), S.d_bk_read.as<uint32_t>(), plan.bbits, plan.stride,
                     fill + (size_t)plan.B * FILL_PAD + OVF_MAX + 32, S.d_spill_hash.as<uint64_t>(), S.d_spill_read.as<uint32_t>(), plan.spill_cap};
    return DG_OK;
}

int bucket_exact_scatter(dg_ctx *c, SketchState &S, const BucketPlan &plan, int64_t nt) {
    hipStream_t s = c->stream;
    if (int rc = common_buffers(c, S, plan)) return rc;
    const int B = plan.B, G = plan.G;
    if (int rc = S.d_matrix.ensure(4 * (size_t)G * B)) return rc;
    if (int rc = S.d_bk_start.ensure(4 * (size_t)(2 * B + 2))) return rc;        // [0, B] starts | [B + 1, 2B] bucket counts
    uint32_t *bstart = S.d_bk_start.as<uint32_t>(), *bcnt = bstart + B + 1;
    const int64_t *sparse = S.d_tile_sparse.as<int64_t>(), *tcnt = S.d_tile_cnt.as<int64_t>();
    DG_HIP(hipMemsetAsync(bcnt, 0, 4 * (size_t)B, s));
    hipLaunchKernelGGL(bk_hist_kernel, dim3(G), dim3(BT), 4 * (size_t)B, s, sparse, tcnt, nt, S.d_hash2.as<uint64_t>(), plan.bbits, S.d_matrix.as<uint32_t>(), bcnt);
    hipLaunchKernelGGL(bk_scan_kernel, dim3(1), dim3(BT), 0, s, bcnt, B, bstart);
    hipLaunchKernelGGL(bk_basefill_kernel, dim3((B + 255) / 256), dim3(256), 0, s, S.d_matrix.as<uint32_t>(), bstart, B, G);
    uint32_t total = 0;
    DG_HIP(hipMemcpyAsync(&total, bstart + B, 4, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));                                   // this is the repeat pass: size the arrays exactly
    if (int rc = S.d_bk_hash.ensure(8 * (size_t)std::max<uint32_t>(total, 1))) return rc;
    if (int rc = S.d_bk_read.ensure(4 * (size_t)std::max<uint32_t>(total, 1))) return rc;
    hipLaunchKernelGGL(bk_scatter_kernel, dim3(G), dim3(BT), 4 * (size_t)B, s, sparse, tcnt, nt, S.d_hash2.as<uint64_t>(), S.d_aux2.as<int64_t>(), plan.bbits,
                       S.d_matrix.as<uint32_t>(), S.d_bk_hash.as<uint64_t>(), S.d_bk_read.as<uint32_t>());
    DG_HIP(hipGetLastError());
    return DG_OK;
}

int bucket_finish(dg_ctx *c, SketchState &S, const BucketPlan &plan, bool fast, uint64_t *out_hash, int32_t *out_cnt, int64_t cap,
                  int64_t *n_distinct, int64_t *n_emitted, int *outcome) {
    hipStream_t s = c->stream;
    const int B = plan.B;
    const uint32_t stride = fast ? plan.stride : 0;
    const uint32_t *bstart = fast ? nullptr : S.d_bk_start.as<uint32_t>(), *fill = fast ? S.d_bk_fill.as<uint32_t>() : nullptr;
    uint32_t *dcount = S.d_bk_dcount.as<uint32_t>(), *dstart = S.d_bk_dstart.as<uint32_t>();
    uint32_t *ovf = fast ? S.d_bk_fill.as<uint32_t>() + (size_t)B * FILL_PAD : S.d_bk_ovf.as<uint32_t>();
    const uint32_t *spill_n = fast ? S.d_bk_fill.as<uint32_t>() + (size_t)B * FILL_PAD + OVF_MAX + 32 : nullptr;
    *outcome = 0;
    if (!fast) DG_HIP(hipMemsetAsync(ovf, 0, 8, s));
    hipLaunchKernelGGL(bk_table_kernel, dim3(B), dim3(BT), table_lds_bytes(plan.sbits), s, S.d_bk_hash.as<uint64_t>(), S.d_bk_read.as<uint32_t>(), bstart, fill, stride,
                       plan.bbits, plan.sbits, plan.residual_cap, plan.has_multi ? 1 : 0, dcount, ovf, spill_n, S.d_spill_hash.as<uint64_t>(), S.d_spill_read.as<uint32_t>(),
                       plan.spill_cap);
    hipLaunchKernelGGL(bk_dscan_kernel, dim3(1), dim3(BT), 0, s, dcount, fill, bstart, B, dstart, ovf, spill_n, S.d_bk_status.as<int64_t>());
    if (out_hash)                                                      // caller's buffers: gather before the status is known (repeated if the host had to finish buckets)
        hipLaunchKernelGGL(bk_gather_kernel, dim3(B), dim3(256), 0, s, bstart, stride, dcount, dstart, S.d_bk_hash.as<uint64_t>(), S.d_bk_read.as<uint32_t>(), out_hash, out_cnt, cap);
    DG_HIP(hipGetLastError());
    DG_HIP(hipMemcpyAsync(S.h_status, S.d_bk_status.p, 40, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    S.stat_buckets = B;
    S.stat_spilled = S.h_status[4];
    if (S.h_status[3]) { *outcome = 1; return DG_OK; }                 // a bucket ran over its stride (pairs were dropped)
    const int64_t n_ovf = S.h_status[2];
    S.stat_overflow = n_ovf;
    if (n_ovf > (S.opt_host_buckets > 0 ? std::min(S.opt_host_buckets, OVF_MAX) : OVF_MAX)) { *outcome = 2; return DG_OK; }
    if (n_ovf > 0) {
        std::vector<uint32_t> list((size_t)n_ovf), tab(fast ? (size_t)B * FILL_PAD : (size_t)B + 1);
        DG_HIP(hipMemcpyAsync(list.data(), ovf + 2, 4 * (size_t)n_ovf, hipMemcpyDeviceToHost, s));
        DG_HIP(hipMemcpyAsync(tab.data(), fast ? fill : bstart, 4 * tab.size(), hipMemcpyDeviceToHost, s));
        DG_HIP(hipStreamSynchronize(s));
        for (uint32_t b : list) {
            const size_t at = fast ? (size_t)b * stride : (size_t)tab[b], n = fast ? tab[(size_t)b * FILL_PAD] : tab[b + 1] - tab[b];
            if (int rc = finish_segment(c, S, (int)b, at, n)) return rc;
        }
        hipLaunchKernelGGL(bk_dscan_kernel, dim3(1), dim3(BT), 0, s, dcount, fill, bstart, B, dstart, ovf, spill_n, S.d_bk_status.as<int64_t>());
        DG_HIP(hipMemcpyAsync(S.h_status, S.d_bk_status.p, 40, hipMemcpyDeviceToHost, s));
        DG_HIP(hipStreamSynchronize(s));
    }
    const int64_t nd = S.h_status[1];
    uint64_t *oh = out_hash;
    int32_t *oc = out_cnt;
    int64_t ocap = cap;
    if (!oh) {                                                         // host API: results stay in the state's buffers
        if (int rc = S.d_uniq.ensure(8 * (size_t)std::max<int64_t>(nd, 1))) return rc;
        if (int rc = S.d_cnt.ensure(4 * (size_t)std::max<int64_t>(nd, 1))) return rc;
        oh = S.d_uniq.as<uint64_t>(); oc = S.d_cnt.as<int32_t>(); ocap = nd;
    }
    if (!out_hash || n_ovf > 0)
        hipLaunchKernelGGL(bk_gather_kernel, dim3(B), dim3(256), 0, s, bstart, stride, dcount, dstart, S.d_bk_hash.as<uint64_t>(), S.d_bk_read.as<uint32_t>(), oh, oc, ocap);
    DG_HIP(hipGetLastError());
    *n_emitted = S.h_status[0];
    *n_distinct = nd;
    return DG_OK;
}

}  // namespace dgi
