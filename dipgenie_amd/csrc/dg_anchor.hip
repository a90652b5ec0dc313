// libdipgenie_hip.so -- haplotype index with vertex spans and the anchor join / filter / sort on the device
// (SURVEY.md s8f-3).  Replaces, behind dg_anchor_*:
//   Solver::index_kmers' position -> vertex list mapping        /root/reference/src/solver.cpp:343-357
//   Solver::compute_anchors + the Anchor_hits assembly          :415-446, 560-575
//   the shared-anchor filter                                     :590-638
//   the occurrence sort                                          :641-663
// so that Anchor_hits never materialises as vectors of vectors: the result is one flat occurrence list
// (read-minimizer id, haplotype, vertex list) in Anchor_hits order.
//
// Order contract (what the downstream graph build depends on):
//   * a vertex list = the distinct vertices under the k-mer, sorted by top_order_map (a bijection: no ties);
//   * the filter drops an id when some vertex list occurs >= threshold * num_walks times among its occurrences;
//   * inside (id, haplotype) the reference takes the occurrences in std::map<std::string,...> iteration order of the
//     key "v0_v1_..._" (lexicographic on DECIMAL strings: "10_" < "9_"), push order inside one key, and then runs
//     std::sort by (front vertex, back vertex).  libstdc++'s std::sort is a plain insertion sort -- stable -- up to 16
//     elements, so for such groups the result is the order by (front, back, key, push order): one device sort with that
//     comparator.  A larger group that holds two DIFFERENT lists with equal (front, back) would expose introsort's
//     unstable partitioning; such groups are counted (n_unstable_groups) and the caller re-does the stage on the host
//     (none exists in any test input or synthetic panel; identical lists tie harmlessly).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dg_internal.hpp"

namespace dgi {

struct HapIndex {                       // one haplotype's minimizers on the device
    DevBuf hash, voff, v;               // uint64 [n], uint32 [n + 1], int32 [nv]
    int64_t n = 0, nv = 0;
};

struct AnchorState {
    int n_haps = 0, n_vertices = 0, k = 0, w = 0, next_h = 0;
    DevBuf d_top, d_step_vtx, d_step_start, d_cnt, d_tmp, d_imp_hash, d_imp_pos;
    std::vector<HapIndex *> haps;
    ~AnchorState() { for (auto *h : haps) delete h; }
};
void anchor_state_free(AnchorState *a) { delete a; }

// ---------------------------------------------------------------------------------------------
// position -> vertex list (solver.cpp:343-357)
// ---------------------------------------------------------------------------------------------
// step s of the haplotype covers bases [start[s], start[s + 1]); the step holding base p is the last one with
// start[s] <= p (empty segments share their start with the next non-empty one and are never the last)
__device__ __forceinline__ int64_t step_of(const int64_t *__restrict__ start, int64_t n_steps, int64_t p) {
    int64_t lo = 0, hi = n_steps;       // first s with start[s] > p
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (start[mid] <= p) lo = mid + 1; else hi = mid; }
    return lo - 1;
}

// MODE 0: number of distinct vertices under minimizer m; MODE 1: write them at v[voff[m]..), sorted by top_order_map
template <int MODE>
__global__ void span_kernel(const int64_t *__restrict__ pos, int64_t n, int k, const int32_t *__restrict__ step_vtx, const int64_t *__restrict__ start,
                            int64_t n_steps, const int32_t *__restrict__ top, uint32_t *__restrict__ cnt, const uint32_t *__restrict__ voff, int32_t *__restrict__ v) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const int64_t p = pos[m];
    int64_t s = step_of(start, n_steps, p);
    int32_t *out = MODE ? v + voff[m] : nullptr;
    uint32_t nu = 0;
    int32_t seen[8];                                                   // MODE 0: the first few vertices in registers, the rest by re-walking
    const int64_t s0 = s;
    for (;; ++s) {
        if (start[s + 1] > start[s]) {                                 // empty segments contribute no base
            const int32_t vtx = step_vtx[s];
            bool dup = false;
            if (MODE) { for (uint32_t q = 0; q < nu; ++q) dup |= out[q] == vtx; }
            else {
                for (uint32_t q = 0; q < nu && q < 8; ++q) dup |= seen[q] == vtx;
                if (!dup && nu > 8) {                                  // beyond the register window: look back along the walk
                    for (int64_t t = s0; t < s && !dup; ++t) dup = start[t + 1] > start[t] && step_vtx[t] == vtx;
                }
            }
            if (!dup) {
                if (MODE) out[nu] = vtx; else if (nu < 8) seen[nu] = vtx;
                ++nu;
            }
        }
        if (start[s + 1] >= p + k || s + 1 >= n_steps) break;
    }
    if (!MODE) { cnt[m] = nu; return; }
    for (uint32_t a = 1; a < nu; ++a) {                                // insertion sort by column rank (distinct ranks)
        const int32_t x = out[a];
        const int32_t rx = top[x];
        uint32_t b = a;
        while (b > 0 && top[out[b - 1]] > rx) { out[b] = out[b - 1]; --b; }
        out[b] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// join, filter, sort
// ---------------------------------------------------------------------------------------------
struct Lists {                          // all haplotypes' minimizers concatenated: list of minimizer g = v[voff[g] .. voff[g + 1])
    const uint32_t *voff;
    const int32_t *v;
};

// "dec(a)_" against "dec(b)_" as strings: digits compare left to right; when one number's digits are a prefix of the
// other's, the shorter string has '_' (0x5F, above every digit) where the longer has a digit, so it is the GREATER one
__device__ __forceinline__ int token_cmp(uint32_t a, uint32_t b) {
    if (a == b) return 0;
    uint32_t pa = 1, pb = 1;            // 10^(digits - 1)
    while (a / pa >= 10) pa *= 10;
    while (b / pb >= 10) pb *= 10;
    if (pa == pb) return a < b ? -1 : 1;
    if (pa < pb) {                      // a is shorter: compare it with b's leading digits
        const uint32_t bp = b / (pb / pa);
        return a < bp ? -1 : 1;         // a == bp: a's '_' meets a digit of b -> a is greater
    }
    const uint32_t ap = a / (pa / pb);
    return ap <= b ? -1 : 1;            // ap == b: b's '_' meets a digit of a -> b is greater
}
// key "v0_v1_..._" of list x against that of list y (std::map<std::string,...> order, solver.cpp:595-628)
__device__ __forceinline__ int key_cmp(const Lists &L, uint32_t gx, uint32_t gy) {
    const uint32_t ax = L.voff[gx], bx = L.voff[gx + 1], ay = L.voff[gy], by = L.voff[gy + 1];
    const uint32_t nx = bx - ax, ny = by - ay, nm = nx < ny ? nx : ny;
    for (uint32_t q = 0; q < nm; ++q) {
        const int c = token_cmp((uint32_t)L.v[ax + q], (uint32_t)L.v[ay + q]);
        if (c) return c;
    }
    return nx == ny ? 0 : (nx < ny ? -1 : 1);      // a proper prefix is the smaller string
}

struct Occs { const int32_t *id; const uint32_t *g; const int32_t *hap; };   // occurrence j: read-minimizer id, minimizer g, haplotype

struct CmpFilter {                      // (id, key, push order): equal vertex lists of one id become adjacent
    Occs O; Lists L;
    __device__ bool operator()(uint32_t x, uint32_t y) const {
        if (O.id[x] != O.id[y]) return O.id[x] < O.id[y];
        const int c = key_cmp(L, O.g[x], O.g[y]);
        if (c) return c < 0;
        return x < y;                   // occurrence index = (haplotype, minimizer) order = the reference's push order
    }
};
struct CmpFinal {                       // (id, haplotype, front, back, key, push order)
    Occs O; Lists L;
    __device__ bool operator()(uint32_t x, uint32_t y) const {
        if (O.id[x] != O.id[y]) return O.id[x] < O.id[y];
        if (O.hap[x] != O.hap[y]) return O.hap[x] < O.hap[y];
        const uint32_t gx = O.g[x], gy = O.g[y];
        const int32_t fx = L.v[L.voff[gx]], fy = L.v[L.voff[gy]];
        if (fx != fy) return fx < fy;
        const int32_t bx = L.v[L.voff[gx + 1] - 1], by = L.v[L.voff[gy + 1] - 1];
        if (bx != by) return bx < by;
        const int c = key_cmp(L, gx, gy);
        if (c) return c < 0;
        return x < y;
    }
};

__global__ void id_lookup_kernel(const uint64_t *__restrict__ ghash, int64_t G, const uint64_t *__restrict__ sp, int64_t n_sp, int32_t *__restrict__ id,
                                 const uint32_t *__restrict__ voff) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint64_t key = ghash[g];
    int64_t lo = 0, hi = n_sp;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sp[mid] < key) lo = mid + 1; else hi = mid; }
    // (a minimizer whose k-mer covers no base of a non-empty segment cannot exist; an empty list would be skipped like :430)
    id[g] = (lo < n_sp && sp[lo] == key && voff[g + 1] > voff[g]) ? (int32_t)lo : -1;
}
__global__ void hap_of_kernel(const int64_t *__restrict__ hap_off, int n_haps, int64_t G, int32_t *__restrict__ hap) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    int lo = 0, hi = n_haps;            // last h with hap_off[h] <= g
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (hap_off[mid] <= g) lo = mid + 1; else hi = mid; }
    hap[g] = lo - 1;
}
// occurrences = minimizers with an id, in (haplotype, minimizer) order: flag -> scan -> scatter
__global__ void occ_flag_kernel(const int32_t *__restrict__ id, int64_t G, uint32_t *__restrict__ flag) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) flag[g] = id[g] >= 0 ? 1u : 0u;
}
__global__ void occ_scatter_kernel(const int32_t *__restrict__ id, const int32_t *__restrict__ hap, const uint32_t *__restrict__ slot, int64_t G,
                                   int32_t *__restrict__ oid, uint32_t *__restrict__ og, int32_t *__restrict__ ohap) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G || id[g] < 0) return;
    const uint32_t j = slot[g];
    oid[j] = id[g]; og[j] = (uint32_t)g; ohap[j] = hap[g];
}
__global__ void iota_kernel(uint32_t *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}
// The filter only asks how often one vertex list occurs inside an id, so equal lists are brought together by two radix
// sorts -- by a 64-bit fingerprint of the list, then (stable) by id -- instead of a comparison sort that reads the lists.
__device__ __forceinline__ uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k;
}
__global__ void fingerprint_kernel(Occs O, Lists L, int64_t n, uint64_t *__restrict__ fp, uint32_t *__restrict__ idx) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t g = O.g[j], a = L.voff[g], b = L.voff[g + 1];
    uint64_t h = mix64((uint64_t)(b - a) + 0x9E3779B97F4A7C15ULL);
    for (uint32_t q = a; q < b; ++q) h = mix64(h ^ ((uint64_t)(uint32_t)L.v[q] + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2)));
    fp[j] = h;
    idx[j] = (uint32_t)j;
}
__global__ void gather_id_kernel(const uint32_t *__restrict__ order, const int32_t *__restrict__ oid, int64_t n, uint32_t *__restrict__ key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = (uint32_t)oid[order[i]];
}
__device__ __forceinline__ bool list_equal(const Lists &L, uint32_t gx, uint32_t gy) {
    const uint32_t ax = L.voff[gx], nx = L.voff[gx + 1] - ax, ay = L.voff[gy], ny = L.voff[gy + 1] - ay;
    if (nx != ny) return false;
    for (uint32_t q = 0; q < nx; ++q) if (L.v[ax + q] != L.v[ay + q]) return false;
    return true;
}
// head[i] = 1 where a new (id, fingerprint) run starts; inside a run every list must equal its predecessor -- a
// fingerprint collision (different lists, same 64 bits) is counted and the caller falls back to the exact order
__global__ void run_head_kernel(const uint32_t *__restrict__ order, int64_t n, Occs O, Lists L, const uint64_t *__restrict__ fp, uint32_t *__restrict__ head,
                                unsigned long long *__restrict__ collisions) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0;
    if (!h) {
        const uint32_t x = order[i - 1], y = order[i];
        h = O.id[x] != O.id[y] || fp[x] != fp[y];
        if (!h && !list_equal(L, O.g[x], O.g[y])) atomicAdd(collisions, 1ULL);
    }
    head[i] = h ? 1u : 0u;
}
__global__ void exact_head_kernel(const uint32_t *__restrict__ order, int64_t n, Occs O, Lists L, uint32_t *__restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0;
    if (!h) { const uint32_t x = order[i - 1], y = order[i]; h = O.id[x] != O.id[y] || !list_equal(L, O.g[x], O.g[y]); }
    head[i] = h ? 1u : 0u;
}
__global__ void run_drop_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ head, int64_t n, Occs O, float thr, uint8_t *__restrict__ dropped) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    int64_t j = i + 1;
    while (j < n && !head[j] && (float)(j - i) < thr) ++j;            // only as far as the threshold needs
    if ((float)(j - i) >= thr) dropped[O.id[order[i]]] = 1;
}
__global__ void keep_flag_kernel(const int32_t *__restrict__ oid, const uint8_t *__restrict__ dropped, int64_t n, uint32_t *__restrict__ flag) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flag[j] = dropped[oid[j]] ? 0u : 1u;
}
__global__ void keep_scatter_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ slot, int64_t n, uint32_t *__restrict__ kept) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && flag[j]) kept[slot[j]] = (uint32_t)j;
}
// final order -> output arrays; unstable[0] counts (id, haplotype) groups of more than 16 occurrences in which two
// different lists share (front, back)
__global__ void emit_kernel(const uint32_t *__restrict__ order, int64_t n, Occs O, Lists L, int32_t *__restrict__ out_id, int32_t *__restrict__ out_hap,
                            uint32_t *__restrict__ out_len, unsigned long long *__restrict__ unstable) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = order[i], gx = O.g[x];
    out_id[i] = O.id[x]; out_hap[i] = O.hap[x]; out_len[i] = L.voff[gx + 1] - L.voff[gx];
    if (i + 1 < n) {
        const uint32_t y = order[i + 1], gy = O.g[y];
        if (O.id[x] == O.id[y] && O.hap[x] == O.hap[y] && L.v[L.voff[gx]] == L.v[L.voff[gy]] && L.v[L.voff[gx + 1] - 1] == L.v[L.voff[gy + 1] - 1] &&
            key_cmp(L, gx, gy) != 0) {
            int64_t a = i, b = i + 1;                                  // size of the (id, haplotype) group around the tie
            while (a > 0 && O.id[order[a - 1]] == O.id[x] && O.hap[order[a - 1]] == O.hap[x] && b - a <= 16) --a;
            while (b + 1 < n && O.id[order[b + 1]] == O.id[x] && O.hap[order[b + 1]] == O.hap[x] && b - a <= 16) ++b;
            if (b - a + 1 > 16) atomicAdd(unstable, 1ULL);
        }
    }
}
__global__ void gather_lists_kernel(const uint32_t *__restrict__ order, int64_t n, Occs O, Lists L, const uint32_t *__restrict__ out_off, int32_t *__restrict__ vpool) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = O.g[order[i]];
    const uint32_t a = L.voff[g], len = L.voff[g + 1] - a, o = out_off[i];
    for (uint32_t q = 0; q < len; ++q) vpool[o + q] = L.v[a + q];
}

static unsigned blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 255) / 256); }

static int exclusive_scan_u32(DevBuf &tmp, const uint32_t *in, uint32_t *out, int64_t n, hipStream_t s) {
    size_t tb = 0;
    DG_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
    if (int rc = tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::exclusive_scan(tmp.p, tb, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
    return DG_OK;
}
template <class K>
static int radix_pairs(DevBuf &tmp, const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, int64_t n, int end_bit, hipStream_t s) {
    size_t tb = 0;
    DG_HIP(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, (size_t)n, 0, end_bit, s));
    if (int rc = tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::radix_sort_pairs(tmp.p, tb, kin, kout, vin, vout, (size_t)n, 0, end_bit, s));
    return DG_OK;
}
template <class Cmp>
static int sort_indices(DevBuf &tmp, uint32_t *in, uint32_t *out, int64_t n, Cmp cmp, hipStream_t s) {
    size_t tb = 0;
    DG_HIP(rocprim::merge_sort(nullptr, tb, in, out, (size_t)n, cmp, s));
    if (int rc = tmp.ensure(tb)) return rc;
    DG_HIP(rocprim::merge_sort(tmp.p, tb, in, out, (size_t)n, cmp, s));
    return DG_OK;
}

}  // namespace dgi

using namespace dgi;

extern "C" int dg_anchor_begin(dg_ctx *c, int32_t n_haps, int32_t n_vertices, const int32_t *top_order_map, int k, int w) {
    if (int rc = bind(c)) return rc;
    if (n_haps < 1 || n_vertices < 1 || !top_order_map || k < 1 || k > 255 || w < 1 || w > 255) { set_error("dg_anchor_begin: bad arguments"); return DG_ERR_ARG; }
    delete c->an;
    c->an = new AnchorState();
    AnchorState &A = *c->an;
    A.n_haps = n_haps; A.n_vertices = n_vertices; A.k = k; A.w = w;
    if (int rc = A.d_top.ensure(4 * (size_t)n_vertices)) return rc;
    DG_HIP(hipMemcpyAsync(A.d_top.p, top_order_map, 4 * (size_t)n_vertices, hipMemcpyHostToDevice, c->stream));
    DG_HIP(hipStreamSynchronize(c->stream));
    return DG_OK;
}

// spans of one haplotype's minimizers (device hash / position lists hd, pd) -> its HapIndex
static int add_haplotype_spans(dg_ctx *c, int32_t h, int64_t len, const uint64_t *hd, const int64_t *pd, int64_t n, const int32_t *step_vtx,
                               const int64_t *step_start, int64_t n_steps, int64_t *n_minimizers);

static int check_haplotype_args(dg_ctx *c, int32_t h, int64_t len, const int32_t *step_vtx, const int64_t *step_start, int64_t n_steps) {
    if (!c->an) { set_error("dg_anchor_add_haplotype: dg_anchor_begin first"); return DG_ERR_STATE; }
    AnchorState &A = *c->an;
    if (h != A.next_h || h >= A.n_haps) { set_error("dg_anchor_add_haplotype: haplotypes must be added in order (got %d, expected %d)", h, A.next_h); return DG_ERR_ARG; }
    if (!step_vtx || !step_start || n_steps < 0 || step_start[0] != 0 || (n_steps > 0 && step_start[n_steps] != len)) {
        set_error("dg_anchor_add_haplotype: step_start must run from 0 to the haplotype length"); return DG_ERR_ARG;
    }
    for (int64_t q = 0; q < n_steps; ++q)
        if (step_start[q + 1] < step_start[q] || step_vtx[q] < 0 || step_vtx[q] >= A.n_vertices) { set_error("dg_anchor_add_haplotype: bad step %lld", (long long)q); return DG_ERR_ARG; }
    return DG_OK;
}

extern "C" int dg_anchor_add_haplotype(dg_ctx *c, int32_t h, const char *seq, int64_t len, const int32_t *step_vtx, const int64_t *step_start,
                                       int64_t n_steps, int64_t *n_minimizers) {
    if (int rc = bind(c)) return rc;
    if (int rc = check_haplotype_args(c, h, len, step_vtx, step_start, n_steps)) return rc;
    const uint64_t *hd = nullptr; const int64_t *pd = nullptr; int64_t n = 0;
    if (int rc = sketch_haplotype_dev(c, seq, len, c->an->k, c->an->w, &hd, &pd, &n)) return rc;
    return add_haplotype_spans(c, h, len, hd, pd, n, step_vtx, step_start, n_steps, n_minimizers);
}

// the same for a haplotype whose minimizer list (the output of dg_sketch_haplotype) was computed elsewhere -- by another rank of a
// haplotype-sharded run (solver.cpp:470-473 runs index_kmers once per haplotype, independently)
extern "C" int dg_anchor_add_haplotype_sketched(dg_ctx *c, int32_t h, int64_t len, const uint64_t *hash, const int64_t *pos, int64_t n, const int32_t *step_vtx,
                                                const int64_t *step_start, int64_t n_steps) {
    if (int rc = bind(c)) return rc;
    if (int rc = check_haplotype_args(c, h, len, step_vtx, step_start, n_steps)) return rc;
    if (n < 0 || (n > 0 && (!hash || !pos))) { set_error("dg_anchor_add_haplotype_sketched: null minimizer list"); return DG_ERR_ARG; }
    for (int64_t q = 0; q < n; ++q)
        if (pos[q] < 0 || pos[q] + c->an->k > len || (q > 0 && pos[q] < pos[q - 1])) { set_error("dg_anchor_add_haplotype_sketched: bad position at %lld", (long long)q); return DG_ERR_ARG; }
    AnchorState &A = *c->an;
    if (int rc = A.d_imp_hash.ensure(8 * (size_t)std::max<int64_t>(n, 1))) return rc;
    if (int rc = A.d_imp_pos.ensure(8 * (size_t)std::max<int64_t>(n, 1))) return rc;
    if (n) {
        DG_HIP(hipMemcpyAsync(A.d_imp_hash.p, hash, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        DG_HIP(hipMemcpyAsync(A.d_imp_pos.p, pos, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    }
    return add_haplotype_spans(c, h, len, A.d_imp_hash.as<uint64_t>(), A.d_imp_pos.as<int64_t>(), n, step_vtx, step_start, n_steps, nullptr);
}

static int add_haplotype_spans(dg_ctx *c, int32_t h, int64_t len, const uint64_t *hd, const int64_t *pd, int64_t n, const int32_t *step_vtx,
                               const int64_t *step_start, int64_t n_steps, int64_t *n_minimizers) {
    (void)h; (void)len;
    AnchorState &A = *c->an;
    hipStream_t s = c->stream;
    HapIndex *H = new HapIndex();
    A.haps.push_back(H);
    A.next_h++;
    H->n = n;
    if (n_minimizers) *n_minimizers = n;
    if (int rc = H->hash.ensure(8 * (size_t)std::max<int64_t>(n, 1))) return rc;
    if (int rc = H->voff.ensure(4 * (size_t)(n + 1))) return rc;
    if (n == 0) { DG_HIP(hipMemsetAsync(H->voff.p, 0, 4, s)); DG_HIP(hipStreamSynchronize(s)); return DG_OK; }
    DG_HIP(hipMemcpyAsync(H->hash.p, hd, 8 * (size_t)n, hipMemcpyDeviceToDevice, s));
    if (int rc = A.d_step_vtx.ensure(4 * (size_t)std::max<int64_t>(n_steps, 1))) return rc;
    if (int rc = A.d_step_start.ensure(8 * (size_t)(n_steps + 1))) return rc;
    DG_HIP(hipMemcpyAsync(A.d_step_vtx.p, step_vtx, 4 * (size_t)n_steps, hipMemcpyHostToDevice, s));
    DG_HIP(hipMemcpyAsync(A.d_step_start.p, step_start, 8 * (size_t)(n_steps + 1), hipMemcpyHostToDevice, s));
    if (int rc = A.d_cnt.ensure(4 * (size_t)(n + 1))) return rc;
    DG_HIP(hipMemsetAsync((char *)A.d_cnt.p + 4 * (size_t)n, 0, 4, s));
    hipLaunchKernelGGL(span_kernel<0>, dim3(blocks(n)), dim3(256), 0, s, pd, n, A.k, A.d_step_vtx.as<int32_t>(), A.d_step_start.as<int64_t>(), n_steps,
                       A.d_top.as<int32_t>(), A.d_cnt.as<uint32_t>(), (const uint32_t *)nullptr, (int32_t *)nullptr);
    if (int rc = exclusive_scan_u32(A.d_tmp, A.d_cnt.as<uint32_t>(), H->voff.as<uint32_t>(), n + 1, s)) return rc;
    uint32_t nv = 0;
    DG_HIP(hipMemcpyAsync(&nv, H->voff.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    H->nv = nv;
    if (int rc = H->v.ensure(4 * (size_t)std::max<uint32_t>(nv, 1))) return rc;
    hipLaunchKernelGGL(span_kernel<1>, dim3(blocks(n)), dim3(256), 0, s, pd, n, A.k, A.d_step_vtx.as<int32_t>(), A.d_step_start.as<int64_t>(), n_steps,
                       A.d_top.as<int32_t>(), (uint32_t *)nullptr, H->voff.as<uint32_t>(), H->v.as<int32_t>());
    DG_HIP(hipGetLastError());
    DG_HIP(hipStreamSynchronize(s));    // the step arrays and the sketch buffers are reused by the next haplotype
    return DG_OK;
}

extern "C" int dg_anchor_finish(dg_ctx *c, const uint64_t *sp_hash, int64_t n_sp, float min_shared, dg_anchor_result *out) {
    if (int rc = bind(c)) return rc;
    if (!c->an || !out) { set_error("dg_anchor_finish: dg_anchor_begin first"); return DG_ERR_STATE; }
    AnchorState &A = *c->an;
    if (A.next_h != A.n_haps) { set_error("dg_anchor_finish: %d of %d haplotypes added", A.next_h, A.n_haps); return DG_ERR_STATE; }
    if (n_sp < 0 || (n_sp > 0 && !sp_hash) || n_sp >= ((int64_t)1 << 31)) { set_error("dg_anchor_finish: bad spectrum"); return DG_ERR_ARG; }
    memset(out, 0, sizeof *out);
    hipStream_t s = c->stream;
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    double tl = ts0.tv_sec + 1e-9 * ts0.tv_nsec;
    auto lap = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(s);
        struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        const double t = ts.tv_sec + 1e-9 * ts.tv_nsec;
        fprintf(stderr, "[dipgenie_hip] anchors: %-16s %.3f s\n", what, t - tl); tl = t;
    };
    // concatenate the haplotypes: minimizer g = hap_off[h] + m, list offsets rebased
    std::vector<int64_t> hap_off(A.n_haps + 1, 0), v_off(A.n_haps + 1, 0);
    for (int h = 0; h < A.n_haps; ++h) { hap_off[h + 1] = hap_off[h] + A.haps[h]->n; v_off[h + 1] = v_off[h] + A.haps[h]->nv; }
    const int64_t G = hap_off[A.n_haps], NV = v_off[A.n_haps];
    if (G >= ((int64_t)1 << 32) - 1 || NV >= ((int64_t)1 << 32) - 1) { set_error("dg_anchor_finish: more than 2^32 minimizers / list entries"); return DG_ERR_UNSUPPORTED; }
    DevBuf d_hash, d_voff, d_v, d_hapoff, d_sp, d_id, d_hap, d_flag, d_slot, d_oid, d_og, d_ohap, d_idx, d_ord, d_head, d_drop, d_kept, d_tmp;
    DevBuf d_out_id, d_out_hap, d_out_len, d_out_off, d_vpool, d_unstable;
    if (int rc = d_hash.ensure(8 * (size_t)std::max<int64_t>(G, 1))) return rc;
    if (int rc = d_voff.ensure(4 * (size_t)(G + 1))) return rc;
    if (int rc = d_v.ensure(4 * (size_t)std::max<int64_t>(NV, 1))) return rc;
    std::vector<uint32_t> rebased;
    for (int h = 0; h < A.n_haps; ++h) {
        const HapIndex &H = *A.haps[h];
        if (H.n == 0) continue;
        DG_HIP(hipMemcpyAsync(d_hash.as<uint64_t>() + hap_off[h], H.hash.p, 8 * (size_t)H.n, hipMemcpyDeviceToDevice, s));
        if (H.nv) DG_HIP(hipMemcpyAsync(d_v.as<int32_t>() + v_off[h], H.v.p, 4 * (size_t)H.nv, hipMemcpyDeviceToDevice, s));
        // list offsets: download, rebase, upload (a few MB per haplotype)
        rebased.resize((size_t)H.n);
        DG_HIP(hipMemcpyAsync(rebased.data(), H.voff.p, 4 * (size_t)H.n, hipMemcpyDeviceToHost, s));
        DG_HIP(hipStreamSynchronize(s));
        for (auto &x : rebased) x += (uint32_t)v_off[h];
        DG_HIP(hipMemcpyAsync(d_voff.as<uint32_t>() + hap_off[h], rebased.data(), 4 * (size_t)H.n, hipMemcpyHostToDevice, s));
        DG_HIP(hipStreamSynchronize(s));
    }
    const uint32_t nv_end = (uint32_t)NV;
    DG_HIP(hipMemcpyAsync(d_voff.as<uint32_t>() + G, &nv_end, 4, hipMemcpyHostToDevice, s));
    if (int rc = d_hapoff.ensure(8 * (size_t)(A.n_haps + 1))) return rc;
    DG_HIP(hipMemcpyAsync(d_hapoff.p, hap_off.data(), 8 * (size_t)(A.n_haps + 1), hipMemcpyHostToDevice, s));
    if (int rc = d_sp.ensure(8 * (size_t)std::max<int64_t>(n_sp, 1))) return rc;
    if (n_sp) DG_HIP(hipMemcpyAsync(d_sp.p, sp_hash, 8 * (size_t)n_sp, hipMemcpyHostToDevice, s));
    DG_HIP(hipStreamSynchronize(s));
    lap("concatenate");
    if (G == 0 || n_sp == 0) return DG_OK;
    const Lists L{d_voff.as<uint32_t>(), d_v.as<int32_t>()};
    // ---- compute_anchors (:415-446): id of every haplotype minimizer, occurrences in (haplotype, minimizer) order
    if (int rc = d_id.ensure(4 * (size_t)G)) return rc;
    if (int rc = d_hap.ensure(4 * (size_t)G)) return rc;
    if (int rc = d_flag.ensure(4 * (size_t)(G + 1))) return rc;
    if (int rc = d_slot.ensure(4 * (size_t)(G + 1))) return rc;
    hipLaunchKernelGGL(id_lookup_kernel, dim3(blocks(G)), dim3(256), 0, s, d_hash.as<uint64_t>(), G, d_sp.as<uint64_t>(), n_sp, d_id.as<int32_t>(), L.voff);
    hipLaunchKernelGGL(hap_of_kernel, dim3(blocks(G)), dim3(256), 0, s, d_hapoff.as<int64_t>(), A.n_haps, G, d_hap.as<int32_t>());
    hipLaunchKernelGGL(occ_flag_kernel, dim3(blocks(G)), dim3(256), 0, s, d_id.as<int32_t>(), G, d_flag.as<uint32_t>());
    DG_HIP(hipMemsetAsync(d_flag.as<uint32_t>() + G, 0, 4, s));
    if (int rc = exclusive_scan_u32(d_tmp, d_flag.as<uint32_t>(), d_slot.as<uint32_t>(), G + 1, s)) return rc;
    uint32_t n_occ = 0;
    DG_HIP(hipMemcpyAsync(&n_occ, d_slot.as<uint32_t>() + G, 4, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    if (n_occ == 0) return DG_OK;
    if (int rc = d_oid.ensure(4 * (size_t)n_occ)) return rc;
    if (int rc = d_og.ensure(4 * (size_t)n_occ)) return rc;
    if (int rc = d_ohap.ensure(4 * (size_t)n_occ)) return rc;
    hipLaunchKernelGGL(occ_scatter_kernel, dim3(blocks(G)), dim3(256), 0, s, d_id.as<int32_t>(), d_hap.as<int32_t>(), d_slot.as<uint32_t>(), G,
                       d_oid.as<int32_t>(), d_og.as<uint32_t>(), d_ohap.as<int32_t>());
    const Occs O{d_oid.as<int32_t>(), d_og.as<uint32_t>(), d_ohap.as<int32_t>()};
    lap("join");
    if (dbg) fprintf(stderr, "[dipgenie_hip] anchors: %lld minimizers, %u with a read hash\n", (long long)G, n_occ);
    // ---- shared-anchor filter (:590-638): equal vertex lists of one id adjacent, runs of >= min_shared drop the id
    DevBuf d_fp, d_fp2, d_key, d_key2, d_coll;
    if (int rc = d_idx.ensure(4 * (size_t)n_occ)) return rc;
    if (int rc = d_ord.ensure(4 * (size_t)n_occ)) return rc;
    if (int rc = d_head.ensure(4 * (size_t)(n_occ + 1))) return rc;
    if (int rc = d_drop.ensure((size_t)n_sp)) return rc;
    if (int rc = d_fp.ensure(8 * (size_t)n_occ)) return rc;
    if (int rc = d_fp2.ensure(8 * (size_t)n_occ)) return rc;
    if (int rc = d_key.ensure(4 * (size_t)n_occ)) return rc;
    if (int rc = d_key2.ensure(4 * (size_t)n_occ)) return rc;
    if (int rc = d_coll.ensure(8)) return rc;
    DG_HIP(hipMemsetAsync(d_coll.p, 0, 8, s));
    hipLaunchKernelGGL(fingerprint_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, O, L, (int64_t)n_occ, d_fp.as<uint64_t>(), d_idx.as<uint32_t>());
    if (int rc = radix_pairs<uint64_t>(d_tmp, d_fp.as<uint64_t>(), d_fp2.as<uint64_t>(), d_idx.as<uint32_t>(), d_ord.as<uint32_t>(), n_occ, 64, s)) return rc;
    hipLaunchKernelGGL(gather_id_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, d_ord.as<uint32_t>(), d_oid.as<int32_t>(), (int64_t)n_occ, d_key.as<uint32_t>());
    int id_bits = 1;
    while (id_bits < 32 && ((int64_t)1 << id_bits) < n_sp) ++id_bits;
    if (int rc = radix_pairs<uint32_t>(d_tmp, d_key.as<uint32_t>(), d_key2.as<uint32_t>(), d_ord.as<uint32_t>(), d_idx.as<uint32_t>(), n_occ, id_bits, s)) return rc;   // stable: fingerprints stay grouped
    uint32_t *filter_order = d_idx.as<uint32_t>();
    DG_HIP(hipMemsetAsync(d_drop.p, 0, (size_t)n_sp, s));
    hipLaunchKernelGGL(run_head_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, filter_order, (int64_t)n_occ, O, L, d_fp.as<uint64_t>(), d_head.as<uint32_t>(),
                       d_coll.as<unsigned long long>());
    unsigned long long collisions = 0;
    DG_HIP(hipMemcpyAsync(&collisions, d_coll.p, 8, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    if (collisions) {                   // exact order instead (never seen: 64-bit fingerprints of <= 10^8 lists)
        hipLaunchKernelGGL(iota_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, d_ord.as<uint32_t>(), (int64_t)n_occ);
        if (int rc = sort_indices(d_tmp, d_ord.as<uint32_t>(), d_idx.as<uint32_t>(), n_occ, CmpFilter{O, L}, s)) return rc;
        hipLaunchKernelGGL(exact_head_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, filter_order, (int64_t)n_occ, O, L, d_head.as<uint32_t>());
    }
    lap("filter sorts");
    hipLaunchKernelGGL(run_drop_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, filter_order, d_head.as<uint32_t>(), (int64_t)n_occ, O, min_shared, d_drop.as<uint8_t>());
    // survivors, still in (haplotype, minimizer) order
    hipLaunchKernelGGL(keep_flag_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, d_oid.as<int32_t>(), d_drop.as<uint8_t>(), (int64_t)n_occ, d_head.as<uint32_t>());
    DG_HIP(hipMemsetAsync(d_head.as<uint32_t>() + n_occ, 0, 4, s));
    if (int rc = d_slot.ensure(4 * (size_t)(n_occ + 1))) return rc;
    if (int rc = exclusive_scan_u32(d_tmp, d_head.as<uint32_t>(), d_slot.as<uint32_t>(), (int64_t)n_occ + 1, s)) return rc;
    uint32_t n_keep = 0;
    DG_HIP(hipMemcpyAsync(&n_keep, d_slot.as<uint32_t>() + n_occ, 4, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    if (n_keep == 0) return DG_OK;
    if (int rc = d_kept.ensure(4 * (size_t)n_keep)) return rc;
    hipLaunchKernelGGL(keep_scatter_kernel, dim3(blocks(n_occ)), dim3(256), 0, s, d_head.as<uint32_t>(), d_slot.as<uint32_t>(), (int64_t)n_occ, d_kept.as<uint32_t>());
    lap("drop + compact");
    if (dbg) fprintf(stderr, "[dipgenie_hip] anchors: %u occurrences survive the filter\n", n_keep);
    // ---- occurrence sort (:641-663) in Anchor_hits order
    if (int rc = sort_indices(d_tmp, d_kept.as<uint32_t>(), d_ord.as<uint32_t>(), n_keep, CmpFinal{O, L}, s)) return rc;
    lap("final sort");
    if (int rc = d_out_id.ensure(4 * (size_t)n_keep)) return rc;
    if (int rc = d_out_hap.ensure(4 * (size_t)n_keep)) return rc;
    if (int rc = d_out_len.ensure(4 * (size_t)(n_keep + 1))) return rc;
    if (int rc = d_out_off.ensure(4 * (size_t)(n_keep + 1))) return rc;
    if (int rc = d_unstable.ensure(8)) return rc;
    DG_HIP(hipMemsetAsync(d_unstable.p, 0, 8, s));
    DG_HIP(hipMemsetAsync(d_out_len.as<uint32_t>() + n_keep, 0, 4, s));
    hipLaunchKernelGGL(emit_kernel, dim3(blocks(n_keep)), dim3(256), 0, s, d_ord.as<uint32_t>(), (int64_t)n_keep, O, L, d_out_id.as<int32_t>(), d_out_hap.as<int32_t>(),
                       d_out_len.as<uint32_t>(), d_unstable.as<unsigned long long>());
    if (int rc = exclusive_scan_u32(d_tmp, d_out_len.as<uint32_t>(), d_out_off.as<uint32_t>(), (int64_t)n_keep + 1, s)) return rc;
    uint32_t n_vtx = 0;
    unsigned long long unstable = 0;
    DG_HIP(hipMemcpyAsync(&n_vtx, d_out_off.as<uint32_t>() + n_keep, 4, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(&unstable, d_unstable.p, 8, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    if (int rc = d_vpool.ensure(4 * (size_t)std::max<uint32_t>(n_vtx, 1))) return rc;
    hipLaunchKernelGGL(gather_lists_kernel, dim3(blocks(n_keep)), dim3(256), 0, s, d_ord.as<uint32_t>(), (int64_t)n_keep, O, L, d_out_off.as<uint32_t>(), d_vpool.as<int32_t>());
    DG_HIP(hipGetLastError());
    out->occ_id = (int32_t *)malloc(4 * (size_t)n_keep); out->occ_hap = (int32_t *)malloc(4 * (size_t)n_keep);
    out->occ_off = (uint32_t *)malloc(4 * (size_t)n_keep); out->occ_len = (uint32_t *)malloc(4 * (size_t)n_keep);
    out->vpool = (int32_t *)malloc(4 * (size_t)std::max<uint32_t>(n_vtx, 1));
    if (!out->occ_id || !out->occ_hap || !out->occ_off || !out->occ_len || !out->vpool) { set_error("host malloc failed"); return DG_ERR_OOM; }
    DG_HIP(hipMemcpyAsync(out->occ_id, d_out_id.p, 4 * (size_t)n_keep, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(out->occ_hap, d_out_hap.p, 4 * (size_t)n_keep, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(out->occ_off, d_out_off.p, 4 * (size_t)n_keep, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(out->occ_len, d_out_len.p, 4 * (size_t)n_keep, hipMemcpyDeviceToHost, s));
    if (n_vtx) DG_HIP(hipMemcpyAsync(out->vpool, d_vpool.p, 4 * (size_t)n_vtx, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    lap("emit + download");
    out->n_occ = n_keep; out->n_vtx = n_vtx; out->n_candidates = n_occ; out->n_unstable_groups = (int64_t)unstable;
    delete c->an;                       // the index is consumed
    c->an = nullptr;
    return DG_OK;
}
