// Device-side construction of the sweep's tables (the default of dg_dp_load_graph since round 3; the host construction in
// dg_dp_tables.hip stays behind option host_tables = 1 and must produce the same bytes: dg_dp_get_table_digest).
//
// The caller's out-CSR + colour CSR go up once (0.4 GB on MHC-24 against 0.9 GB of finished tables before) and everything the
// sweep reads is derived where it is used:
//   in-CSR            one stable radix sort of the out-edges by destination: out-edge order IS (source position asc,
//                     adjacency order asc), the reference's tie order (approximator.cpp:657-659)
//   row records, row in-edge matrices          one thread per vertex
//   column groups, dead columns, heavy rows, slot-block headers, level descriptors    one thread per level, in two passes
//                     (count -> exclusive scans -> fill); the greedy grouping of a level's columns is sequential by nature
//   slot records      one wave per 64-record block
// plus the validation the host construction did (edges to the next level only, weights 0/1, parallel edges of one weight,
// sorted-unique colour lists): kernels clamp what they index with and record the first violation in a word the host reads at
// the one synchronisation in the middle (totals -> allocation sizes).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dg_dp.hpp"

namespace dgi {

namespace {

constexpr int64_t ROWX_MAX_LEVEL_CELLS = (int64_t)1 << 21;   // (same rule as the host construction)
constexpr int64_t ROWX_BUDGET_WORDS = (int64_t)1 << 30;

enum BuildErrCode { BE_OK = 0, BE_OUT_OFF, BE_EDGE_LEVEL, BE_WEIGHT, BE_PARALLEL, BE_COL_OFF, BE_COL_SORT };
struct BuildStat {                                           // one per build, zeroed before the first kernel
    int err, err_a, err_b, pad_;
    unsigned long long max_list, edge_pairs, colour_entries, max_level_cells, max_level_units;
    int max_k2, any_wide;
};

__device__ __forceinline__ void build_fail(BuildStat *st, int code, int a, int b) {
    if (atomicCAS(&st->err, 0, code) == 0) { st->err_a = a; st->err_b = b; }
}

__global__ __launch_bounds__(256) void bt_level_of_kernel(const int32_t *__restrict__ level_off, int L, int nV, int32_t *__restrict__ level_of) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= nV) return;
    int lo = 0, hi = L - 1;                                   // last level whose first vertex <= v
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (level_off[mid] <= v) lo = mid; else hi = mid - 1; }
    level_of[v] = lo;
}

// out-edge e = (u -> v, w): sort key v, value (position of u in its level | w << 31); in-degrees by atomics
__global__ __launch_bounds__(256) void bt_edges_kernel(int nV, int64_t E, const int64_t *__restrict__ out_off, const int32_t *__restrict__ out_dst,
                                                       const uint8_t *__restrict__ out_w, const int32_t *__restrict__ level_of,
                                                       const int32_t *__restrict__ level_off, uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                       uint32_t *__restrict__ indeg, BuildStat *st) {
    const int u = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (u >= nV) return;
    int64_t o0 = out_off[u], o1 = out_off[u + 1];
    if (o0 < 0 || o1 < o0 || o1 > E) { build_fail(st, BE_OUT_OFF, u, 0); o0 = min(max(o0, (int64_t)0), E); o1 = min(max(o1, o0), E); }
    const int lu = level_of[u];
    const uint32_t pos = (uint32_t)(u - level_off[lu]);
    for (int64_t e = o0; e < o1; ++e) {
        int v = out_dst[e];
        const uint32_t w = out_w[e];
        if (v < 0 || v >= nV || level_of[v] != lu + 1) { build_fail(st, BE_EDGE_LEVEL, u, v); v = 0; }
        if (w > 1) build_fail(st, BE_WEIGHT, (int)w, u);
        key[e] = (uint32_t)v;
        val[e] = pos | ((w & 1u) << 31);
        atomicAdd(&indeg[v], 1u);
    }
}

// parallel edges must carry equal weights (always true for graphs built by Approximator::solve), and the
// traffic figure of the score-delta kernel: every in-edge of a coloured transition is paired with the T in-edges of its level
__global__ __launch_bounds__(256) void bt_in_edges_kernel(int64_t E, const uint32_t *__restrict__ in_edge, const int32_t *__restrict__ in_dst,
                                                          const uint32_t *__restrict__ in_off, const int32_t *__restrict__ level_of,
                                                          const int32_t *__restrict__ level_off, const uint8_t *__restrict__ has_col, ColourCsr col,
                                                          BuildStat *st) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long f = 0;
    if (e < E) {
        const int tv = in_dst[e];
        const uint32_t w = in_edge[e];
        if (e > 0 && in_dst[e - 1] == tv && (in_edge[e - 1] & 0x7FFFFFFFu) == (w & 0x7FFFFFFFu) && in_edge[e - 1] != w) build_fail(st, BE_PARALLEL, tv, 0);
        const int l = level_of[tv];
        if (l >= 1 && (has_col[l - 1] | has_col[l])) {
            const int a0 = level_off[l - 1], b0 = level_off[l], b1 = level_off[l + 1];
            const int sv = min(a0 + (int)(w & 0x7FFFFFFFu), b0 - 1);
            const unsigned long long T = in_off[b1] - in_off[b0];
            f = 2ULL * T * (unsigned long long)((col.hom_off[sv + 1] - col.hom_off[sv]) + (col.het_off[sv + 1] - col.het_off[sv]) +
                                                (col.hom_off[tv + 1] - col.hom_off[tv]) + (col.het_off[tv + 1] - col.het_off[tv]));
        }
    }
    for (int s = 32; s > 0; s >>= 1) f += __shfl_down(f, s);
    if ((threadIdx.x & 63) == 0 && f) atomicAdd(&st->colour_entries, f);
}

// row record {first in-edge, in-degree, in-edge 0, in-edge 1}; bit 16 of the two words: "the source vertex has exactly one
// in-edge itself" (the chain walk then skips a back-pointer load)
__global__ __launch_bounds__(256) void bt_rowrec_kernel(int nV, const uint32_t *__restrict__ in_off, const uint32_t *__restrict__ in_edge,
                                                        const int32_t *__restrict__ level_of, const int32_t *__restrict__ level_off, uint4 *__restrict__ rowrec) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= nV) return;
    const uint32_t e0 = in_off[v], dv = in_off[v + 1] - e0;
    const int a0v = level_off[max(0, level_of[v] - 1)];
    uint32_t word[2] = {0, 0};
    for (uint32_t q = 0; q < 2; ++q)
        if (dv > q) {
            word[q] = in_edge[e0 + q];
            const int src = min(a0v + (int)(word[q] & 0x7FFFu), nV - 1);
            if (in_off[src + 1] - in_off[src] == 1) word[q] |= 1u << 16;
        }
    rowrec[v] = make_uint4(e0, dv, word[0], word[1]);
}

// colour lists must be sorted-unique (the merges rely on it) and short enough for uint16 score deltas
__global__ __launch_bounds__(256) void bt_colours_kernel(int nV, ColourCsr col, int64_t n_hom, int64_t n_het, const int32_t *__restrict__ level_of,
                                                         uint8_t *__restrict__ has_col, BuildStat *st) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= nV) return;
    unsigned long long longest = 0;
    bool any = false;
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t *off = pass ? col.het_off : col.hom_off;
        const int32_t *cv = pass ? col.het_col : col.hom_col;
        const int64_t n = pass ? n_het : n_hom;
        int64_t a = off[v], b = off[v + 1];
        if (a < 0 || b < a || b > n) { build_fail(st, BE_COL_OFF, v, 0); a = b = 0; }
        longest = max(longest, (unsigned long long)(b - a));
        any |= b > a;
        for (int64_t q = a + 1; q < b; ++q)
            if (cv[q] <= cv[q - 1]) { build_fail(st, BE_COL_SORT, v, 0); break; }
    }
    if (any) has_col[level_of[v]] = 1;
    if (longest) atomicMax(&st->max_list, longest);
}

// ---- per-level passes -------------------------------------------------------------------------------------------
// quantities scanned over the levels (exclusive prefix sums; entry L = total)
enum { Q_GRP = 0, Q_DEAD, Q_HEAVY, Q_BLOCKS, Q_ROWX, Q_UNITS, Q_CELLS, Q_DELTA, Q_DBLK, Q_DTRANS, NQ };

struct LevelGeom {                                         // what both passes derive from the level's in-degrees
    int a0, k, b0, k2;
    uint32_t in_base, T, max_indeg;
    int n_heavy, ngroups, ndead, nblocks, fast_ok, bp_wide;
};

// walks the columns of level l: on_col(c, dv) for every column, on_group(gb0, ge0, maxdv) when a group closes.
// Column groups: greedy runs of whole columns with <= 64 in-edges; a column with more gets its own group.
template <class FC, class FG>
__device__ __forceinline__ void walk_level(const uint32_t *__restrict__ in_off, int b0, int k2, FC &&on_col, FG &&on_group) {
    uint32_t cur_size = 0, gb0 = 0, maxdv = 1;
    bool open = false;
    uint32_t e0 = in_off[b0];
    for (int c = 0; c < k2; ++c) {
        const uint32_t e1 = in_off[b0 + c + 1], dv = e1 - e0;
        on_col(c, dv);
        if (dv != 0) {
            if (cur_size == 0 || cur_size + dv > 64 || dv > 64) {
                if (open) on_group(gb0, e0, maxdv);
                gb0 = e0; cur_size = 0; maxdv = 1; open = true;
            }
            cur_size += dv;
            maxdv = max(maxdv, dv);
            if (dv > 64) cur_size = 65;                        // force a new group after a giant column
        }
        e0 = e1;
    }
    if (open) on_group(gb0, e0, maxdv);
}

__device__ __forceinline__ uint32_t group_tag_steps(uint32_t maxdv) {
    uint32_t steps = 0;
    while ((1u << steps) < min(maxdv, 64u)) ++steps;
    return steps;
}

__device__ __forceinline__ LevelGeom level_geom(const int32_t *__restrict__ level_off, const uint32_t *__restrict__ in_off, int l) {
    LevelGeom G;
    G.a0 = level_off[l - 1]; G.k = level_off[l] - G.a0;
    G.b0 = level_off[l]; G.k2 = level_off[l + 1] - G.b0;
    G.in_base = in_off[G.b0];
    G.T = in_off[G.b0 + G.k2] - G.in_base;
    G.max_indeg = 0; G.n_heavy = 0; G.ngroups = 0; G.ndead = 0; G.nblocks = 0;
    bool giant = false;
    walk_level(in_off, G.b0, G.k2,
               [&](int, uint32_t dv) { G.max_indeg = max(G.max_indeg, dv); G.n_heavy += dv > (uint32_t)COOP_MIN; G.ndead += dv == 0; },
               [&](uint32_t gb0, uint32_t ge0, uint32_t) { ++G.ngroups; const bool g = ge0 - gb0 > 64; giant |= g; G.nblocks += g ? (int)((ge0 - gb0 + 63) / 64) : 1; });
    G.fast_ok = G.T < (1u << 20) ? 1 : 0;                       // the slot word keeps 20 bits of in-edge index
    G.bp_wide = G.max_indeg > (uint32_t)BP_MAX_RANK ? 1 : 0;    // ranks do not fit 8 bits: wide words, generic kernel
    if (G.bp_wide) G.fast_ok = 0;
    if (giant && G.fast_ok) G.fast_ok = 2;                      // the general sweep variant
    if (G.nblocks == 0) G.nblocks = 1;                          // level without in-edges: one all-padding block
    return G;
}

__device__ __forceinline__ bool level_has_rowx(const LevelGeom &G, int64_t ncell, int use_rowx) {
    return use_rowx && G.fast_ok == 1 && G.max_indeg > 2 && G.max_indeg <= (uint32_t)ROWX_MAX && ncell <= ROWX_MAX_LEVEL_CELLS;
}

__global__ __launch_bounds__(128) void bt_level_count_kernel(int L, int RP, int use_rowx, const int32_t *__restrict__ level_off, const uint32_t *__restrict__ in_off,
                                                             const uint8_t *__restrict__ has_col, int64_t *__restrict__ cnt /* [NQ][L + 1] */, BuildStat *st) {
    const int l = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (l > L) return;
    const size_t LS = (size_t)L + 1;
    if (l == 0 || l == L) { for (int q = 0; q < NQ; ++q) cnt[q * LS + l] = 0; return; }
    const LevelGeom G = level_geom(level_off, in_off, l);
    const int64_t ncell = (int64_t)G.k2 * G.k2 * RP;
    const int64_t units = ((G.bp_wide ? 2 * ncell : ncell) + 1) & ~(int64_t)1;
    const bool coloured = has_col[l - 1] | has_col[l];
    const int64_t tt = (int64_t)G.T * G.T;
    cnt[Q_GRP * LS + l] = max(G.ngroups, 1) + 1;               // group starts + the end sentinel (a level without in-edges: one empty group)
    cnt[Q_DEAD * LS + l] = G.ndead;
    cnt[Q_HEAVY * LS + l] = G.n_heavy;
    cnt[Q_BLOCKS * LS + l] = G.nblocks;
    cnt[Q_ROWX * LS + l] = level_has_rowx(G, ncell, use_rowx) ? (int64_t)G.k2 * G.max_indeg : 0;
    cnt[Q_UNITS * LS + l] = units;
    cnt[Q_CELLS * LS + l] = ncell;
    cnt[Q_DELTA * LS + l] = coloured ? tt : 0;
    cnt[Q_DBLK * LS + l] = coloured ? (tt + DELTA_PER_BLOCK - 1) / DELTA_PER_BLOCK : 0;
    cnt[Q_DTRANS * LS + l] = coloured ? 1 : 0;
    atomicAdd(&st->edge_pairs, (unsigned long long)tt);
    atomicMax(&st->max_level_cells, (unsigned long long)ncell);
    atomicMax(&st->max_level_units, (unsigned long long)units);
    atomicMax(&st->max_k2, G.k2);
    if (G.bp_wide) st->any_wide = 1;
}

struct FillOut {
    LevelDesc *descs;
    uint32_t *grp_begin;
    int32_t *dead_cols, *heavy_rows, *dtrans;
    int64_t *dblk_first;
    uint4 *heads;                                          // per slot block: {first in-edge, end of the group, in_base, tag}
    int32_t *head_b0;
    int64_t rowx_words;                                    // budgeted size of the row in-edge matrices
};

__global__ __launch_bounds__(128) void bt_level_fill_kernel(int L, int RP, int use_rowx, const int32_t *__restrict__ level_off, const uint32_t *__restrict__ in_off,
                                                            const uint8_t *__restrict__ has_col, const int64_t *__restrict__ pre /* [NQ][L + 1] exclusive */, FillOut O) {
    const int l = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (l >= L) return;
    LevelDesc d;
    memset(&d, 0, sizeof d);
    if (l == 0) { O.descs[0] = d; return; }
    const size_t LS = (size_t)L + 1;
    const LevelGeom G = level_geom(level_off, in_off, l);
    const int64_t ncell = (int64_t)G.k2 * G.k2 * RP;
    d.a0 = G.a0; d.k = G.k; d.b0 = G.b0; d.k2 = G.k2; d.in_base = G.in_base; d.T = (int32_t)G.T;
    d.grp_first = (int32_t)pre[Q_GRP * LS + l];
    d.dead_first = (int32_t)pre[Q_DEAD * LS + l]; d.ndead = G.ndead;
    d.heavy_first = (int32_t)pre[Q_HEAVY * LS + l]; d.n_heavy = G.n_heavy;
    d.slot_first = pre[Q_BLOCKS * LS + l] * 64;
    d.fast_ok = G.fast_ok; d.nblocks = G.nblocks; d.bp_wide = G.bp_wide; d.bp_nt = 0;
    d.ngroups = max(G.ngroups, 1);
    d.bp_off = pre[Q_UNITS * LS + l];
    const bool coloured = has_col[l - 1] | has_col[l];
    d.delta_off = coloured ? DELTA_PAD + pre[Q_DELTA * LS + l] : -1;
    d.rowx_off = pre[Q_ROWX * LS + l];
    d.rowx_stride = level_has_rowx(G, ncell, use_rowx) && d.rowx_off + (int64_t)G.k2 * G.max_indeg <= O.rowx_words ? (int32_t)G.max_indeg : 0;
    d.dmax = (int32_t)G.max_indeg;
    for (int q = 0; q < HEAVY_INLINE; ++q) d.heavy_in[q] = (int16_t)-1;
    if (coloured) {
        const int64_t t = pre[Q_DTRANS * LS + l];
        O.dtrans[t] = l;
        O.dblk_first[t] = pre[Q_DBLK * LS + l];
    }
    uint32_t *gb = O.grp_begin + d.grp_first;
    int32_t *dead = O.dead_cols + d.dead_first, *heavy = O.heavy_rows + d.heavy_first;
    int ng = 0, nd = 0, nh = 0;
    int64_t blk = pre[Q_BLOCKS * LS + l];
    walk_level(in_off, G.b0, G.k2,
               [&](int c, uint32_t dv) {
                   if (dv == 0) dead[nd++] = c;
                   if (dv > (uint32_t)COOP_MIN) { if (nh < HEAVY_INLINE) d.heavy_in[nh] = (int16_t)c; heavy[nh++] = c; }
               },
               [&](uint32_t gb0, uint32_t ge0, uint32_t maxdv) {
                   gb[ng++] = gb0;
                   const bool giant = ge0 - gb0 > 64;
                   const uint32_t nb = giant ? (ge0 - gb0 + 63) / 64 : 1, steps = group_tag_steps(maxdv);
                   for (uint32_t bq = 0; bq < nb; ++bq, ++blk) {
                       O.heads[blk] = make_uint4(gb0 + bq * 64, ge0, G.in_base, giant ? (bq == 0 ? 15u : 14u) : steps);
                       O.head_b0[blk] = G.b0;
                   }
               });
    gb[ng++] = G.in_base + G.T;                                // sentinel: end of the level's in-edges
    if (G.ngroups == 0) {                                      // level without in-edges: one empty group, one all-padding block
        gb[ng++] = G.in_base + G.T;
        O.heads[blk] = make_uint4(G.in_base + G.T, G.in_base + G.T, G.in_base, 0u);
        O.head_b0[blk] = G.b0;
    }
    O.descs[l] = d;
}

// 64-wide slot table of the fast kernel: record q of a block = in-edge e_first + q of its group
//   {source position | weight << 15 | destination column << 16,  in-edge index in the level | its rank in the column << 20 | tag << 28}
__global__ __launch_bounds__(256) void bt_slots_kernel(int64_t n_blocks, const uint4 *__restrict__ heads, const int32_t *__restrict__ head_b0,
                                                       const uint32_t *__restrict__ in_edge, const int32_t *__restrict__ in_dst, const uint32_t *__restrict__ in_off,
                                                       uint2 *__restrict__ slots) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const uint4 h = heads[b];
    const int b0 = head_b0[b];
    const uint32_t e = h.x + (threadIdx.x & 63);
    uint2 r;
    if (e < h.y) {
        const uint32_t pv = in_edge[e];
        const int tv = in_dst[e];
        r.x = (pv & 0x7FFFu) | ((pv >> 31) << 15) | ((uint32_t)(tv - b0) << 16);
        r.y = (e - h.z) | ((e - in_off[tv]) << 20) | (h.w << 28);
    } else {
        r.x = 0xFFFFFFFFu;
        r.y = h.w << 28;
    }
    slots[b * 64 + (threadIdx.x & 63)] = r;
}

// row in-edge matrices: in-edge words of every destination row padded to the level's largest in-degree (zeroed before)
__global__ __launch_bounds__(256) void bt_rowx_kernel(int nV, const int32_t *__restrict__ level_of, const LevelDesc *__restrict__ descs,
                                                      const uint32_t *__restrict__ in_off, const uint32_t *__restrict__ in_edge, uint32_t *__restrict__ rowx) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= nV) return;
    const int l = level_of[v];
    if (l < 1) return;
    const int stride = descs[l].rowx_stride;
    if (stride <= 0) return;
    uint32_t *rx = rowx + descs[l].rowx_off + (int64_t)(v - descs[l].b0) * stride;
    const uint32_t e0 = in_off[v], dv = min(in_off[v + 1] - e0, (uint32_t)stride);
    for (uint32_t q = 0; q < dv; ++q) rx[q] = in_edge[e0 + q];
}

unsigned nblk(int64_t n, int per) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }

int up(DevBuf &b, const void *src, size_t bytes, hipStream_t s) {
    if (int rc = b.ensure(bytes)) return rc;
    if (bytes) DG_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
    return DG_OK;
}

}  // namespace

// Builds every table of the sweep on the device.  On return the d_* buffers of S are complete, S.descs / dtrans / dblk_first
// are on the host as well and the counters of S are set; nothing is synchronised beyond the two downloads.
int dp_build_tables_device(dg_ctx *c, const dg_dp_graph *g, DpState &S, std::vector<int32_t> &dtrans, std::vector<int64_t> &dblk_first, int &max_k) {
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    double tl0 = wall_s();
    auto lap = [&](const char *what) { if (dbg) { (void)hipStreamSynchronize(c->stream); double t = wall_s(); fprintf(stderr, "[dipgenie_hip] load: %-22s %.3f s\n", what, t - tl0); tl0 = t; } };
    const int nV = S.nV, L = S.L;
    hipStream_t s = c->stream;
    for (int l = 0; l < L; ++l)
        if (g->level_off[l + 1] <= g->level_off[l]) { set_error("level %d is empty", l); return DG_ERR_ARG; }
    if (g->out_off[0] != 0) { set_error("out_off must start at 0"); return DG_ERR_ARG; }
    if (g->hom_off[0] != 0 || g->het_off[0] != 0) { set_error("colour offsets must start at 0"); return DG_ERR_ARG; }
    const int64_t E = g->out_off[nV], n_hom = g->hom_off[nV], n_het = g->het_off[nV];
    if (E < 0 || E >= (int64_t)1 << 31) { set_error("unsupported number of edges (%lld)", (long long)E); return DG_ERR_UNSUPPORTED; }
    if (n_hom < 0 || n_het < 0) { set_error("colour offsets must be non-negative"); return DG_ERR_ARG; }
    // ---- uploads: the caller's graph as it is
    DevBuf t_level_off, t_out_off, t_out_dst, t_out_w, t_level_of, t_key, t_val, t_indeg, t_hascol, t_cnt, t_pre, t_stat, t_tmp, t_heads, t_head_b0;
    if (int rc = up(t_level_off, g->level_off, 4 * ((size_t)L + 1), s)) return rc;
    if (int rc = up(t_out_off, g->out_off, 8 * ((size_t)nV + 1), s)) return rc;
    if (int rc = up(t_out_dst, g->out_dst, 4 * (size_t)E, s)) return rc;
    if (int rc = up(t_out_w, g->out_w, (size_t)E, s)) return rc;
    if (int rc = up(S.d_hom_off, g->hom_off, 8 * ((size_t)nV + 1), s)) return rc;
    if (int rc = up(S.d_het_off, g->het_off, 8 * ((size_t)nV + 1), s)) return rc;
    if (int rc = up(S.d_hom_col, g->hom_col, 4 * (size_t)n_hom, s)) return rc;
    if (int rc = up(S.d_het_col, g->het_col, 4 * (size_t)n_het, s)) return rc;
    if (int rc = t_level_of.ensure(4 * (size_t)nV)) return rc;
    if (int rc = t_key.ensure(4 * (size_t)E + 16)) return rc;
    if (int rc = t_val.ensure(4 * (size_t)E + 16)) return rc;
    if (int rc = t_indeg.ensure(4 * ((size_t)nV + 1))) return rc;
    if (int rc = t_hascol.ensure((size_t)L + 1)) return rc;
    if (int rc = t_stat.ensure(sizeof(BuildStat))) return rc;
    if (int rc = S.d_in_off.ensure(4 * ((size_t)nV + 1))) return rc;
    if (int rc = S.d_in_edge.ensure(4 * (size_t)E + 16)) return rc;
    if (int rc = S.d_in_dst.ensure(4 * (size_t)E + 16)) return rc;
    if (int rc = S.d_rowrec.ensure(16 * (size_t)nV)) return rc;
    DG_HIP(hipMemsetAsync(t_indeg.p, 0, 4 * ((size_t)nV + 1), s));
    DG_HIP(hipMemsetAsync(t_hascol.p, 0, (size_t)L + 1, s));
    DG_HIP(hipMemsetAsync(t_stat.p, 0, sizeof(BuildStat), s));
    lap("uploads");
    const int32_t *level_off = t_level_off.as<int32_t>(), *level_of = t_level_of.as<int32_t>();
    BuildStat *st = t_stat.as<BuildStat>();
    const ColourCsr col = colour_csr(S);
    uint32_t *in_off = S.d_in_off.as<uint32_t>(), *in_edge = S.d_in_edge.as<uint32_t>();
    int32_t *in_dst = S.d_in_dst.as<int32_t>();
    // ---- in-CSR
    hipLaunchKernelGGL(bt_level_of_kernel, dim3(nblk(nV, 256)), dim3(256), 0, s, level_off, L, nV, t_level_of.as<int32_t>());
    hipLaunchKernelGGL(bt_edges_kernel, dim3(nblk(nV, 256)), dim3(256), 0, s, nV, E, t_out_off.as<int64_t>(), t_out_dst.as<int32_t>(), t_out_w.as<uint8_t>(), level_of,
                       level_off, t_key.as<uint32_t>(), t_val.as<uint32_t>(), t_indeg.as<uint32_t>(), st);
    {
        size_t tb = 0;
        DG_HIP(rocprim::exclusive_scan(nullptr, tb, t_indeg.as<uint32_t>(), in_off, 0u, (size_t)nV + 1, rocprim::plus<uint32_t>(), s));
        if (int rc = t_tmp.ensure(tb)) return rc;
        DG_HIP(rocprim::exclusive_scan(t_tmp.p, tb, t_indeg.as<uint32_t>(), in_off, 0u, (size_t)nV + 1, rocprim::plus<uint32_t>(), s));
    }
    if (E > 0) {
        int end_bit = 1;
        while (end_bit < 32 && ((int64_t)1 << end_bit) < nV) ++end_bit;
        size_t tb = 0;
        DG_HIP(rocprim::radix_sort_pairs(nullptr, tb, t_key.as<uint32_t>(), (uint32_t *)in_dst, t_val.as<uint32_t>(), in_edge, (size_t)E, 0, end_bit, s));
        if (int rc = t_tmp.ensure(tb)) return rc;
        DG_HIP(rocprim::radix_sort_pairs(t_tmp.p, tb, t_key.as<uint32_t>(), (uint32_t *)in_dst, t_val.as<uint32_t>(), in_edge, (size_t)E, 0, end_bit, s));
    }
    hipLaunchKernelGGL(bt_colours_kernel, dim3(nblk(nV, 256)), dim3(256), 0, s, nV, col, n_hom, n_het, level_of, t_hascol.as<uint8_t>(), st);
    hipLaunchKernelGGL(bt_in_edges_kernel, dim3(nblk(E, 256)), dim3(256), 0, s, E, in_edge, in_dst, in_off, level_of, level_off, t_hascol.as<uint8_t>(), col, st);
    hipLaunchKernelGGL(bt_rowrec_kernel, dim3(nblk(nV, 256)), dim3(256), 0, s, nV, in_off, in_edge, level_of, level_off, S.d_rowrec.as<uint4>());
    // ---- level pass 1: counts -> exclusive scans
    const size_t LS = (size_t)L + 1;
    if (int rc = t_cnt.ensure(8 * LS * NQ)) return rc;
    if (int rc = t_pre.ensure(8 * LS * NQ)) return rc;
    hipLaunchKernelGGL(bt_level_count_kernel, dim3(nblk((int64_t)L + 1, 128)), dim3(128), 0, s, L, S.RP, (int)(S.use_rowx != 0), level_off, in_off, t_hascol.as<uint8_t>(),
                       t_cnt.as<int64_t>(), st);
    for (int q = 0; q < NQ; ++q) {
        size_t tb = 0;
        DG_HIP(rocprim::exclusive_scan(nullptr, tb, t_cnt.as<int64_t>() + q * LS, t_pre.as<int64_t>() + q * LS, (int64_t)0, LS, rocprim::plus<int64_t>(), s));
        if (int rc = t_tmp.ensure(tb)) return rc;
        DG_HIP(rocprim::exclusive_scan(t_tmp.p, tb, t_cnt.as<int64_t>() + q * LS, t_pre.as<int64_t>() + q * LS, (int64_t)0, LS, rocprim::plus<int64_t>(), s));
    }
    int64_t tot[NQ];
    BuildStat hs;
    for (int q = 0; q < NQ; ++q) DG_HIP(hipMemcpyAsync(&tot[q], t_pre.as<int64_t>() + q * LS + L, 8, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(&hs, st, sizeof hs, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    lap("in-CSR + level counts");
    if (hs.err != BE_OK) {
        switch (hs.err) {
        case BE_OUT_OFF: set_error("out_off not monotone at %d", hs.err_a); return DG_ERR_ARG;
        case BE_EDGE_LEVEL: set_error("edge %d->%d does not go to the next level", hs.err_a, hs.err_b); return DG_ERR_ARG;
        case BE_WEIGHT: set_error("edge weight %d > 1", hs.err_a); return DG_ERR_ARG;
        case BE_PARALLEL: set_error("parallel edges with different weights into vertex %d: tie order would be schedule dependent", hs.err_a); return DG_ERR_UNSUPPORTED;
        case BE_COL_OFF: set_error("colour offsets not monotone at %d", hs.err_a); return DG_ERR_ARG;
        default: set_error("colour list of vertex %d is not sorted-unique", hs.err_a); return DG_ERR_ARG;
        }
    }
    if ((int64_t)hs.max_list * 4 > 65535) { set_error("colour lists too long for uint16 score deltas (%lld)", (long long)hs.max_list); return DG_ERR_UNSUPPORTED; }
    max_k = std::max(hs.max_k2, 1);
    if (max_k > MAX_K) { set_error("level width %d exceeds the supported %d", max_k, MAX_K); return DG_ERR_UNSUPPORTED; }
    if (tot[Q_GRP] >= (int64_t)1 << 31 || tot[Q_DEAD] >= (int64_t)1 << 31) { set_error("group tables too large"); return DG_ERR_UNSUPPORTED; }
    if (tot[Q_DBLK] >= (int64_t)1 << 31) { set_error("delta grid too large"); return DG_ERR_UNSUPPORTED; }
    S.cells = (uint64_t)tot[Q_CELLS];
    S.edge_pairs = hs.edge_pairs; S.colour_entries = hs.colour_entries;
    S.relaxations = S.edge_pairs * (uint64_t)S.RP;
    S.total_units = tot[Q_UNITS];
    S.max_level_units = (int64_t)hs.max_level_units;
    S.max_level_cells = std::max<int64_t>(S.RP, (int64_t)hs.max_level_cells);
    S.delta_entries = DELTA_PAD + tot[Q_DELTA];
    S.n_delta_blocks = tot[Q_DBLK];
    S.lean_chain = S.use_lean_chain && !hs.any_wide && S.max_level_cells < ((int64_t)1 << 30) && (int64_t)S.RP * max_k < ((int64_t)1 << 24) && nV < (1 << 27);
    const int64_t rowx_words = std::min(tot[Q_ROWX], ROWX_BUDGET_WORDS);
    S.n_grp = tot[Q_GRP]; S.n_dead = tot[Q_DEAD]; S.n_heavy_rows = tot[Q_HEAVY]; S.n_slot_records = tot[Q_BLOCKS] * 64; S.n_rowx_words = rowx_words;
    // ---- level pass 2: fill
    const int64_t n_dt = tot[Q_DTRANS], n_blocks = tot[Q_BLOCKS];
    if (int rc = S.d_descs.ensure(sizeof(LevelDesc) * (size_t)L)) return rc;
    if (int rc = S.d_grp.ensure(4 * (size_t)tot[Q_GRP] + 16)) return rc;
    if (int rc = S.d_dead.ensure(4 * (size_t)tot[Q_DEAD] + 16)) return rc;
    if (int rc = S.d_heavy.ensure(4 * ((size_t)tot[Q_HEAVY] + 1))) return rc;
    if (int rc = S.d_dtrans.ensure(4 * (size_t)n_dt + 16)) return rc;
    if (int rc = S.d_dblk_first.ensure(8 * (size_t)n_dt + 16)) return rc;
    if (int rc = S.d_slots.ensure(8 * 64 * (size_t)n_blocks)) return rc;
    if (int rc = S.d_rowx.ensure(4 * ((size_t)rowx_words + 4))) return rc;
    if (int rc = t_heads.ensure(16 * (size_t)n_blocks)) return rc;
    if (int rc = t_head_b0.ensure(4 * (size_t)n_blocks)) return rc;
    DG_HIP(hipMemsetAsync(S.d_rowx.p, 0, 4 * ((size_t)rowx_words + 4), s));
    DG_HIP(hipMemsetAsync(S.d_heavy.p, 0, 4 * ((size_t)tot[Q_HEAVY] + 1), s));
    FillOut O{S.d_descs.as<LevelDesc>(), S.d_grp.as<uint32_t>(), S.d_dead.as<int32_t>(), S.d_heavy.as<int32_t>(), S.d_dtrans.as<int32_t>(), S.d_dblk_first.as<int64_t>(),
              t_heads.as<uint4>(), t_head_b0.as<int32_t>(), rowx_words};
    hipLaunchKernelGGL(bt_level_fill_kernel, dim3(nblk(L, 128)), dim3(128), 0, s, L, S.RP, (int)(S.use_rowx != 0), level_off, in_off, t_hascol.as<uint8_t>(),
                       t_pre.as<int64_t>(), O);
    hipLaunchKernelGGL(bt_slots_kernel, dim3(nblk(n_blocks, 4)), dim3(256), 0, s, n_blocks, t_heads.as<uint4>(), t_head_b0.as<int32_t>(), in_edge, in_dst, in_off,
                       S.d_slots.as<uint2>());
    if (rowx_words > 0)
        hipLaunchKernelGGL(bt_rowx_kernel, dim3(nblk(nV, 256)), dim3(256), 0, s, nV, level_of, S.d_descs.as<LevelDesc>(), in_off, in_edge, S.d_rowx.as<uint32_t>());
    S.descs.resize(L);
    dtrans.resize((size_t)n_dt);
    dblk_first.resize((size_t)n_dt);
    DG_HIP(hipMemcpyAsync(S.descs.data(), S.d_descs.p, sizeof(LevelDesc) * (size_t)L, hipMemcpyDeviceToHost, s));
    if (n_dt) {
        DG_HIP(hipMemcpyAsync(dtrans.data(), S.d_dtrans.p, 4 * (size_t)n_dt, hipMemcpyDeviceToHost, s));
        DG_HIP(hipMemcpyAsync(dblk_first.data(), S.d_dblk_first.p, 8 * (size_t)n_dt, hipMemcpyDeviceToHost, s));
    }
    DG_HIP(hipStreamSynchronize(s));
    DG_HIP(hipGetLastError());
    S.level_units.assign(L, 0);
    S.level_dmax.assign(L, 0);
    for (int l = 1; l < L; ++l) {
        const LevelDesc &d = S.descs[l];
        const int64_t ncell = (int64_t)d.k2 * d.k2 * S.RP;
        S.level_units[l] = ((d.bp_wide ? 2 * ncell : ncell) + 1) & ~(int64_t)1;
        S.level_dmax[l] = d.dmax;
    }
    lap("fill + descriptors back");
    return DG_OK;                                              // the temporaries (out-CSR copy, sort buffers, counts) are released here
}

}  // namespace dgi
