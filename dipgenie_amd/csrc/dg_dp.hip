// libdipgenie_hip.so -- diploid pair-of-paths DP for MI355X (gfx950).
//
// Replaces the level loop + sink read-out of Approximator::diploid_dp_approximation_solver
// (/root/reference/src/approximator.cpp:532-716, 757-785).  Design:
//   * gather form: one work-item per destination cell (i2, j2, r2) of level l+1, reducing over
//     in(u2) x in(v2); in-edges are stored sorted by source position, so a strict '>' scan
//     reproduces the reference's take-if total order (value desc, pred_i asc, pred_j asc, :657-659)
//     without locks or atomics, and no destination is ever "reset" (:565-576 disappears);
//   * rolling value state is 4 B/cell ([i][j][r], r fastest -> lanes of a wave read consecutive
//     addresses); s_het / edge chains are not carried (reference cell = 40 B).  Instead every cell
//     streams one 4-byte back-pointer (pred_i | pred_j<<15 | wu<<30 | wv<<31) to HBM and a
//     traceback kernel walks the lattice from the sink, emitting the weighted-edge lists (:757-764,
//     :673-692) and re-deriving s_het from the colour lists of the L winning edge pairs;
//   * score deltas (:604-624) do not depend on r nor on other levels: one launch fills, for every
//     transition that touches a colour, the T x T matrix delta[e_u][e_v] (uint16), T = #in-edges of
//     the destination level; colourless transitions (73 % on MHC_4) skip the lookup.
#include <algorithm>
#include <cstring>

#include "dg_internal.hpp"

namespace dgi {

constexpr int32_t NEG_INF = INT32_MIN / 4;              // approximator.cpp:413
constexpr uint32_t BP_NONE = 0xFFFFFFFFu;
constexpr int MAX_K = 1 << 15;                          // back-pointer packs positions in 15 bits

struct LevelDesc {                                      // transition (l-1) -> l, indexed by l
    int32_t a0, k;                                      // source level: first vertex id, width
    int32_t b0, k2;                                     // destination level
    uint32_t in_base;                                   // first in-edge of the destination level
    int32_t T;                                          // in-edges into the destination level
    int64_t delta_off;                                  // offset of the T*T uint16 matrix, -1 if all zero
    int64_t bp_off;                                     // offset of this level's cells in the bp lattice
};

struct TraceOut {
    int32_t value, s_het, n_p1, n_p2, overflow, pad[3];
};

struct DpState {
    int32_t nV = 0, L = 0, R = 0, RP = 0, cap = 0;
    bool loaded = false;
    int64_t want_digest = 0, use_graph = 1, max_blocks = 2048;
    std::vector<LevelDesc> descs;
    uint64_t cells = 0, relaxations = 0, edge_pairs = 0, colour_entries = 0;
    int64_t total_cells = 0, max_level_cells = 0, delta_entries = 0, n_delta_blocks = 0;
    DevBuf d_descs, d_in_off, d_in_edge, d_in_dst, d_hom_off, d_het_off, d_hom_col, d_het_col;
    DevBuf d_delta, d_bp, d_val[2], d_digest, d_trace, d_edges, d_dblk_first, d_dtrans;
    std::vector<uint64_t> digest_host;
    dg_dp_timing timing;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

void dp_state_free(DpState *s) {
    if (!s) return;
    for (auto &e : s->ev) if (e) (void)hipEventDestroy(e);
    delete s;
}

// ---------------------------------------------------------------------------------------------
// set-ops on sorted colour lists (approximator.cpp:269-311), 4-way merge without building unions
// ---------------------------------------------------------------------------------------------
template <bool SYMDIFF>
__device__ __forceinline__ int union2x2(const int32_t *A, int na, const int32_t *B, int nb,
                                        const int32_t *C, int nc, const int32_t *D, int nd) {
    if (SYMDIFF) { if ((na | nb | nc | nd) == 0) return 0; }
    else { if ((na | nb) == 0 || (nc | nd) == 0) return 0; }
    int i = 0, j = 0, k = 0, m = 0, cnt = 0;
    while (i < na || j < nb || k < nc || m < nd) {
        int x = INT32_MAX;
        if (i < na) x = min(x, A[i]);
        if (j < nb) x = min(x, B[j]);
        if (k < nc) x = min(x, C[k]);
        if (m < nd) x = min(x, D[m]);
        bool inL = false, inR = false;
        while (i < na && A[i] == x) { inL = true; ++i; }
        while (j < nb && B[j] == x) { inL = true; ++j; }
        while (k < nc && C[k] == x) { inR = true; ++k; }
        while (m < nd && D[m] == x) { inR = true; ++m; }
        if (SYMDIFF ? (inL != inR) : (inL && inR)) ++cnt;
    }
    return cnt;
}

struct ColourCsr { const int64_t *hom_off, *het_off; const int32_t *hom_col, *het_col; };

__device__ __forceinline__ int score_inter(const ColourCsr &c, int u1, int v1, int u2, int v2) {
    const int64_t a = c.hom_off[u1], b = c.hom_off[v1], d = c.hom_off[u2], e = c.hom_off[v2];
    return union2x2<false>(c.hom_col + a, (int)(c.hom_off[u1 + 1] - a), c.hom_col + b, (int)(c.hom_off[v1 + 1] - b),
                           c.hom_col + d, (int)(c.hom_off[u2 + 1] - d), c.hom_col + e, (int)(c.hom_off[v2 + 1] - e));
}
__device__ __forceinline__ int score_symd(const ColourCsr &c, int u1, int v1, int u2, int v2) {
    const int64_t a = c.het_off[u1], b = c.het_off[v1], d = c.het_off[u2], e = c.het_off[v2];
    return union2x2<true>(c.het_col + a, (int)(c.het_off[u1 + 1] - a), c.het_col + b, (int)(c.het_off[v1 + 1] - b),
                          c.het_col + d, (int)(c.het_off[u2 + 1] - d), c.het_col + e, (int)(c.het_off[v2 + 1] - e));
}

// ---------------------------------------------------------------------------------------------
// score-delta precompute: delta[e_u][e_v] = inter + symd for every coloured transition
// One block handles DELTA_PER_BLOCK consecutive entries of one transition's T*T matrix.
// ---------------------------------------------------------------------------------------------
constexpr int DELTA_PER_BLOCK = 256 * 8;

__global__ __launch_bounds__(256) void dp_delta_kernel(const LevelDesc *__restrict__ descs,
                                                       const int32_t *__restrict__ dtrans,      // coloured transition -> level
                                                       const int64_t *__restrict__ dblk_first,  // first block of each coloured transition
                                                       int n_dtrans, const uint32_t *__restrict__ in_edge,
                                                       const int32_t *__restrict__ in_dst, ColourCsr col,
                                                       uint16_t *__restrict__ delta) {
    __shared__ int s_t;
    if (threadIdx.x == 0) {   // binary search: last transition whose first block <= blockIdx.x
        int lo = 0, hi = n_dtrans - 1;
        const int64_t b = blockIdx.x;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (dblk_first[mid] <= b) lo = mid; else hi = mid - 1; }
        s_t = lo;
    }
    __syncthreads();
    const int t = s_t;
    const LevelDesc d = descs[dtrans[t]];
    const int64_t n = (int64_t)d.T * d.T;
    const int64_t first = ((int64_t)blockIdx.x - dblk_first[t]) * DELTA_PER_BLOCK;
    uint16_t *out = delta + d.delta_off;
    for (int q = 0; q < DELTA_PER_BLOCK / 256; ++q) {
        const int64_t e = first + q * 256 + threadIdx.x;
        if (e >= n) break;
        const int eu = (int)(e / d.T), ev = (int)(e - (int64_t)eu * d.T);
        const uint32_t pu = in_edge[d.in_base + eu], pv = in_edge[d.in_base + ev];
        const int u1 = d.a0 + (int)(pu & 0x7FFFFFFFu), v1 = d.a0 + (int)(pv & 0x7FFFFFFFu);
        const int u2 = in_dst[d.in_base + eu], v2 = in_dst[d.in_base + ev];
        const int sc = score_inter(col, u1, v1, u2, v2) + score_symd(col, u1, v1, u2, v2);
        out[e] = (uint16_t)sc;
    }
}

// ---------------------------------------------------------------------------------------------
// level sweep, whole-chip form: one launch per transition, grid-stride over destination cells
// ---------------------------------------------------------------------------------------------
template <bool DIGEST>
__global__ __launch_bounds__(256) void dp_level_kernel(const LevelDesc *__restrict__ descs, int lvl, int RP,
                                                       const uint32_t *__restrict__ in_off,
                                                       const uint32_t *__restrict__ in_edge,
                                                       const uint16_t *__restrict__ delta,
                                                       const int32_t *__restrict__ cur, int32_t *__restrict__ nxt,
                                                       uint32_t *__restrict__ bp, unsigned long long *digest) {
    const LevelDesc d = descs[lvl];
    const int64_t ncell = (int64_t)d.k2 * d.k2 * RP;
    const bool has_delta = d.delta_off >= 0;
    const uint16_t *dm = delta + (has_delta ? d.delta_off : 0);
    const int64_t rowstride = (int64_t)d.k * RP;
    unsigned long long dsum = 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < ncell; t += (int64_t)gridDim.x * 256) {
        const int p = (int)(t / RP), r2 = (int)(t - (int64_t)p * RP);
        const int i2 = p / d.k2, j2 = p - i2 * d.k2;
        const uint32_t eu0 = in_off[d.b0 + i2], eu1 = in_off[d.b0 + i2 + 1];
        const uint32_t ev0 = in_off[d.b0 + j2], ev1 = in_off[d.b0 + j2 + 1];
        int best = NEG_INF;
        uint32_t bpv = BP_NONE;
        for (uint32_t eu = eu0; eu < eu1; ++eu) {
            const uint32_t pu = in_edge[eu];
            const int i = (int)(pu & 0x7FFFFFFFu), wu = (int)(pu >> 31);
            const int32_t *row = cur + (int64_t)i * rowstride;
            const uint16_t *drow = dm + (int64_t)(eu - d.in_base) * d.T;
            for (uint32_t ev = ev0; ev < ev1; ++ev) {
                const uint32_t pv = in_edge[ev];
                const int j = (int)(pv & 0x7FFFFFFFu), wv = (int)(pv >> 31);
                const int r = r2 - wu - wv;
                if (r < 0) continue;                                   // r2 = r + wu + wv <= R  (:646-647)
                const int val = row[(int64_t)j * RP + r];
                if (val == NEG_INF) continue;                          // :633
                const int cand = val + (has_delta ? (int)drow[ev - d.in_base] : 0);
                if (cand > best) { best = cand; bpv = (uint32_t)i | ((uint32_t)j << 15) | ((uint32_t)wu << 30) | ((uint32_t)wv << 31); }
            }
        }
        nxt[t] = best;
        bp[d.bp_off + t] = bpv;
        if (DIGEST && best != NEG_INF) {
            const unsigned long long idx = ((unsigned long long)r2 * d.k2 + i2) * d.k2 + j2;   // oracle's r-major index
            dsum += (unsigned long long)(uint32_t)(best + 1) * (idx + 1);
        }
    }
    if (DIGEST && dsum) atomicAdd(&digest[lvl], dsum);
}

__global__ void dp_init_kernel(int32_t *cur, int RP) {   // level 0: k = 1, every r starts at 0 (:534-535)
    if ((int)threadIdx.x < RP) cur[threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------------------------
// traceback: walk the back-pointer lattice from the sink cell (r = R, i = j = 0)   (:757-785)
// Emits the weighted-edge lists in REVERSE path order; the host reverses them.
// ---------------------------------------------------------------------------------------------
__global__ void dp_traceback_kernel(const LevelDesc *__restrict__ descs, int L, int RP, int R,
                                    const uint32_t *__restrict__ bp, const int32_t *__restrict__ final_val,
                                    ColourCsr col, int cap, int32_t *__restrict__ edges /* 4*cap */, TraceOut *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    TraceOut o;
    o.value = final_val[R];              // sink level: cell (i=0, j=0, r=R) of a k_sink-wide level
    o.s_het = 0; o.n_p1 = 0; o.n_p2 = 0; o.overflow = 0;
    if (o.value != NEG_INF) {
        int i = 0, j = 0, r = R;
        for (int l = L - 1; l >= 1; --l) {
            const LevelDesc d = descs[l];
            const int64_t t = ((int64_t)i * d.k2 + j) * RP + r;
            const uint32_t b = bp[d.bp_off + t];
            const int pi = (int)(b & 0x7FFFu), pj = (int)((b >> 15) & 0x7FFFu);
            const int wu = (int)((b >> 30) & 1u), wv = (int)(b >> 31);
            const int u1 = d.a0 + pi, v1 = d.a0 + pj, u2 = d.b0 + i, v2 = d.b0 + j;
            if (d.delta_off >= 0) o.s_het += score_symd(col, u1, v1, u2, v2);       // :662
            const int reps = (l == L - 1) ? 1 : 0;   // final edges are appended unconditionally (:684-692)
            for (int q = 0; q < reps + wu; ++q) {    // reverse order: the unconditional one first, then the weighted one
                if (o.n_p1 < cap) { edges[o.n_p1] = u1; edges[cap + o.n_p1] = u2; } else o.overflow = 1;
                ++o.n_p1;
            }
            for (int q = 0; q < reps + wv; ++q) {
                if (o.n_p2 < cap) { edges[2 * cap + o.n_p2] = v1; edges[3 * cap + o.n_p2] = v2; } else o.overflow = 1;
                ++o.n_p2;
            }
            i = pi; j = pj; r -= wu + wv;
        }
    }
    *out = o;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t s) {
    if (int rc = b.ensure(bytes)) return rc;
    if (bytes) DG_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
    return DG_OK;
}

static int dp_load(dg_ctx *c, const dg_dp_graph *g) {
    if (!g || !g->level_off || !g->out_off || !g->out_dst || !g->out_w || !g->hom_off || !g->het_off) {
        set_error("dg_dp_load_graph: null array"); return DG_ERR_ARG;
    }
    const int nV = g->n_vertices, L = g->n_levels, R = g->R;
    if (nV < 2 || L < 2 || R < 0 || R > 4096) { set_error("dg_dp_load_graph: bad sizes (V=%d L=%d R=%d)", nV, L, R); return DG_ERR_ARG; }
    if (g->level_off[0] != 0 || g->level_off[L] != nV) { set_error("level_off must span [0, n_vertices]"); return DG_ERR_ARG; }
    if (g->level_off[1] != 1) { set_error("level 0 must hold exactly the source vertex"); return DG_ERR_ARG; }
    if (!c->dp) c->dp = new DpState();
    DpState &S = *c->dp;
    S.loaded = false;
    S.nV = nV; S.L = L; S.R = R; S.RP = R + 1;
    std::vector<int32_t> level_of(nV);
    int max_k = 0;
    for (int l = 0; l < L; ++l) {
        const int k = g->level_off[l + 1] - g->level_off[l];
        if (k <= 0) { set_error("level %d is empty", l); return DG_ERR_ARG; }
        max_k = std::max(max_k, k);
        for (int v = g->level_off[l]; v < g->level_off[l + 1]; ++v) level_of[v] = l;
    }
    if (max_k > MAX_K) { set_error("level width %d exceeds the supported %d", max_k, MAX_K); return DG_ERR_UNSUPPORTED; }
    const int64_t E = g->out_off[nV];
    if (E >= (int64_t)1 << 31) { set_error("too many edges (%lld)", (long long)E); return DG_ERR_UNSUPPORTED; }
    // in-CSR: in-edges of every vertex sorted by (source position asc, adjacency order asc)
    std::vector<uint32_t> in_off((size_t)nV + 1, 0), in_edge((size_t)E);
    std::vector<int32_t> in_dst((size_t)E);
    for (int u = 0; u < nV; ++u) {
        if (g->out_off[u + 1] < g->out_off[u]) { set_error("out_off not monotone at %d", u); return DG_ERR_ARG; }
        for (int64_t e = g->out_off[u]; e < g->out_off[u + 1]; ++e) {
            const int v = g->out_dst[e];
            if (v < 0 || v >= nV || level_of[v] != level_of[u] + 1) {
                set_error("edge %d->%d does not go to the next level", u, v); return DG_ERR_ARG;
            }
            if (g->out_w[e] > 1) { set_error("edge weight %d > 1", (int)g->out_w[e]); return DG_ERR_ARG; }
            in_off[v + 1]++;
        }
    }
    for (int v = 0; v < nV; ++v) in_off[v + 1] += in_off[v];
    {
        std::vector<uint32_t> fill(in_off.begin(), in_off.end() - 1);
        for (int u = 0; u < nV; ++u) {
            const uint32_t pos = (uint32_t)(u - g->level_off[level_of[u]]);
            for (int64_t e = g->out_off[u]; e < g->out_off[u + 1]; ++e) {
                const int v = g->out_dst[e];
                const uint32_t slot = fill[v]++;
                in_edge[slot] = pos | ((uint32_t)g->out_w[e] << 31);
                in_dst[slot] = v;
            }
        }
    }
    for (int v = 0; v < nV; ++v)      // parallel edges must carry equal weights (always true for graphs built by
        for (uint32_t e = in_off[v] + 1; e < in_off[v + 1]; ++e)   // Approximator::solve)
            if ((in_edge[e] & 0x7FFFFFFFu) == (in_edge[e - 1] & 0x7FFFFFFFu) && in_edge[e] != in_edge[e - 1]) {
                set_error("parallel edges with different weights into vertex %d: tie order would be schedule dependent", v);
                return DG_ERR_UNSUPPORTED;
            }
    // colour lists must be sorted (the merges rely on it) and fit the uint16 delta
    int64_t max_list = 0;
    for (int v = 0; v < nV; ++v) {
        for (int pass = 0; pass < 2; ++pass) {
            const int64_t *off = pass ? g->het_off : g->hom_off;
            const int32_t *colv = pass ? g->het_col : g->hom_col;
            if (off[v + 1] < off[v]) { set_error("colour offsets not monotone at %d", v); return DG_ERR_ARG; }
            max_list = std::max(max_list, off[v + 1] - off[v]);
            for (int64_t q = off[v] + 1; q < off[v + 1]; ++q)
                if (colv[q] <= colv[q - 1]) { set_error("colour list of vertex %d is not sorted-unique", v); return DG_ERR_ARG; }
        }
    }
    if (max_list * 4 > 65535) { set_error("colour lists too long for uint16 score deltas (%lld)", (long long)max_list); return DG_ERR_UNSUPPORTED; }

    // level descriptors
    S.descs.assign(L, LevelDesc{});
    std::vector<uint8_t> has_col(L, 0);
    std::vector<int64_t> col_sum(L, 0);   // colour entries per vertex summed per level (for counters)
    for (int v = 0; v < nV; ++v) {
        const int64_t n = (g->hom_off[v + 1] - g->hom_off[v]) + (g->het_off[v + 1] - g->het_off[v]);
        if (n) has_col[level_of[v]] = 1;
    }
    S.cells = S.relaxations = S.edge_pairs = S.colour_entries = 0;
    S.total_cells = 0; S.max_level_cells = S.RP; S.delta_entries = 0;
    std::vector<int32_t> dtrans;
    std::vector<int64_t> dblk_first;
    int64_t nblk = 0;
    for (int l = 1; l < L; ++l) {
        LevelDesc &d = S.descs[l];
        d.a0 = g->level_off[l - 1]; d.k = g->level_off[l] - d.a0;
        d.b0 = g->level_off[l]; d.k2 = g->level_off[l + 1] - d.b0;
        d.in_base = in_off[d.b0];
        d.T = (int32_t)(in_off[d.b0 + d.k2] - d.in_base);
        const int64_t ncell = (int64_t)d.k2 * d.k2 * S.RP;
        d.bp_off = S.total_cells;
        S.total_cells += ncell;
        S.max_level_cells = std::max(S.max_level_cells, ncell);
        S.cells += (uint64_t)ncell;
        S.edge_pairs += (uint64_t)d.T * (uint64_t)d.T;
        if (has_col[l - 1] || has_col[l]) {
            d.delta_off = S.delta_entries;
            S.delta_entries += (int64_t)d.T * d.T;
            dtrans.push_back(l);
            dblk_first.push_back(nblk);
            nblk += ((int64_t)d.T * d.T + DELTA_PER_BLOCK - 1) / DELTA_PER_BLOCK;
            int64_t f = 0;   // sum over in-edges of |col(src)| + |col(dst)|
            for (uint32_t e = d.in_base; e < d.in_base + (uint32_t)d.T; ++e) {
                const int s = d.a0 + (int)(in_edge[e] & 0x7FFFFFFFu), t = in_dst[e];
                f += (g->hom_off[s + 1] - g->hom_off[s]) + (g->het_off[s + 1] - g->het_off[s]) +
                     (g->hom_off[t + 1] - g->hom_off[t]) + (g->het_off[t + 1] - g->het_off[t]);
            }
            S.colour_entries += (uint64_t)(2 * (int64_t)d.T * f);
        } else {
            d.delta_off = -1;
        }
    }
    S.relaxations = S.edge_pairs * (uint64_t)S.RP;
    S.n_delta_blocks = nblk;
    if (nblk >= (int64_t)1 << 31) { set_error("delta grid too large"); return DG_ERR_UNSUPPORTED; }

    // memory budget
    const size_t bp_bytes = (size_t)S.total_cells * 4, st_bytes = (size_t)S.max_level_cells * 4 * 2,
                 dl_bytes = (size_t)S.delta_entries * 2;
    size_t free_b = 0, total_b = 0;
    DG_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t have = free_b + S.d_bp.bytes + S.d_delta.bytes + S.d_val[0].bytes + S.d_val[1].bytes;
    if (bp_bytes + st_bytes + dl_bytes + ((size_t)1 << 30) > have) {
        set_error("back-pointer lattice needs %.1f GB (+%.1f GB state/delta) but only %.1f GB of HBM is free; "
                  "segmented (checkpoint + recompute) mode is not implemented yet",
                  bp_bytes / 1e9, (st_bytes + dl_bytes) / 1e9, have / 1e9);
        return DG_ERR_OOM;
    }
    hipStream_t s = c->stream;
    if (int rc = upload(S.d_descs, S.descs.data(), sizeof(LevelDesc) * L, s)) return rc;
    if (int rc = upload(S.d_in_off, in_off.data(), 4 * in_off.size(), s)) return rc;
    if (int rc = upload(S.d_in_edge, in_edge.data(), 4 * in_edge.size(), s)) return rc;
    if (int rc = upload(S.d_in_dst, in_dst.data(), 4 * in_dst.size(), s)) return rc;
    if (int rc = upload(S.d_hom_off, g->hom_off, 8 * ((size_t)nV + 1), s)) return rc;
    if (int rc = upload(S.d_het_off, g->het_off, 8 * ((size_t)nV + 1), s)) return rc;
    if (int rc = upload(S.d_hom_col, g->hom_col, 4 * (size_t)g->hom_off[nV], s)) return rc;
    if (int rc = upload(S.d_het_col, g->het_col, 4 * (size_t)g->het_off[nV], s)) return rc;
    if (int rc = upload(S.d_dtrans, dtrans.data(), 4 * dtrans.size(), s)) return rc;
    if (int rc = upload(S.d_dblk_first, dblk_first.data(), 8 * dblk_first.size(), s)) return rc;
    if (int rc = S.d_delta.ensure(dl_bytes)) return rc;
    if (int rc = S.d_bp.ensure(bp_bytes)) return rc;
    if (int rc = S.d_val[0].ensure(st_bytes / 2)) return rc;
    if (int rc = S.d_val[1].ensure(st_bytes / 2)) return rc;
    if (int rc = S.d_digest.ensure(8 * (size_t)L)) return rc;
    if (int rc = S.d_trace.ensure(sizeof(TraceOut))) return rc;
    S.cap = R + 8;
    if (int rc = S.d_edges.ensure(4 * 4 * (size_t)S.cap)) return rc;
    DG_HIP(hipStreamSynchronize(s));      // host staging vectors die here
    for (auto &e : S.ev) if (!e) DG_HIP(hipEventCreate(&e));
    memset(&S.timing, 0, sizeof S.timing);
    S.timing.edge_pairs = S.edge_pairs;
    S.timing.colour_entries = S.colour_entries;
    S.timing.state_bytes = st_bytes; S.timing.bp_bytes = bp_bytes; S.timing.delta_bytes = dl_bytes;
    (void)dtrans;
    S.loaded = true;
    return DG_OK;
}

static int dp_run(dg_ctx *c, dg_dp_result *res) {
    DpState *Sp = c->dp;
    if (!Sp || !Sp->loaded) { set_error("dg_dp_run: no graph loaded"); return DG_ERR_STATE; }
    if (!res) { set_error("dg_dp_run: null result"); return DG_ERR_ARG; }
    DpState &S = *Sp;
    hipStream_t s = c->stream;
    const LevelDesc *descs = S.d_descs.as<LevelDesc>();
    ColourCsr col{S.d_hom_off.as<int64_t>(), S.d_het_off.as<int64_t>(), S.d_hom_col.as<int32_t>(), S.d_het_col.as<int32_t>()};
    const int n_dtrans = (int)(S.d_dtrans.bytes && S.n_delta_blocks ? 0 : 0);
    (void)n_dtrans;
    DG_HIP(hipEventRecord(S.ev[0], s));
    int ndt = 0;
    for (int l = 1; l < S.L; ++l) if (S.descs[l].delta_off >= 0) ++ndt;
    if (S.n_delta_blocks > 0)
        hipLaunchKernelGGL(dp_delta_kernel, dim3((unsigned)S.n_delta_blocks), dim3(256), 0, s, descs, S.d_dtrans.as<int32_t>(),
                           S.d_dblk_first.as<int64_t>(), ndt, S.d_in_edge.as<uint32_t>(), S.d_in_dst.as<int32_t>(), col,
                           S.d_delta.as<uint16_t>());
    DG_HIP(hipEventRecord(S.ev[1], s));
    if (S.want_digest) DG_HIP(hipMemsetAsync(S.d_digest.p, 0, 8 * (size_t)S.L, s));
    hipLaunchKernelGGL(dp_init_kernel, dim3(1), dim3(256 * ((S.RP + 255) / 256)), 0, s, S.d_val[0].as<int32_t>(), S.RP);
    for (int l = 1; l < S.L; ++l) {
        const LevelDesc &d = S.descs[l];
        const int64_t ncell = (int64_t)d.k2 * d.k2 * S.RP;
        const unsigned grid = (unsigned)std::min<int64_t>((ncell + 255) / 256, S.max_blocks);
        const int32_t *cur = S.d_val[(l - 1) & 1].as<int32_t>();
        int32_t *nxt = S.d_val[l & 1].as<int32_t>();
        if (S.want_digest)
            hipLaunchKernelGGL(dp_level_kernel<true>, dim3(grid), dim3(256), 0, s, descs, l, S.RP, S.d_in_off.as<uint32_t>(),
                               S.d_in_edge.as<uint32_t>(), S.d_delta.as<uint16_t>(), cur, nxt, S.d_bp.as<uint32_t>(),
                               S.d_digest.as<unsigned long long>());
        else
            hipLaunchKernelGGL(dp_level_kernel<false>, dim3(grid), dim3(256), 0, s, descs, l, S.RP, S.d_in_off.as<uint32_t>(),
                               S.d_in_edge.as<uint32_t>(), S.d_delta.as<uint16_t>(), cur, nxt, S.d_bp.as<uint32_t>(),
                               (unsigned long long *)nullptr);
    }
    DG_HIP(hipEventRecord(S.ev[2], s));
    hipLaunchKernelGGL(dp_traceback_kernel, dim3(1), dim3(64), 0, s, descs, S.L, S.RP, S.R, S.d_bp.as<uint32_t>(),
                       S.d_val[(S.L - 1) & 1].as<int32_t>(), col, S.cap, S.d_edges.as<int32_t>(), S.d_trace.as<TraceOut>());
    DG_HIP(hipEventRecord(S.ev[3], s));
    DG_HIP(hipGetLastError());
    TraceOut to;
    std::vector<int32_t> edges(4 * (size_t)S.cap);
    DG_HIP(hipMemcpyAsync(&to, S.d_trace.p, sizeof to, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(edges.data(), S.d_edges.p, 4 * edges.size(), hipMemcpyDeviceToHost, s));
    if (S.want_digest) {
        S.digest_host.assign(S.L, 0);
        DG_HIP(hipMemcpyAsync(S.digest_host.data(), S.d_digest.p, 8 * (size_t)S.L, hipMemcpyDeviceToHost, s));
    }
    DG_HIP(hipStreamSynchronize(s));
    DG_HIP(hipEventElapsedTime(&S.timing.delta_ms, S.ev[0], S.ev[1]));
    DG_HIP(hipEventElapsedTime(&S.timing.forward_ms, S.ev[1], S.ev[2]));
    DG_HIP(hipEventElapsedTime(&S.timing.traceback_ms, S.ev[2], S.ev[3]));
    DG_HIP(hipEventElapsedTime(&S.timing.total_ms, S.ev[0], S.ev[3]));
    S.timing.n_forward_launches = S.L - 1;
    if (to.overflow || to.n_p1 > S.cap || to.n_p2 > S.cap) { set_error("traceback edge list overflow (%d, %d > %d)", to.n_p1, to.n_p2, S.cap); return DG_ERR_STATE; }
    res->value = to.value; res->s_het = to.s_het; res->n_p1 = to.n_p1; res->n_p2 = to.n_p2;
    res->cells = S.cells; res->relaxations = S.relaxations;
    for (int q = 0; q < to.n_p1; ++q) {       // device order is sink -> source
        const int src = to.n_p1 - 1 - q;
        if (q < res->cap && res->p1_from && res->p1_to) { res->p1_from[q] = edges[src]; res->p1_to[q] = edges[S.cap + src]; }
    }
    for (int q = 0; q < to.n_p2; ++q) {
        const int src = to.n_p2 - 1 - q;
        if (q < res->cap && res->p2_from && res->p2_to) { res->p2_from[q] = edges[2 * S.cap + src]; res->p2_to[q] = edges[3 * S.cap + src]; }
    }
    return DG_OK;
}

}  // namespace dgi

extern "C" int dg_dp_load_graph(dg_ctx *c, const dg_dp_graph *g) {
    if (int rc = dgi::bind(c)) return rc;
    return dgi::dp_load(c, g);
}
extern "C" int dg_dp_run(dg_ctx *c, dg_dp_result *r) {
    if (int rc = dgi::bind(c)) return rc;
    return dgi::dp_run(c, r);
}
extern "C" int dg_dp_solve_diploid(dg_ctx *c, const dg_dp_graph *g, dg_dp_result *r) {
    if (int rc = dg_dp_load_graph(c, g)) return rc;
    return dg_dp_run(c, r);
}
extern "C" int dg_dp_get_timing(dg_ctx *c, dg_dp_timing *t) {
    if (!c || !c->dp || !t) { dgi::set_error("dg_dp_get_timing: no state"); return DG_ERR_STATE; }
    *t = c->dp->timing;
    return DG_OK;
}
extern "C" int dg_dp_get_level_digest(dg_ctx *c, uint64_t *out, int64_t n) {
    if (!c || !c->dp || !out) { dgi::set_error("dg_dp_get_level_digest: no state"); return DG_ERR_STATE; }
    if ((int64_t)c->dp->digest_host.size() != n) { dgi::set_error("digest not collected (set option digest=1) or size mismatch"); return DG_ERR_STATE; }
    memcpy(out, c->dp->digest_host.data(), 8 * (size_t)n);
    return DG_OK;
}
extern "C" int dg_dp_set_option(dg_ctx *c, const char *key, int64_t v) {
    if (!c || !key) { dgi::set_error("dg_dp_set_option: null"); return DG_ERR_ARG; }
    if (!c->dp) c->dp = new dgi::DpState();
    if (!strcmp(key, "digest")) c->dp->want_digest = v;
    else if (!strcmp(key, "graph")) c->dp->use_graph = v;
    else if (!strcmp(key, "max_blocks")) c->dp->max_blocks = v > 0 ? v : 2048;
    else { dgi::set_error("unknown option %s", key); return DG_ERR_ARG; }
    return DG_OK;
}
