// libdipgenie_hip.so -- diploid pair-of-paths DP for MI355X (gfx950).
//
// Replaces the level loop + sink read-out of Approximator::diploid_dp_approximation_solver
// (/root/reference/src/approximator.cpp:532-716, 757-785).  Design:
//   * gather form: every destination cell (i2, j2, r2) of level l+1 reduces over in(u2) x in(v2) -- one wave per
//     (destination row, group of <= 64 column in-edges, chunk of recombination counts), lanes on the column
//     in-edges; in-edges are stored sorted by source position, so a lexicographic max over (value, in-edge ranks)
//     reproduces the reference's take-if total order (value desc, pred_i asc, pred_j asc, :657-659) without locks
//     or atomics, and no destination is ever "reset" (:565-576 disappears);
//   * rolling value state is 4 B/cell, layout [i][r][j] (j fastest: lanes of a wave run along the destination
//     columns and read (near-)consecutive addresses); s_het / edge chains are not carried (reference cell = 40 B).
//     Instead every cell streams one 2-byte back-pointer (ranks of the winning in-edges) to HBM and a traceback
//     walks the lattice from the sink, emitting the weighted-edge lists (:757-764, :673-692) and re-deriving
//     s_het from the colour lists of the L winning edge pairs;
//   * score deltas (:604-624) do not depend on r nor on other levels: a launch fills, for every transition that
//     touches a colour, the T x T matrix delta[e_u][e_v] (uint16), T = #in-edges of the destination level;
//     colourless transitions (73 % on MHC_4) skip the lookup; per in-edge colour flags and "self" scores answer every
//     pair with at most one coloured edge without a merge, the rest are queued in LDS and merged densely;
//   * one launch per level (the levels are a dependency chain); the host picks the chunk size RC per level from a
//     cost model and gives rows with many in-edges cooperative workgroups; every 128 levels a look-ahead launch
//     streams the next batch's graph tables through the Infinity Cache; on narrow graphs the launches of 1,000
//     levels are captured into a hipGraph and replayed (the chain is host-bound there).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

#include "dg_internal.hpp"

namespace dgi {

constexpr int32_t NEG_INF = INT32_MIN / 4;              // approximator.cpp:413
constexpr uint32_t BP_NONE = 0xFFFFFFFFu;              // wide (32-bit) back-pointer of an unreachable cell
constexpr int MAX_K = 1 << 15;                          // hop words pack positions in 15 bits
// Back-pointers are 16 bits per cell: (eu << 8) | ev, the ranks of the winning in-edges inside the destination row's
// and column's in-edge lists (sorted by source position, so rank order IS the reference's (pred_i asc, pred_j asc)
// tie order); 0xFFFF = unreachable.  Only a level with an in-degree > 255 keeps the wide word
// pred_i | pred_j << 15 | wu << 30 | wv << 31 (two 16-bit units per cell) and runs on the generic kernel.
constexpr int BP_MAX_RANK = 255;
#ifndef DG_HEAVY_U1
#define DG_HEAVY_U1 8                                    // in-edges per step of a heavy row at RC = 1 / RC = 2
#endif
#ifndef DG_HEAVY_U2
#define DG_HEAVY_U2 8
#endif

struct LevelDesc {                                      // transition (l-1) -> l, indexed by l
    int32_t a0, k;                                      // source level: first vertex id, width
    int32_t b0, k2;                                     // destination level
    uint32_t in_base;                                   // first in-edge of the destination level
    int32_t T;                                          // in-edges into the destination level
    int64_t delta_off;                                  // offset of the T*T uint16 matrix, -1 if all zero
    int64_t bp_off;                                     // offset of this level in the bp lattice, in 16-bit units (even)
    int32_t grp_first, ngroups;                         // column groups (runs of <=64 in-edges covering whole columns)
    int32_t dead_first, ndead;                          // destination columns with no in-edge
    int64_t slot_first;                                 // first entry of this level in the 64-wide slot table
    int32_t fast_ok, nblocks;                           // fast kernel usable; number of 64-slot blocks (>= ngroups: giant columns take several)
    int32_t heavy_first, n_heavy;                       // rows with more than COOP_MIN in-edges: slice of the heavy-row table
    int32_t bp_wide, bp_nt;                             // wide back-pointers on this level; stream them with non-temporal stores
};

struct TraceOut {
    int32_t value, s_het, n_e, overflow;
};

struct DpState {
    int32_t nV = 0, L = 0, R = 0, RP = 0, cap = 0;
    bool loaded = false;
    int64_t want_digest = 0, use_fast = 1, use_team = 0, team_grid = 256, max_blocks = 1024, team_fallbacks = 0;
    int last_team_size = 0;
    bool all_fast = false;
    int64_t team_max_tasks = 100, team_min_levels = 16, adaptive_rc = 3, chip_waves = 8192, waves_per_block = 4;
    // lattice segments: destination levels [seg_begin[s], seg_begin[s+1]); one segment = whole lattice resident.
    // More than one = checkpoint + recompute (value-only pass, then each segment re-swept with back-pointers, last first).
    std::vector<int> seg_begin;
    std::vector<int64_t> ckpt_off;                     // element offset of checkpoint s (state of level seg_begin[s]-1)
    int64_t segment_cells = 0;                          // option: force segments of at most this many cells (tests)
    int64_t host_threads = 16;                          // option: threads used by dg_dp_load_graph's table construction
    int64_t bp_nt_min_cells = 16384;                    // option: levels with at least this many cells stream their back-pointers non-temporally
    int64_t chain_spec = 1;                             // option: speculative chain walk (0: the plain one)
    int64_t coop_cost_ns = 0;                           // model: barrier + LDS merge of a cooperative task
    int64_t use_coop = 1;                               // option: cooperative tasks for rows with many in-edges
    int64_t warm_rows = 1;                              // option: warm the row records before every chain walk
    bool graph_failed = false;                          // capture or instantiation failed once: plain launches from then on
    int64_t graph_batch = -1;                           // option: levels per captured hipGraph (0 = plain launches, -1 = by level width)
    std::map<std::tuple<int, int, const void *>, hipGraphExec_t> graphs;   // (first level, end level, biased lattice pointer) -> replayable batch
    int64_t warm_ahead = 128;                           // option: sweep look-ahead, levels per batch (0 = off)
    int64_t sync_every = 0;                             // option: drain the stream every N level launches (profiler aid)
    struct Segment { int begin, end; bool team; };
    std::vector<Segment> schedule;
    size_t state_alloc_bytes = 0;
    std::vector<LevelDesc> descs;
    uint64_t cells = 0, relaxations = 0, edge_pairs = 0, colour_entries = 0;
    int64_t total_units = 0, max_level_units = 0;       // back-pointer lattice, in 16-bit units (1 per cell, 2 on wide levels)
    int64_t max_level_cells = 0, delta_entries = 0, n_delta_blocks = 0, pad_front = 0;
    std::vector<int64_t> level_units;                   // units of every level (even)
    // score-delta windows: coloured transitions [dwin_t[w], dwin_t[w+1]) are resident together (one window = everything
    // unless the matrices outgrow delta_cap_entries; then each window is recomputed right before its first level)
    std::vector<int32_t> dwin_t, level_win;             // level_win[l] = window of level l's transition, -1 if colourless
    int cur_win = -1;                                   // window whose matrices are in d_delta right now
    std::vector<int32_t> dtrans_host;
    std::vector<int64_t> dblk_first_host;
    int64_t delta_cap_entries = (int64_t)4 << 30, delta_buf_entries = 0;
    std::vector<int32_t> level_dmax;                    // largest in-degree among the level's vertices
    int64_t rc_cap = 65536, rc_t0_ns = 3000, rc_tg_ps = 24000, rc_tw_ps = 50;   // cost model of the per-level RC choice
    DevBuf d_descs, d_in_off, d_in_edge, d_in_dst, d_hom_off, d_het_off, d_hom_col, d_het_col, d_eflag, d_eself;
    DevBuf d_delta, d_bp, d_val[2], d_digest, d_trace, d_edges, d_dblk_first, d_dtrans, d_ctrl, d_grp, d_dead, d_heavy, d_rowrec, d_slots, d_path, d_ckpt, d_chain;
    std::vector<uint64_t> digest_host;
    dg_dp_timing timing;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // The resident back-pointer lattice lives in a pool of equal chunks that a background thread allocates one by
    // one (a 100+ GB hipMalloc takes seconds, more when another process has just freed HBM): reservation can start
    // before the graph exists (dg_dp_prealloc) without starving the small allocations of the sketch stage, and the
    // sweep starts on chunk 0 while later chunks are still being mapped.  Levels never straddle chunks.
    struct Pool {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<void *> chunks;           // each chunk_units * 2 bytes
        size_t chunk_units = (size_t)4 << 30;  // 8 GB
        size_t target = 0;                     // chunks wanted
        size_t cap_chunks = 0;                 // upper bound for reservations made before the graph is known
        bool running = false, failed = false, paused = false;
        std::thread th;
    } pool;
    std::vector<int> chunk_begin;              // destination levels [chunk_begin[c], chunk_begin[c+1]) live in chunk c
    size_t chunk_units_cfg = (size_t)4 << 30;  // configured chunk size (option lattice_chunk_cells); pool.chunk_units = the live pool's
    int seg_chunks = 1;                        // chunks per lattice segment (= all of them when the lattice is resident)
};

static void graphs_clear(DpState &S) {                 // captured level batches: stale as soon as the graph, the lattice or an option changes
    for (auto &kv : S.graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    S.graphs.clear();
}

void dp_state_free(DpState *s) {
    if (!s) return;
    graphs_clear(*s);
    { std::unique_lock<std::mutex> lk(s->pool.mu); s->pool.target = 0; }
    if (s->pool.th.joinable()) s->pool.th.join();
    for (void *q : s->pool.chunks) (void)hipFree(q);
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
constexpr int DELTA_PER_BLOCK = 256 * 16;
constexpr int TEAM_CTL_SLOTS = 64;                      // control blocks recycled by successive team launches
constexpr int DELTA_PAD = 8;                            // delta[0..8) stays zero: the colourless transitions' slot

// Per in-edge (source -> destination), once per graph: which of the four colour lists that an edge pair can contribute are
// non-empty -- bit 0 Hom(source), bit 1 Hom(destination), bit 2 Het(source), bit 3 Het(destination).  Most edges of a
// coloured transition carry no colour at all; with the flags the delta kernel answers those pairs from two byte loads
// instead of sixteen offset loads, and pairs with ONE coloured edge from that edge's own score (self[]).
__global__ __launch_bounds__(64) void dp_edge_flags_kernel(const LevelDesc *__restrict__ descs, int L, const uint32_t *__restrict__ in_edge,
                                                           const int32_t *__restrict__ in_dst, ColourCsr col, uint8_t *__restrict__ flags,
                                                           uint16_t *__restrict__ self) {
    const int l = (int)blockIdx.x + 1;
    if (l >= L) return;
    const LevelDesc d = descs[l];
    for (int e = (int)threadIdx.x; e < d.T; e += 64) {
        const int u1 = d.a0 + (int)(in_edge[d.in_base + e] & 0x7FFFFFFFu), u2 = in_dst[d.in_base + e];
        const uint32_t f = (col.hom_off[u1 + 1] > col.hom_off[u1] ? 1u : 0u) | (col.hom_off[u2 + 1] > col.hom_off[u2] ? 2u : 0u) |
                           (col.het_off[u1 + 1] > col.het_off[u1] ? 4u : 0u) | (col.het_off[u2 + 1] > col.het_off[u2] ? 8u : 0u);
        flags[d.in_base + e] = (uint8_t)f;
        // score of this edge paired with an edge that carries no colour: |Hom(u1) n Hom(u2)| + |Het(u1) /\ Het(u2)|
        int sc = 0;
        if ((f & 3u) == 3u) {
            const int64_t a = col.hom_off[u1], b = col.hom_off[u2];
            sc += union2x2<false>(col.hom_col + a, (int)(col.hom_off[u1 + 1] - a), nullptr, 0, col.hom_col + b, (int)(col.hom_off[u2 + 1] - b), nullptr, 0);
        }
        if (f & 12u) {
            const int64_t a = col.het_off[u1], b = col.het_off[u2];
            sc += union2x2<true>(col.het_col + a, (int)(col.het_off[u1 + 1] - a), nullptr, 0, col.het_col + b, (int)(col.het_off[u2 + 1] - b), nullptr, 0);
        }
        self[d.in_base + e] = (uint16_t)sc;
    }
}

__global__ __launch_bounds__(256) void dp_delta_kernel(const LevelDesc *__restrict__ descs,
                                                       const int32_t *__restrict__ dtrans,      // coloured transition -> level
                                                       const int64_t *__restrict__ dblk_first,  // first block of each coloured transition
                                                       int n_dtrans, const uint32_t *__restrict__ in_edge,
                                                       const int32_t *__restrict__ in_dst, ColourCsr col,
                                                       uint16_t *__restrict__ delta /* biased like SweepArgs::delta */, int64_t block0,
                                                       const uint8_t *__restrict__ eflags, const uint16_t *__restrict__ eself) {
    __shared__ int s_t;
    if (threadIdx.x == 0) {   // binary search: last transition whose first block <= blockIdx.x
        int lo = 0, hi = n_dtrans - 1;
        const int64_t b = block0 + blockIdx.x;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (dblk_first[mid] <= b) lo = mid; else hi = mid - 1; }
        s_t = lo;
    }
    __syncthreads();
    const int t = s_t;
    const LevelDesc d = descs[dtrans[t]];
    const int64_t n = (int64_t)d.T * d.T;
    const int64_t first = (block0 + (int64_t)blockIdx.x - dblk_first[t]) * DELTA_PER_BLOCK;
    uint16_t *out = delta + d.delta_off;
    // Pass 1, eight entries per thread at a time (independent index arithmetic and flag loads go out back to back): pairs whose
    // four colour lists cannot contribute -- the vast majority -- are answered with 0 on the spot, the others are queued in
    // LDS.  Pass 2 works the queue off densely: the sorted-list merges are the expensive part (16 dependent loads and
    // data-dependent loops), and with one coloured in-edge in ten nearly every wave used to hold a few of them, so every
    // wave paid for them (34 ms on MHC-24; 2.6 ms with the merges switched off).
    __shared__ uint16_t s_q[DELTA_PER_BLOCK];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const bool small = n < ((int64_t)1 << 31);
    for (int q0 = 0; q0 < DELTA_PER_BLOCK / 256; q0 += 8) {
        int64_t ee[8];
        uint32_t f[8];                                                    // bit 8: entry exists; bits 0-7 / 16-23: flags of e_u / e_v
        int eu8[8], ev8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t e = first + (int64_t)(q0 + u) * 256 + threadIdx.x;
            const bool ok = e < n;
            const int64_t ec = ok ? e : 0;
            const int eu = small ? (int)((uint32_t)ec / (uint32_t)d.T) : (int)(ec / d.T);
            const int ev = (int)(ec - (int64_t)eu * d.T);
            ee[u] = e; eu8[u] = eu; ev8[u] = ev;
            f[u] = ok ? ((uint32_t)eflags[d.in_base + eu] | ((uint32_t)eflags[d.in_base + ev] << 16) | 0x100u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!(f[u] & 0x100u)) continue;
            const uint32_t fu = f[u] & 0xFFu, fv = f[u] >> 16;
            if ((fu | fv) == 0u) out[ee[u]] = 0;                            // no colour on either edge
            else if (fv == 0u) out[ee[u]] = eself[d.in_base + eu8[u]];      // one coloured edge: its own score
            else if (fu == 0u) out[ee[u]] = eself[d.in_base + ev8[u]];
            else s_q[atomicAdd(&s_n, 1)] = (uint16_t)((q0 + u) * 256 + (int)threadIdx.x);   // both coloured: merge (pass 2)
        }
    }
    __syncthreads();
    const int nq = s_n;
    for (int i = (int)threadIdx.x; i < nq; i += 256) {
        const int64_t e = first + (int64_t)s_q[i];
        const int eu = small ? (int)((uint32_t)e / (uint32_t)d.T) : (int)(e / d.T);
        const int ev = (int)(e - (int64_t)eu * d.T);
        const uint32_t ff = (uint32_t)eflags[d.in_base + eu] | (uint32_t)eflags[d.in_base + ev];
        const uint32_t pu = in_edge[d.in_base + eu], pv = in_edge[d.in_base + ev];
        const int u1 = d.a0 + (int)(pu & 0x7FFFFFFFu), v1 = d.a0 + (int)(pv & 0x7FFFFFFFu);
        const int u2 = in_dst[d.in_base + eu], v2 = in_dst[d.in_base + ev];
        int sc = 0;
        if ((ff & 3u) == 3u) sc += score_inter(col, u1, v1, u2, v2);
        if ((ff & 12u) != 0u) sc += score_symd(col, u1, v1, u2, v2);
        out[e] = (uint16_t)sc;
    }
}

// ---------------------------------------------------------------------------------------------
// level sweep, edge-pair form (the production kernel).  State layout [i][r][j] (j fastest).
//
// The work of one transition is the T x T grid of in-edge pairs (e_u, e_v) -- exactly the index
// space of the delta matrix.  One wave owns one task = (destination row i2, column group g): a group
// is a run of <= 64 consecutive in-edges e_v that covers whole destination columns (host-built), so
// lane <-> e_v and every destination cell's candidates sit in adjacent lanes.  The wave walks the
// row's in-edges e_u (wave-uniform), and per step every lane does one delta load (coalesced along
// e_v) and RC value loads (coalesced along the source column j) -- no per-lane inner loop, so a
// vertex with in-degree 24 costs 24 steps instead of 24 x 24.  Each lane keeps, per recombination
// count, the best candidate as the pair (value, ord) with ord = (~i, ~j, wu, wv) packed so that a
// plain lexicographic max IS the reference's take-if order (value desc, pred_i asc, pred_j asc,
// approximator.cpp:657-659).  A log-step segmented max over lanes of equal destination column
// finishes the cell; the segment head stores the value and the back-pointer.
// ---------------------------------------------------------------------------------------------
struct SweepArgs {
    const LevelDesc *descs;
    const uint32_t *in_off, *in_edge, *grp_begin;
    const int32_t *in_dst, *dead_cols;
    const uint16_t *delta, *delta_zero;                 // delta: biased so that delta[d.delta_off] is valid for the resident window
    int32_t *buf0, *buf1;
    uint16_t *bp;
    unsigned long long *digest;
    int RP;
};

__device__ __forceinline__ uint32_t ord_word(int i, int j, int wu, int wv) {
    return ((uint32_t)(0x7FFF - i) << 17) | ((uint32_t)(0x7FFF - j) << 2) | ((uint32_t)wu << 1) | (uint32_t)wv;
}
__device__ __forceinline__ uint32_t bp_from_ord(uint32_t o) {
    const uint32_t i = 0x7FFFu - (o >> 17), j = 0x7FFFu - ((o >> 2) & 0x7FFFu);
    return i | (j << 15) | (((o >> 1) & 1u) << 30) | ((o & 1u) << 31);
}
// narrow form: ord = (255 - eu) << 8 | (255 - ev) is never 0 for a real candidate, and the stored back-pointer is
// simply ~ord (an untouched best keeps ord 0 -> 0xFFFF = unreachable)
// non-temporal 16-bit store as inline asm: with the builtin on one side of a branch and a plain store on the other the
// optimiser merges the two into ONE plain store (the !nontemporal hint is dropped)
__device__ __forceinline__ void store_bp_nt(uint16_t *p, uint32_t v) { asm volatile("global_store_short %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t ord_rank(int eu, int ev) { return ((uint32_t)(BP_MAX_RANK - eu) << 8) | (uint32_t)(BP_MAX_RANK - ev); }

template <int RC, bool DIGEST>
__device__ __forceinline__ void sweep_level_pairs(const SweepArgs &A, int lvl, int wave_id, int n_waves) {
    const LevelDesc d = A.descs[lvl];
    const int RP = A.RP;
    const int32_t *__restrict__ cur = ((lvl - 1) & 1) ? A.buf1 : A.buf0;
    int32_t *__restrict__ nxt = (lvl & 1) ? A.buf1 : A.buf0;
    const int lane = threadIdx.x & 63;
    const int nchunk = (RP + RC - 1) / RC;
    const int64_t ntask = (int64_t)d.k2 * d.ngroups * nchunk;
    const bool has_delta = d.delta_off >= 0;
    const uint16_t *dm = has_delta ? A.delta + d.delta_off : A.delta_zero;   // (A.delta is biased by the resident delta window)
    const int dT = has_delta ? d.T : 0, dmask = has_delta ? -1 : 0;
    const uint32_t *gb = A.grp_begin + d.grp_first;
    unsigned long long dsum = 0;
    for (int64_t task = wave_id; task < ntask; task += n_waves) {
        const int g = (int)(task % d.ngroups);
        const int64_t rest = task / d.ngroups;
        const int rc = (int)(rest % nchunk), i2 = (int)(rest / nchunk);
        const int r0 = rc * RC;
        const uint32_t gbeg = gb[g], gend = gb[g + 1];
        const uint32_t eu0 = A.in_off[d.b0 + i2], eu1 = A.in_off[d.b0 + i2 + 1];
        int bval[RC];
        uint32_t bord[RC];
#pragma unroll
        for (int q = 0; q < RC; ++q) { bval[q] = NEG_INF; bord[q] = 0; }
        int j2 = -1 - lane;                                             // inactive lanes: unique negative ids
        // a group wider than 64 is one giant column (host guarantee): lanes accumulate over its chunks
        for (uint32_t cb = gbeg; cb < gend; cb += 64) {
            const uint32_t ev = cb + lane;
            const bool act = ev < gend;
            int j = 0, wv = 0, evr = 0;
            if (act) {
                const uint32_t pv = A.in_edge[ev];
                j = (int)(pv & 0x7FFFFFFFu); wv = (int)(pv >> 31);
                const int cv = A.in_dst[ev];
                j2 = cv - d.b0;
                if (!d.bp_wide) evr = (int)(ev - A.in_off[cv]);            // rank inside the column's in-edge list
            }
            const int dcol = (int)(ev - d.in_base) & dmask;
            for (uint32_t eu = eu0; eu < eu1; ++eu) {
                const uint32_t pu = A.in_edge[eu];
                const int i = (int)(pu & 0x7FFFFFFFu), wu = (int)(pu >> 31);
                if (act) {
                    const int w = wu + wv;
                    const int dl = (int)dm[(int64_t)(eu - d.in_base) * dT + dcol];
                    const uint32_t ord = d.bp_wide ? ord_word(i, j, wu, wv) : ord_rank((int)(eu - eu0), evr);
                    // rows r = r2 - w; the buffers carry front/tail padding so r = -1, -2 (and r2 >= RP in a
                    // ragged last chunk) are legal reads that the select discards: RC loads back to back
                    const int32_t *base = cur + ((int64_t)i * RP + (r0 - w)) * d.k + j;
                    int vals[RC];
#pragma unroll
                    for (int q = 0; q < RC; ++q) vals[q] = base[q * d.k];
#pragma unroll
                    for (int q = 0; q < RC; ++q) {
                        const int cand = vals[q] + dl;
                        const bool ok = (r0 + q < RP) & (r0 + q - w >= 0) & (vals[q] != NEG_INF);       // :633, :646-647
                        const bool take = ok & ((cand > bval[q]) | ((cand == bval[q]) & (ord > bord[q])));   // :657-659
                        bval[q] = take ? cand : bval[q];
                        bord[q] = take ? ord : bord[q];
                    }
                }
            }
        }
        // segmented max over lanes with equal destination column (lanes of a column are adjacent)
        const int span = (int)min(gend - gbeg, 64u);
        for (int s = 1; s < span; s <<= 1) {
            const int oj2 = __shfl_down(j2, s);
            const bool same = (lane + s < 64) & (oj2 == j2);
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const int ov = __shfl_down(bval[q], s);
                const uint32_t oo = (uint32_t)__shfl_down((int)bord[q], s);
                const bool take = same & ((ov > bval[q]) | ((ov == bval[q]) & (oo > bord[q])));
                bval[q] = take ? ov : bval[q];
                bord[q] = take ? oo : bord[q];
            }
        }
        const int pj2 = __shfl_up(j2, 1);
        const bool head = (j2 >= 0) & ((lane == 0) | (pj2 != j2));
        if (head) {
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const int r2 = r0 + q;
                if (r2 < RP) {
                    const int64_t idx = ((int64_t)i2 * RP + r2) * d.k2 + j2;
                    nxt[idx] = bval[q];
                    if (A.bp) {
                        if (d.bp_wide) __builtin_nontemporal_store((uint32_t)(bval[q] == NEG_INF ? BP_NONE : bp_from_ord(bord[q])), (uint32_t *)(A.bp + d.bp_off + 2 * idx));
                        else __builtin_nontemporal_store((uint16_t)~bord[q], &A.bp[d.bp_off + idx]);
                    }
                    if (DIGEST && bval[q] != NEG_INF) {
                        const unsigned long long o = ((unsigned long long)r2 * d.k2 + i2) * d.k2 + j2;   // oracle's r-major index
                        dsum += (unsigned long long)(uint32_t)(bval[q] + 1) * (o + 1);
                    }
                }
            }
        }
        // destination columns without any in-edge are unreachable: nobody owns them, clear them here
        if (g == 0 && d.ndead > 0) {
            for (int t = lane; t < d.ndead * RC; t += 64) {
                const int q = t % RC, c = A.dead_cols[d.dead_first + t / RC];
                if (r0 + q < RP) {
                    const int64_t idx = ((int64_t)i2 * RP + r0 + q) * d.k2 + c;
                    nxt[idx] = NEG_INF;
                    if (A.bp) {
                        if (d.bp_wide) __builtin_nontemporal_store((uint32_t)(BP_NONE), (uint32_t *)(A.bp + d.bp_off + 2 * idx));
                        else __builtin_nontemporal_store((uint16_t)0xFFFFu, &A.bp[d.bp_off + idx]);
                    }
                }
            }
        }
    }
    if (DIGEST && dsum) atomicAdd(&A.digest[lvl], dsum);
}

// ---------------------------------------------------------------------------------------------
// Fast form of the same sweep.  A level's kernel starts with cold caches (kernel boundary), so its
// duration is a chain of dependent memory round trips (~1 us each) -- the fast form cuts the chain to
//   kernarg (LevelDesc by value)  ->  {row record, slot record}  ->  {delta, RC values}  ->  stores
// * slot table: every column group is padded to exactly 64 records {j | wv<<15 | j2<<16, ev_local |
//   steps<<28}, so a lane finds its in-edge without reading group offsets;
// * row record {first in-edge, in-degree, in-edge 0, in-edge 1} serves 97 % of the rows in one load;
//   heavier rows fetch their in-edge list once, one per lane, and broadcast with readlane;
// * the row's in-edges are processed two per step so two sets of RC loads are in flight;
// * the segmented max runs only ceil(log2(max column in-degree of the group)) steps.
// A column with in-degree > 64 (more than 64 haplotypes recombining into one vertex) spans several blocks that one
// wave walks in turn; rows of any in-degree fetch their in-edges 64 at a time.  Only sizes beyond the 2-D grid or
// 2^28 in-edges per level fall back to the generic kernel above.
// ---------------------------------------------------------------------------------------------
struct FastArgs {
    const uint4 *rowrec;
    const uint2 *slots;
    const uint32_t *in_edge;
    const int32_t *dead_cols;
    const uint16_t *delta, *delta_zero;
    int32_t *base0, *base1;                             // padded allocation starts of the two state buffers
    uint16_t *bp;                                       // fast-form levels always store narrow back-pointers
    unsigned long long *digest;
    int RP, pad_bytes;                                  // pad_bytes: front padding of the state buffers
    uint32_t buf_bytes;                                 // size of one padded state buffer
};

// neighbour exchange by one lane as DPP wave shifts (a few cycles) instead of ds_bpermute (an LDS crossbar round trip):
// most column groups need exactly one step of the segmented max (columns with at most two in-edges)
__device__ __forceinline__ int lane_down1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x130 /* wave_shl:1: lane i <- lane i + 1 */, 0xF, 0xF, false); }
__device__ __forceinline__ int lane_up1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1: lane i <- lane i - 1 */, 0xF, 0xF, false); }

template <int RC>
__device__ __forceinline__ void relax_select(const int (&vals)[RC], int dl, uint32_t ord, int r0, int w, int RP,
                                             int (&bval)[RC], uint32_t (&bord)[RC]) {
#pragma unroll
    for (int q = 0; q < RC; ++q) {
        const int cand = vals[q] + dl;
        const bool ok = (r0 + q < RP) & (r0 + q - w >= 0) & (vals[q] != NEG_INF);                   // :633, :646-647
        const bool take = ok & ((cand > bval[q]) | ((cand == bval[q]) & (ord > bord[q])));           // :657-659
        bval[q] = take ? cand : bval[q];
        bord[q] = take ? ord : bord[q];
    }
}

// One task of the fast form: destination row i2, column group g, recombination chunk starting at r0.
// AUX is the cache policy of the state loads: 0 = plain (per-level launches: the kernel boundary makes
// the previous level visible), 16 = sc1 (team kernel: served by the XCD's L2, bypassing the CU's L1).
// Lean variant: every in-degree of the level is <= 64 (no giant column blocks, row in-edges fit one per lane).
// It is a separate function on purpose: a lone wave retires ~1 instruction per 4-8 cycles, and the extra loop
// structure of the general variant below costs 25-30 % on the narrow levels that dominate MHC-scale graphs.
// COOP (per-level launches only): 0 = every row; 1 = rows with more than COOP_MIN in-edges return (they are done by the
// cooperative region of the same launch); 2 = cooperative: the four waves of the workgroup walk a quarter of the row's
// in-edges each, wave 0 merges the partial bests through LDS and finishes the task.  A row with 24 in-edges (one per
// haplotype) is three dependent load rounds for one wave and sets the level's critical path; split, it is one.
constexpr int COOP_MIN = 8;
template <int RC, bool DIGEST, int AUX, bool PROF = false, int COOP = 0>
__device__ __forceinline__ void sweep_task(const FastArgs &A, const LevelDesc &d, __amdgpu_buffer_rsrc_t cur_rsrc,
                                           int32_t *__restrict__ nxt, int i2, int g, int r0, int lvl, unsigned long long *pp = nullptr,
                                           int part = 0, uint2 *ex = nullptr) {
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
    if (PROF) q0 = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x & 63;
    const int RP = A.RP;
    const uint4 rr = A.rowrec[d.b0 + i2];                               // {eu0, du, pu0, pu1}
    const uint2 sl = A.slots[d.slot_first + (int64_t)g * 64 + lane];
    const bool act = sl.x != 0xFFFFFFFFu;
    const int j = (int)(sl.x & 0x7FFFu), wv = (int)((sl.x >> 15) & 1u);
    const int j2 = act ? (int)((sl.x >> 16) & 0x7FFFu) : -1 - lane;
    const int steps = __builtin_amdgcn_readfirstlane((int)(sl.y >> 28));
    const bool has_delta = d.delta_off >= 0;
    const uint16_t *dm = has_delta ? A.delta + d.delta_off : A.delta_zero;   // (A.delta is biased by the resident delta window)
    const int dT = has_delta ? d.T : 0;
    const int dcol = has_delta ? (int)(sl.y & 0x000FFFFFu) : 0;
    const int evr = (int)((sl.y >> 20) & 0xFFu);                       // rank of this lane's in-edge inside its column's list
    const int du = (int)rr.y;
#ifdef DG_SKELETON                                                      // timing probe: geometry + the two record loads + the stores, no task body
    {
        const int pj2s = lane_up1(j2);
        if (act & ((lane == 0) | (pj2s != j2)) & (du >= 0)) {
#pragma unroll
            for (int q = 0; q < RC; ++q) if (r0 + q < RP) {
                const int idx = (i2 * RP + r0 + q) * d.k2 + j2;
                nxt[idx] = NEG_INF;
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + idx], 0xFFFFu); else A.bp[d.bp_off + idx] = (uint16_t)0xFFFFu; }
            }
        }
        return;
    }
#endif
    if (COOP == 1 && du > COOP_MIN) return;
    uint32_t mypu = 0;
    if (du > 2 && lane < du) mypu = A.in_edge[rr.x + lane];            // du <= 64 on this path
    if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); q1 = __builtin_amdgcn_s_memtime(); }
    int bval[RC];
    uint32_t bord[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q) { bval[q] = NEG_INF; bord[q] = 0; }
    const int64_t erow0 = (int64_t)(rr.x - d.in_base) * dT;
    const int rowbytes = d.k * 4;
    // Source rows are r = r2 - w.  Byte offsets are relative to the padded buffer start (the resource), so rows
    // r = -1, -2 land in the front padding and r2 >= RP (ragged last chunk) in the tail padding; the select
    // discards what is out of range, which lets every load be issued unconditionally, back to back.
    if (du <= 2) {
        // 97 % of the rows: both in-edges came with the row record, no loop, no extra load
        if (act && du > 0) {
            const int ia = (int)(rr.z & 0x7FFFu), wa = (int)(rr.z >> 31) + wv;    // (bit 16: flag for the chain walk)
            const int ib = (int)(rr.w & 0x7FFFu), wb = (int)(rr.w >> 31) + wv;
            const int offa = ((ia * RP + (r0 - wa)) * d.k + j) * 4 + A.pad_bytes;
            const int offb = ((ib * RP + (r0 - wb)) * d.k + j) * 4 + A.pad_bytes;
            int va[RC], vb[RC];
            const int dla = (int)dm[erow0 + dcol];
            int dlb = 0;
#pragma unroll
            for (int q = 0; q < RC; ++q) va[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offa + q * rowbytes, 0, AUX);
            if (du == 2) {
                dlb = (int)dm[erow0 + dT + dcol];
#pragma unroll
                for (int q = 0; q < RC; ++q) vb[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offb + q * rowbytes, 0, AUX);
            }
            relax_select<RC>(va, dla, ord_rank(0, evr), r0, wa, RP, bval, bord);
            if (du == 2) relax_select<RC>(vb, dlb, ord_rank(1, evr), r0, wb, RP, bval, bord);
        }
    } else {
        // heavy rows (recombination fan-in): U in-edges per step -- all their loads (U deltas + U*RC values) go out
        // back to back, then the selects run; (value, ord) max is associative and commutative, so the order inside
        // a step is irrelevant.  Small RC leaves registers for a deep step: in-degree 23 takes 2 steps at RC = 1.
        constexpr int U = RC >= 8 ? 2 : (RC >= 4 ? 4 : (RC >= 3 ? 8 : (RC == 2 ? DG_HEAVY_U2 : DG_HEAVY_U1)));
        const int t_lo = COOP == 2 ? (du * part) >> 2 : 0, t_hi = COOP == 2 ? (du * (part + 1)) >> 2 : du;
        for (int t = t_lo; t < t_hi; t += U) {
            // the in-edge words live in SGPRs only until the load offset is formed; the select needs just their weight
            // bits, kept in one mask (a deep step would otherwise hold U scalars and spill)
            uint32_t wmask = 0;
            if (act) {
                int vals[U][RC], dl[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (t + u < t_hi) {                                   // wave-uniform
                        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)mypu, t + u);
                        wmask |= (p >> 31) << u;
                        const int iu = (int)(p & 0x7FFFFFFFu), w = (int)(p >> 31) + wv;
                        const int off = ((iu * RP + (r0 - w)) * d.k + j) * 4 + A.pad_bytes;
                        dl[u] = (int)dm[erow0 + (int64_t)(t + u) * dT + dcol];
#pragma unroll
                        for (int q = 0; q < RC; ++q) vals[u][q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, off + q * rowbytes, 0, AUX);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (t + u < t_hi) {
                        const int wu = (int)((wmask >> u) & 1u);
                        relax_select<RC>(vals[u], dl[u], ord_rank(t + u, evr), r0, wu + wv, RP, bval, bord);
                    }
                }
            }
        }
    }
    if (COOP == 2) {                                                    // partial bests of waves 1..3 -> wave 0
        if (part > 0) {
#pragma unroll
            for (int q = 0; q < RC; ++q) ex[((part - 1) * RC + q) * 64 + lane] = make_uint2((uint32_t)bval[q], bord[q]);
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const uint2 o = ex[(p * RC + q) * 64 + lane];
                const int ov = (int)o.x;
                const bool take = (ov > bval[q]) | ((ov == bval[q]) & (o.y > bord[q]));
                bval[q] = take ? ov : bval[q];
                bord[q] = take ? o.y : bord[q];
            }
        }
    }
    if (PROF) q2 = __builtin_amdgcn_s_memtime();
    // segmented max over lanes with equal destination column (lanes of a column are adjacent)
    if (steps > 0) {                                                    // first step: distance 1, DPP
        const int oj2 = lane_down1(j2);
        const bool same = (lane + 1 < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int ov = lane_down1(bval[q]);
            const uint32_t oo = (uint32_t)lane_down1((int)bord[q]);
            const bool take = same & ((ov > bval[q]) | ((ov == bval[q]) & (oo > bord[q])));
            bval[q] = take ? ov : bval[q];
            bord[q] = take ? oo : bord[q];
        }
    }
    for (int st = 1, sh = 2; st < steps; ++st, sh <<= 1) {
        const int oj2 = __shfl_down(j2, sh);
        const bool same = (lane + sh < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int ov = __shfl_down(bval[q], sh);
            const uint32_t oo = (uint32_t)__shfl_down((int)bord[q], sh);
            const bool take = same & ((ov > bval[q]) | ((ov == bval[q]) & (oo > bord[q])));
            bval[q] = take ? ov : bval[q];
            bord[q] = take ? oo : bord[q];
        }
    }
    const int pj2 = lane_up1(j2);
    const bool head = act & ((lane == 0) | (pj2 != j2));
    unsigned long long dsum = 0;
    if (PROF) q3 = __builtin_amdgcn_s_memtime();
    if (head) {
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int r2 = r0 + q;
            if (r2 < RP) {
                const int idx = (i2 * RP + r2) * d.k2 + j2;            // fast form: a state buffer is < 2 GB, 32-bit indices
                nxt[idx] = bval[q];
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + idx], ~bord[q]); else A.bp[d.bp_off + idx] = (uint16_t)~bord[q]; }
                if (DIGEST && bval[q] != NEG_INF) {
                    const unsigned long long o = ((unsigned long long)r2 * d.k2 + i2) * d.k2 + j2;   // oracle's r-major index
                    dsum += (unsigned long long)(uint32_t)(bval[q] + 1) * (o + 1);
                }
            }
        }
    }
    if (g == 0 && d.ndead > 0) {                                        // columns nobody owns: unreachable
        for (int t = lane; t < d.ndead * RC; t += 64) {
            const int q = t % RC, c = A.dead_cols[d.dead_first + t / RC];
            if (r0 + q < RP) {
                const int64_t idx = ((int64_t)i2 * RP + r0 + q) * d.k2 + c;
                nxt[idx] = NEG_INF;
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + idx], 0xFFFFu); else A.bp[d.bp_off + idx] = (uint16_t)0xFFFFu; }
            }
        }
    }
    if (DIGEST && dsum) atomicAdd(&A.digest[lvl], dsum);
    if (PROF) { q4 = __builtin_amdgcn_s_memtime(); pp[0] += q1 - q0; pp[1] += q2 - q1; pp[2] += q3 - q2; pp[3] += q4 - q3; }
}

// General variant: any in-degree (see the notes on giant columns above); used for levels tagged fast_ok == 2.
template <int RC, bool DIGEST, int AUX, bool PROF = false, int COOP = 0>
__device__ __forceinline__ void sweep_task_general(const FastArgs &A, const LevelDesc &d, __amdgpu_buffer_rsrc_t cur_rsrc,
                                           int32_t *__restrict__ nxt, int i2, int g, int r0, int lvl, unsigned long long *pp = nullptr,
                                           int part = 0, uint2 *ex = nullptr) {
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
    if (PROF) q0 = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x & 63;
    const int RP = A.RP;
    const uint4 rr = A.rowrec[d.b0 + i2];                               // {eu0, du, pu0, pu1}
    uint2 sl = A.slots[d.slot_first + (int64_t)g * 64 + lane];
    // steps field: 0..6 = log2 steps of the segmented max; 15 = first block of a giant column (in-degree > 64: its
    // in-edges fill several consecutive blocks, all walked by THIS wave); 14 = continuation block (nothing to do)
    int steps = __builtin_amdgcn_readfirstlane((int)(sl.y >> 28));
    if (steps == 14) return;
    const int j2 = (sl.x != 0xFFFFFFFFu) ? (int)((sl.x >> 16) & 0x7FFFu) : -1 - lane;
    int nblk = 1;
    if (steps == 15) {
        const int col = __builtin_amdgcn_readfirstlane(j2);
        nblk = ((int)A.rowrec[d.b0 + col].y + 63) >> 6;
        steps = 6;
    }
    const bool has_delta = d.delta_off >= 0;
    const uint16_t *dm = has_delta ? A.delta + d.delta_off : A.delta_zero;   // (A.delta is biased by the resident delta window)
    const int dT = has_delta ? d.T : 0;
    const int du = (int)rr.y;
    if (COOP == 1 && du > COOP_MIN) return;
    const int t_lo = COOP == 2 ? (du * part) >> 2 : 0, t_hi = COOP == 2 ? (du * (part + 1)) >> 2 : du;   // this wave's share of the row's in-edges
    if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); q1 = __builtin_amdgcn_s_memtime(); }
    int bval[RC];
    uint32_t bord[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q) { bval[q] = NEG_INF; bord[q] = 0; }
    const int64_t erow0 = (int64_t)(rr.x - d.in_base) * dT;
    const int rowbytes = d.k * 4;
    bool act = false;
    // Source rows are r = r2 - w.  Byte offsets are relative to the padded buffer start (the resource), so rows
    // r = -1, -2 land in the front padding and r2 >= RP (ragged last chunk) in the tail padding; the select
    // discards what is out of range, which lets every load be issued unconditionally, back to back.
    for (int blk = 0; blk < nblk; ++blk) {
        if (blk > 0) sl = A.slots[d.slot_first + (int64_t)(g + blk) * 64 + lane];
        const bool actb = sl.x != 0xFFFFFFFFu;
        act |= actb;
        const int j = (int)(sl.x & 0x7FFFu), wv = (int)((sl.x >> 15) & 1u);
        const int dcol = has_delta ? (int)(sl.y & 0x000FFFFFu) : 0;
        const int evr = (int)((sl.y >> 20) & 0xFFu);
        if (du <= 2) {
            // 97 % of the rows: both in-edges came with the row record, no loop, no extra load
            if (actb && du > 0) {
                const int ia = (int)(rr.z & 0x7FFFu), wa = (int)(rr.z >> 31) + wv;
                const int ib = (int)(rr.w & 0x7FFFu), wb = (int)(rr.w >> 31) + wv;
                const int offa = ((ia * RP + (r0 - wa)) * d.k + j) * 4 + A.pad_bytes;
                const int offb = ((ib * RP + (r0 - wb)) * d.k + j) * 4 + A.pad_bytes;
                int va[RC], vb[RC];
                const int dla = (int)dm[erow0 + dcol];
                int dlb = 0;
#pragma unroll
                for (int q = 0; q < RC; ++q) va[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offa + q * rowbytes, 0, AUX);
                if (du == 2) {
                    dlb = (int)dm[erow0 + dT + dcol];
#pragma unroll
                    for (int q = 0; q < RC; ++q) vb[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offb + q * rowbytes, 0, AUX);
                }
                relax_select<RC>(va, dla, ord_rank(0, evr), r0, wa, RP, bval, bord);
                if (du == 2) relax_select<RC>(vb, dlb, ord_rank(1, evr), r0, wb, RP, bval, bord);
            }
        } else {
            // heavy rows (recombination fan-in): the in-edge list is fetched 64 at a time, one per lane, and broadcast
            // with readlane; U in-edges per step -- all their loads (U deltas + U*RC values) go out back to back, then
            // the selects run; (value, ord) max is associative and commutative, so the order inside a step is
            // irrelevant.  Small RC leaves registers for a deep step: in-degree 23 takes 2 steps at RC = 1.
            constexpr int U = RC >= 4 ? 2 : 4;                          // (deeper steps spill SGPRs in this variant)
            for (int c0 = t_lo; c0 < t_hi; c0 += 64) {
                const int dc = min(64, t_hi - c0);
                uint32_t mypu = 0;
                if (lane < dc) mypu = A.in_edge[rr.x + c0 + lane];
                for (int t = 0; t < dc; t += U) {
                    uint32_t pu[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) pu[u] = (uint32_t)__builtin_amdgcn_readlane((int)mypu, min(t + u, dc - 1));
                    if (actb) {
                        int vals[U][RC], dl[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (t + u < dc) {                           // wave-uniform
                                const int iu = (int)(pu[u] & 0x7FFFFFFFu), w = (int)(pu[u] >> 31) + wv;
                                const int off = ((iu * RP + (r0 - w)) * d.k + j) * 4 + A.pad_bytes;
                                dl[u] = (int)dm[erow0 + (int64_t)(c0 + t + u) * dT + dcol];
#pragma unroll
                                for (int q = 0; q < RC; ++q) vals[u][q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, off + q * rowbytes, 0, AUX);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (t + u < dc) {
                                const int wu = (int)(pu[u] >> 31);
                                relax_select<RC>(vals[u], dl[u], ord_rank(c0 + t + u, evr), r0, wu + wv, RP, bval, bord);
                            }
                        }
                    }
                }
            }
        }
    }
    if (COOP == 2) {                                                    // partial bests of waves 1..3 -> wave 0 (see sweep_task)
        if (part > 0) {
#pragma unroll
            for (int q = 0; q < RC; ++q) ex[((part - 1) * RC + q) * 64 + lane] = make_uint2((uint32_t)bval[q], bord[q]);
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const uint2 o = ex[(p * RC + q) * 64 + lane];
                const int ov = (int)o.x;
                const bool take = (ov > bval[q]) | ((ov == bval[q]) & (o.y > bord[q]));
                bval[q] = take ? ov : bval[q];
                bord[q] = take ? o.y : bord[q];
            }
        }
    }
    if (PROF) q2 = __builtin_amdgcn_s_memtime();
    // segmented max over lanes with equal destination column (lanes of a column are adjacent)
    if (steps > 0) {                                                    // first step: distance 1, DPP
        const int oj2 = lane_down1(j2);
        const bool same = (lane + 1 < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int ov = lane_down1(bval[q]);
            const uint32_t oo = (uint32_t)lane_down1((int)bord[q]);
            const bool take = same & ((ov > bval[q]) | ((ov == bval[q]) & (oo > bord[q])));
            bval[q] = take ? ov : bval[q];
            bord[q] = take ? oo : bord[q];
        }
    }
    for (int st = 1, sh = 2; st < steps; ++st, sh <<= 1) {
        const int oj2 = __shfl_down(j2, sh);
        const bool same = (lane + sh < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int ov = __shfl_down(bval[q], sh);
            const uint32_t oo = (uint32_t)__shfl_down((int)bord[q], sh);
            const bool take = same & ((ov > bval[q]) | ((ov == bval[q]) & (oo > bord[q])));
            bval[q] = take ? ov : bval[q];
            bord[q] = take ? oo : bord[q];
        }
    }
    const int pj2 = lane_up1(j2);
    const bool head = act & ((lane == 0) | (pj2 != j2));
    unsigned long long dsum = 0;
    if (PROF) q3 = __builtin_amdgcn_s_memtime();
    if (head) {
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int r2 = r0 + q;
            if (r2 < RP) {
                const int idx = (i2 * RP + r2) * d.k2 + j2;            // fast form: a state buffer is < 2 GB, 32-bit indices
                nxt[idx] = bval[q];
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + idx], ~bord[q]); else A.bp[d.bp_off + idx] = (uint16_t)~bord[q]; }
                if (DIGEST && bval[q] != NEG_INF) {
                    const unsigned long long o = ((unsigned long long)r2 * d.k2 + i2) * d.k2 + j2;   // oracle's r-major index
                    dsum += (unsigned long long)(uint32_t)(bval[q] + 1) * (o + 1);
                }
            }
        }
    }
    if (g == 0 && d.ndead > 0) {                                        // columns nobody owns: unreachable
        for (int t = lane; t < d.ndead * RC; t += 64) {
            const int q = t % RC, c = A.dead_cols[d.dead_first + t / RC];
            if (r0 + q < RP) {
                const int64_t idx = ((int64_t)i2 * RP + r0 + q) * d.k2 + c;
                nxt[idx] = NEG_INF;
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + idx], 0xFFFFu); else A.bp[d.bp_off + idx] = (uint16_t)0xFFFFu; }
            }
        }
    }
    if (DIGEST && dsum) atomicAdd(&A.digest[lvl], dsum);
    if (PROF) { q4 = __builtin_amdgcn_s_memtime(); pp[0] += q1 - q0; pp[1] += q2 - q1; pp[2] += q3 - q2; pp[3] += q4 - q3; }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t state_rsrc(const int32_t *padded_base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)padded_base, 0, bytes, 0x00020000);
}

// per-level launch of the fast form: grid = (ceil(nblocks/4), nchunk, k2), one task per wave (a 3-D grid: splitting a
// combined index would cost a runtime integer division -- ~40 instructions of a 380-instruction task)
template <int RC, bool DIGEST, bool GENERAL, bool COOP = false>
__global__ __launch_bounds__(256) void dp_sweep_fast_kernel(FastArgs A, LevelDesc d, int lvl, const int32_t *__restrict__ heavy_rows = nullptr) {
    const int32_t *cur = ((lvl - 1) & 1) ? A.base1 : A.base0;           // padded allocation starts
    int32_t *nxt = ((lvl & 1) ? A.base1 : A.base0) + A.pad_bytes / 4;
    const int r0 = (int)blockIdx.y * RC;                                // chunks of a row are neighbours in dispatch order: they share its delta row
    if (COOP && (int)blockIdx.z >= d.k2) {
        // cooperative region (blockDim = 4 waves): workgroup (x, y, k2 + 4 h + b) = slot block 4 x + b of the h-th heavy row
        __shared__ uint2 ex[3 * RC * 64];
        const int hz = (int)blockIdx.z - d.k2;
        const int g = (int)blockIdx.x * 4 + (hz & 3);
        if (g >= d.nblocks) return;                                     // workgroup-uniform: nobody is left at the barrier
        const int i2 = heavy_rows[d.heavy_first + (hz >> 2)];
        if (GENERAL) sweep_task_general<RC, DIGEST, 0, false, 2>(A, d, state_rsrc(cur, A.buf_bytes), nxt, i2, g, r0, lvl, nullptr, (int)(threadIdx.x >> 6), ex);
        else sweep_task<RC, DIGEST, 0, false, 2>(A, d, state_rsrc(cur, A.buf_bytes), nxt, i2, g, r0, lvl, nullptr, (int)(threadIdx.x >> 6), ex);
        return;
    }
    const int g = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    if (g >= d.nblocks) return;                                         // wave-uniform; no block barrier below
    const int i2 = (int)blockIdx.z;
    if (GENERAL) sweep_task_general<RC, DIGEST, 0, false, COOP ? 1 : 0>(A, d, state_rsrc(cur, A.buf_bytes), nxt, i2, g, r0, lvl);
    else sweep_task<RC, DIGEST, 0, false, COOP ? 1 : 0>(A, d, state_rsrc(cur, A.buf_bytes), nxt, i2, g, r0, lvl);
}

// ---------------------------------------------------------------------------------------------
// Team form: ONE launch for the whole level chain.  Workgroups read their XCD id (HW_REG_XCC_ID) and
// only those that share the elected XCD stay: an XCD's CUs share one L2, so inside the team a level
// hand-off needs no cache write-back/invalidate -- producers' plain stores are in the L2 once their
// vmcnt drains, consumers read the state with sc1 loads (L2-served, never from the CU's L1), and the
// level barrier is one L2 atomic counter.  Read-only graph tables stay warm in L1/L2 for the whole
// run, which is what the per-level launches cannot offer (every kernel starts cold).  Correctness does
// not depend on where the dispatcher puts workgroups: a team is DEFINED by the hardware XCC id its
// members read, whatever its size; every spin is bounded and reports through ctl->error, in which
// case the host falls back to per-level launches.
// ---------------------------------------------------------------------------------------------
struct TeamCtl {
    uint32_t registered, leader_xcc_plus1, bar, error;
    uint32_t team_count[8];
    unsigned long long t_cycles, t_real;                // diagnostic: shader cycles / 100 MHz ticks spent in the level loop (rank 0)
    unsigned long long phase[6], tphase[4];             // diagnostic (PROF build): cycles per phase, rank 0 wave 0
};

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int RC, bool PROF, int AUX = 16>
__global__ __launch_bounds__(512) void dp_team_kernel(FastArgs A, const LevelDesc *__restrict__ descs, int lvl_begin, int lvl_end, TeamCtl *ctl) {
    __shared__ int s_rank, s_size, s_go;
    const int wave = (int)(threadIdx.x >> 6), nwave_wg = (int)(blockDim.x >> 6);
    if (threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        xcc &= 7u;
        const uint32_t rank = atomicAdd(&ctl->team_count[xcc], 1u);
        atomicCAS(&ctl->leader_xcc_plus1, 0u, xcc + 1u);
        __threadfence();
        atomicAdd(&ctl->registered, 1u);
        int go = -1;
        for (uint32_t spin = 0; spin < (1u << 22); ++spin) {            // every workgroup must be resident: bounded
            if (ld_sc1(&ctl->registered) >= gridDim.x || ld_sc1(&ctl->error)) { go = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (go < 0) { atomicExch(&ctl->error, 1u); go = 0; }
        if (ld_sc1(&ctl->error)) go = 0;
        const uint32_t work = ld_sc1(&ctl->leader_xcc_plus1) - 1u;
        s_rank = (int)rank;
        s_size = (int)ld_sc1(&ctl->team_count[work & 7u]);
        s_go = (go && work == xcc) ? 1 : 0;
    }
    __syncthreads();
    if (!s_go) return;
    const int rank = s_rank, size = s_size;
    const int wave_id = rank * nwave_wg + wave, n_waves = size * nwave_wg;
    const int nchunk = (A.RP + RC - 1) / RC;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tp[4] = {0, 0, 0, 0};
    for (int lvl = lvl_begin; lvl < lvl_end; ++lvl) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (PROF) t0 = __builtin_amdgcn_s_memtime();
        const LevelDesc d = descs[lvl];
        if (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); }
        const int32_t *cur = ((lvl - 1) & 1) ? A.base1 : A.base0;
        int32_t *nxt = ((lvl & 1) ? A.base1 : A.base0) + A.pad_bytes / 4;
        const __amdgpu_buffer_rsrc_t rs = state_rsrc(cur, A.buf_bytes);
        // A lone wave issues ~1 instruction per 4-8 cycles, so on narrow levels the RC-fold unrolled task IS the
        // critical path: split the recombination counts over more waves (chunks of 1 or 4) while the team has idle waves.
        const int base_tasks = d.k2 * d.nblocks;
        if (base_tasks * A.RP <= n_waves) {
            const int per_row = d.nblocks * A.RP, ntask = d.k2 * per_row;
            for (int task = wave_id; task < ntask; task += n_waves) {
                const int i2 = task / per_row, rem = task - i2 * per_row;
                sweep_task<1, false, AUX, PROF>(A, d, rs, nxt, i2, rem / A.RP, rem % A.RP, lvl, tp);
            }
        } else if (RC > 4 && base_tasks * ((A.RP + 3) / 4) <= 2 * n_waves) {
            const int nc4 = (A.RP + 3) / 4, per_row = d.nblocks * nc4, ntask = d.k2 * per_row;
            for (int task = wave_id; task < ntask; task += n_waves) {
                const int i2 = task / per_row, rem = task - i2 * per_row;
                sweep_task<4, false, AUX, PROF>(A, d, rs, nxt, i2, rem / nc4, (rem % nc4) * 4, lvl, tp);
            }
        } else {
            const int per_row = d.nblocks * nchunk, ntask = d.k2 * per_row;
            for (int task = wave_id; task < ntask; task += n_waves) {
                const int i2 = task / per_row, rem = task - i2 * per_row;
                const int g = nchunk == 1 ? rem : rem / nchunk;
                const int r0 = nchunk == 1 ? 0 : (rem % nchunk) * RC;
                sweep_task<RC, false, AUX, PROF>(A, d, rs, nxt, i2, g, r0, lvl, tp);
            }
        }
        if (PROF) t2 = __builtin_amdgcn_s_memtime();
        // level barrier inside the team (one XCD): drain this wave's stores into the L2, then count
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (PROF) t3 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (PROF) t4 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) {
            atomicAdd(&ctl->bar, 1u);
            const uint32_t want = (uint32_t)size * (uint32_t)(lvl - lvl_begin + 1);
            uint32_t spin = 0;
            while (ld_sc1(&ctl->bar) < want) {
                if (++spin > (1u << 22) || ld_sc1(&ctl->error)) { atomicExch(&ctl->error, 2u); break; }
            }
        }
        __syncthreads();
        if (PROF) { const unsigned long long t5 = __builtin_amdgcn_s_memtime(); ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; }
        if (ld_sc1(&ctl->error)) return;
        if (PROF) ph[5] += __builtin_amdgcn_s_memtime() - t0;
    }
    if (PROF && rank == 0 && threadIdx.x == 0) { for (int q = 0; q < 6; ++q) ctl->phase[q] = ph[q]; for (int q = 0; q < 4; ++q) ctl->tphase[q] = tp[q]; }
    if (rank == 0 && threadIdx.x == 0) { ctl->t_cycles = __builtin_amdgcn_s_memtime() - c0; ctl->t_real = __builtin_amdgcn_s_memrealtime() - w0; }
}

// plain launch: one level per launch, level given by argument
template <int RC, bool DIGEST>
__global__ __launch_bounds__(256) void dp_sweep_kernel(SweepArgs A, int lvl) {
    sweep_level_pairs<RC, DIGEST>(A, lvl, (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), (int)(gridDim.x * 4));
}

__global__ void dp_init_kernel(int32_t *cur, int RP) {   // level 0: k = 1, every r starts at 0 (:534-535)
    if ((int)threadIdx.x < RP) cur[threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------------------------
// traceback   (approximator.cpp:757-785)
// Chain kernel (one wave): walk the back-pointer lattice downwards from a known cell.  The chain is one
// dependent HBM load per level, so everything else is kept off it: the level descriptors of the next 64
// levels are fetched one per lane and broadcast with readlane, and the hop words are parked in path[].
// In segmented mode it is called once per segment, last segment first, carrying the cell in `st`.
// Finish kernel (1024 threads, levels in parallel): re-derive s_het from the colour lists of the winning
// edge pairs (:662) and emit the weighted edges (:673-692; both final edges unconditionally) as
// (level, from, to, which) records; the host orders them by level.
// ---------------------------------------------------------------------------------------------
struct ChainState { int32_t i, j, r, value; };
constexpr int32_t CHAIN_CORRUPT = INT32_MIN;            // ChainState::value after a hop left its level

// Pulls the row records of a range of levels into the memory-side Infinity Cache right before the chain walk reads two
// of them per level (they were last touched by the sweep, hundreds of GB of lattice writes ago).
__global__ __launch_bounds__(256) void dp_warm_kernel(const uint4 *__restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint4 v = p[i]; asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
}

// Sweep look-ahead: streams the graph tables (row records, slot records, in-edges, score deltas) of a batch of upcoming
// levels through the memory-side Infinity Cache.  Every table byte is read exactly once per pass, so without this each
// level's two dependent load rounds go all the way to HBM; the batch is a few MB, read at full chip bandwidth.
struct WarmRanges { const char *p[4]; long long n16[4]; };            // start (16-byte aligned down) and length in 16-byte units
__global__ __launch_bounds__(256) void dp_warm_tables_kernel(WarmRanges W) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 *p = (const uint4 *)W.p[q];
        for (long long i = tid; i < W.n16[q]; i += nth) { uint4 v = p[i]; asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
    }
}

__global__ __launch_bounds__(64) void dp_trace_chain_kernel(const LevelDesc *__restrict__ descs, int l_hi, int l_lo, int RP, int R,
                                                            const uint16_t *__restrict__ bp /* biased by the segment's first unit */,
                                                            const int32_t *__restrict__ final_val /* non-null on the first call */,
                                                            const uint4 *__restrict__ rowrec, const uint32_t *__restrict__ in_edge,
                                                            uint32_t *__restrict__ path, ChainState *st) {
    const int lane = threadIdx.x & 63;
    int i, j, r, value, forced = 0;                                     // forced: bit 0 row, bit 1 column has one in-edge
    if (final_val) { value = final_val[(int64_t)R * descs[l_hi].k2]; i = 0; j = 0; r = R; }   // sink level, layout [i][r][j]: cell (0, R, 0)
    else { i = st->i; j = st->j; r = st->r; value = st->value; }
    if (value != NEG_INF && value != CHAIN_CORRUPT) {
        for (int base = l_hi; base >= l_lo && value != CHAIN_CORRUPT; base -= 64) {
            const int my_l = base - lane;
            long long bo = 0;
            int kk = 1, bb = 0, kp = 1;                                 // bb = first vertex of the level | wide << 31; kp = source width
            if (my_l >= l_lo) { bo = descs[my_l].bp_off; kk = descs[my_l].k2; kp = descs[my_l].k; bb = descs[my_l].b0 | (descs[my_l].bp_wide << 31); }
            const int n = min(64, base - l_lo + 1);
            for (int t = 0; t < n; ++t) {
                const int l = base - t;
                const long long bo_l = ((long long)__builtin_amdgcn_readlane((int)(bo >> 32), t) << 32) |
                                       (unsigned int)__builtin_amdgcn_readlane((int)bo, t);
                const int k2 = __builtin_amdgcn_readlane(kk, t);
                const int bw = __builtin_amdgcn_readlane(bb, t);
                const long long cell = ((long long)i * RP + r) * k2 + j;
                uint32_t hop;
                if (bw < 0) {                                           // wide level: the hop word itself
                    hop = *(const uint32_t *)(bp + bo_l + 2 * cell);
                    forced = 0;
                } else {
                    // One round trip: the back-pointer and the two row records go out together (even lanes fetch the
                    // row's record, odd lanes the column's -- lane-varying addresses, so the compiler can neither turn
                    // them into scalar loads nor defer them behind the back-pointer); ranks 0 and 1 (97 % of the
                    // vertices) are inside the record, higher ranks cost one more load.  When the previous hop said that
                    // row and column both have a single in-edge the winner is (0, 0) whatever the lattice says: no
                    // back-pointer load, the step costs an Infinity-Cache hit instead of an HBM miss.
                    const int odd = lane & 1;
                    const uint4 rr = rowrec[bw + (odd ? j : i)];
                    uint32_t p;
                    if (forced == 3) {
                        p = rr.z;
                    } else {
                        const uint32_t b = bp[bo_l + cell];
                        uint32_t e0 = rr.x;
                        asm volatile("" : "+v"(e0));                    // keep the whole record in the first round trip
                        const uint32_t rank = odd ? (b & 0xFFu) : (b >> 8);
                        p = rank == 0 ? rr.z : rr.w;
                        if (rank > 1) p = in_edge[e0 + rank];           // (no flag bit in the full in-edge array)
                    }
                    const uint32_t pu = (uint32_t)__builtin_amdgcn_readlane((int)p, 0), pv = (uint32_t)__builtin_amdgcn_readlane((int)p, 1);
                    hop = (pu & 0x7FFFu) | ((pv & 0x7FFFu) << 15) | ((pu >> 31) << 30) | ((pv >> 31) << 31);
                    forced = (int)((pu >> 16) & 1u) | (int)(((pv >> 16) & 1u) << 1);
                }
                if (lane == 0) path[l] = hop;
                i = (int)(hop & 0x7FFFu); j = (int)((hop >> 15) & 0x7FFFu);
                r -= (int)((hop >> 30) & 1u) + (int)(hop >> 31);
                // a hop that leaves the source level means the lattice is corrupt (a level nobody swept): stop before the next
                // load goes wild -- the host reports DG_ERR_STATE instead of the GPU faulting
                const int kprev = __builtin_amdgcn_readlane(kp, t);
                if ((i >= kprev) | (j >= kprev) | (r < 0)) { value = CHAIN_CORRUPT; i = j = 0; r = 0; break; }
            }
        }
    }
    if (lane == 0) { st->i = i; st->j = j; st->r = r; st->value = value; }
}

// Speculative form of the chain walk (the default).  A step of the walk above costs one HBM round trip because the next
// cell's address needs this level's back-pointer.  But the candidates are few: with at most two in-edges into the row
// and into the column (97 % of the vertices) the predecessor is one of four cells, all known as soon as the two row
// records are in.  Lanes 0..3 (mirrored by the other lanes) therefore load the back-pointer AND the row records of
// "their" candidate one level ahead; when this level's back-pointer arrives it only selects the lane whose loads are
// already in flight, so two levels' HBM round trips overlap.  A rank above 1 or a wide level resolves the predecessor
// with extra loads and re-issues exact loads for it.
// Shape of the code (it matters: the compiler derives its s_waitcnt from it): every path through a step ends with the
// three loads (row record, column record, back-pointer word) of the next level as the most recent memory operations, and
// a step consumes only what the previous step issued; the step is instantiated twice with the two register sets swapped,
// because copying a register that still has a load in flight would wait for that load.
struct ChainWalk {
    int i, j, r, value, csel;
};
struct ChainRegs { uint4 row, col; uint32_t b; };
struct ChainDesc { long long bo; int k2, bw, kp; };    // bp offset, width, first vertex | wide << 31, source width

// loads of one cell (level descriptor bo/k2/bw): row records + the aligned 32-bit word that holds its back-pointer
__device__ __forceinline__ void chain_issue(ChainRegs &Y, const uint16_t *__restrict__ bp, const uint4 *__restrict__ rowrec, long long bo, int k2, int bw,
                                            int RP, int ci, int cj, int cr) {
    const int b0 = bw & 0x7FFFFFFF;
    const long long unit = bo + (((long long)ci * RP + cr) * k2 + cj) * (bw < 0 ? 2 : 1);
    Y.row = rowrec[b0 + ci];
    Y.col = rowrec[b0 + cj];
    Y.b = *(const uint32_t *)(bp + (unit & ~1LL));
}

__device__ __forceinline__ void chain_spec_step(ChainWalk &W, int l, int l_lo, int RP, const ChainDesc &D, const ChainDesc &N,
                                                const uint16_t *__restrict__ bp, const uint4 *__restrict__ rowrec, const uint32_t *__restrict__ in_edge,
                                                uint32_t &hops, int slot, int lane, ChainRegs &X, ChainRegs &Y) {
    const int ca = lane & 1, cb = (lane >> 1) & 1;                      // this lane's candidate: rank ca of the row, cb of the column
    const long long bo_l = D.bo, nbo = N.bo;
    const int k2 = D.k2, bw = D.bw, kprev = D.kp, nk2 = N.k2, nbw = N.bw;
    const int cs = W.csel;
    const uint32_t rix = (uint32_t)__builtin_amdgcn_readlane((int)X.row.x, cs), riy = (uint32_t)__builtin_amdgcn_readlane((int)X.row.y, cs);
    const uint32_t riz = (uint32_t)__builtin_amdgcn_readlane((int)X.row.z, cs), riw = (uint32_t)__builtin_amdgcn_readlane((int)X.row.w, cs);
    const uint32_t rjx = (uint32_t)__builtin_amdgcn_readlane((int)X.col.x, cs), rjy = (uint32_t)__builtin_amdgcn_readlane((int)X.col.y, cs);
    const uint32_t rjz = (uint32_t)__builtin_amdgcn_readlane((int)X.col.z, cs), rjw = (uint32_t)__builtin_amdgcn_readlane((int)X.col.w, cs);
    // one level ahead: the four candidate cells of level l - 1; a lane without a valid candidate re-reads this level's cell
    const bool have_next = l - 1 >= l_lo;
    const uint32_t wa = ca ? riw : riz, wb = cb ? rjw : rjz;
    const int ci = (int)(wa & 0x7FFFu), cj = (int)(wb & 0x7FFFu), cr = W.r - (int)(wa >> 31) - (int)(wb >> 31);
    const bool ok = have_next & ((uint32_t)ca < riy) & ((uint32_t)cb < rjy) & (cr >= 0) & (ci < kprev) & (cj < kprev);
    chain_issue(Y, bp, rowrec, ok ? nbo : bo_l, ok ? nk2 : k2, ok ? nbw : bw, RP, ok ? ci : W.i, ok ? cj : W.j, ok ? cr : W.r);
    // this level's back-pointer word (issued one step ago)
    const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)X.b, cs);
    uint32_t hop;
    bool exact_next = false;                                            // the predecessor is not among the candidates in flight
    if (bw < 0) {                                                       // wide level: the word is the hop
        hop = word;
        exact_next = true;
    } else {
        const long long unit = bo_l + ((long long)W.i * RP + W.r) * k2 + W.j;
        const uint32_t bv = (unit & 1) ? (word >> 16) : (word & 0xFFFFu);
        const uint32_t eu = bv >> 8, ev = bv & 0xFFu;
        if (eu >= riy || ev >= rjy) { W.value = CHAIN_CORRUPT; W.i = W.j = 0; W.r = 0; return; }   // (0xFFFF = unreachable lands here too)
        const uint32_t pu = eu == 0 ? riz : (eu == 1 ? riw : in_edge[rix + eu]);
        const uint32_t pv = ev == 0 ? rjz : (ev == 1 ? rjw : in_edge[rjx + ev]);
        hop = (pu & 0x7FFFu) | ((pv & 0x7FFFu) << 15) | ((pu >> 31) << 30) | ((pv >> 31) << 31);
        exact_next = (eu > 1) | (ev > 1);
        W.csel = (int)((eu & 1u) | ((ev & 1u) << 1));
    }
    hops = lane == slot ? hop : hops;                                   // lane t keeps the hop of the batch's t-th level: one store per batch,
    W.i = (int)(hop & 0x7FFFu); W.j = (int)((hop >> 15) & 0x7FFFu);     // none inside the walk (stores share the loads' counter)
    W.r -= (int)((hop >> 30) & 1u) + (int)(hop >> 31);
    // a hop that leaves the source level means the lattice is corrupt (a level nobody swept): stop before the next
    // load goes wild -- the host reports DG_ERR_STATE instead of the GPU faulting
    if ((W.i >= kprev) | (W.j >= kprev) | (W.r < 0)) { W.value = CHAIN_CORRUPT; W.i = W.j = 0; W.r = 0; return; }
    if (have_next && (exact_next || nbw < 0)) {                         // rare: exact loads of the predecessor replace the candidates
        chain_issue(Y, bp, rowrec, nbo, nk2, nbw, RP, W.i, W.j, W.r);
        W.csel = 0;
    }
}

__global__ __launch_bounds__(64) void dp_trace_chain_spec_kernel(const LevelDesc *__restrict__ descs, int l_hi, int l_lo, int RP, int R,
                                                                 const uint16_t *__restrict__ bp /* biased by the segment's first unit */,
                                                                 const int32_t *__restrict__ final_val /* non-null on the first call */,
                                                                 const uint4 *__restrict__ rowrec, const uint32_t *__restrict__ in_edge,
                                                                 uint32_t *__restrict__ path, ChainState *st) {
    const int lane = threadIdx.x & 63;
    ChainWalk W;
    W.csel = 0;
    if (final_val) { W.value = final_val[(int64_t)R * descs[l_hi].k2]; W.i = 0; W.j = 0; W.r = R; }   // sink level, layout [i][r][j]: cell (0, R, 0)
    else { W.i = st->i; W.j = st->j; W.r = st->r; W.value = st->value; }
    if (W.value != NEG_INF && W.value != CHAIN_CORRUPT) {
        ChainRegs A, B;
        bool first = true;
        for (int base = l_hi; base >= l_lo && W.value != CHAIN_CORRUPT; base -= 56) {
            const int my_l = base - lane;                               // 64 descriptors, 56 levels (a multiple of 8) per batch
            long long bo = 0;
            int kk = 1, bb = 0, kp = 1;                                 // bb = first vertex of the level | wide << 31; kp = source width
            if (my_l >= l_lo) { bo = descs[my_l].bp_off; kk = descs[my_l].k2; kp = descs[my_l].k; bb = descs[my_l].b0 | (descs[my_l].bp_wide << 31); }
            asm volatile("" ::"v"(bo), "v"(kk), "v"(bb), "v"(kp));      // descriptors complete before the walk (no wait inside the loop)
            const int n = min(56, base - l_lo + 1);
            uint32_t hops = 0;
#define DG_DESC(T) ChainDesc{((long long)__builtin_amdgcn_readlane((int)(bo >> 32), (T)) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)bo, (T)), \
                            __builtin_amdgcn_readlane(kk, (T)), __builtin_amdgcn_readlane(bb, (T)), __builtin_amdgcn_readlane(kp, (T))}
            ChainDesc D0 = DG_DESC(0), D1;                              // descriptor of the level at hand / one level ahead, rotated like the registers
            if (first) {                                                // prologue: exact loads of the starting cell
                chain_issue(A, bp, rowrec, D0.bo, D0.k2, D0.bw, RP, W.i, W.j, W.r);
                first = false;
            }
            // unrolled by hand (8 steps per trip): the loop's back edge copies the registers of the loads in flight, and so
            // waits for them -- one un-overlapped step per trip
#define DG_CHAIN_STOP(T) (W.value == CHAIN_CORRUPT || (T) >= n)
#define DG_CHAIN_STEP(T, X, Y, DC, DN) DN = DG_DESC((T) + 1); chain_spec_step(W, base - (T), l_lo, RP, DC, DN, bp, rowrec, in_edge, hops, (T), lane, X, Y)
            for (int t = 0; t < n; t += 8) {
                DG_CHAIN_STEP(t, A, B, D0, D1);     if (DG_CHAIN_STOP(t + 1)) break;
                DG_CHAIN_STEP(t + 1, B, A, D1, D0); if (DG_CHAIN_STOP(t + 2)) break;
                DG_CHAIN_STEP(t + 2, A, B, D0, D1); if (DG_CHAIN_STOP(t + 3)) break;
                DG_CHAIN_STEP(t + 3, B, A, D1, D0); if (DG_CHAIN_STOP(t + 4)) break;
                DG_CHAIN_STEP(t + 4, A, B, D0, D1); if (DG_CHAIN_STOP(t + 5)) break;
                DG_CHAIN_STEP(t + 5, B, A, D1, D0); if (DG_CHAIN_STOP(t + 6)) break;
                DG_CHAIN_STEP(t + 6, A, B, D0, D1); if (DG_CHAIN_STOP(t + 7)) break;
                DG_CHAIN_STEP(t + 7, B, A, D1, D0); if (DG_CHAIN_STOP(t + 8)) break;
            }
#undef DG_CHAIN_STOP
#undef DG_CHAIN_STEP
#undef DG_DESC
            if (lane < n && W.value != CHAIN_CORRUPT) path[base - lane] = hops;
        }
    }
    if (lane == 0) { st->i = W.i; st->j = W.j; st->r = W.r; st->value = W.value; }
}

__global__ __launch_bounds__(1024) void dp_trace_finish_kernel(const LevelDesc *__restrict__ descs, int L, const uint32_t *__restrict__ path,
                                                               ColourCsr col, int cap_e, int32_t *__restrict__ edges /* 4*cap_e */,
                                                               const ChainState *st, TraceOut *out) {
    __shared__ int s_shet, s_ne;
    if (threadIdx.x == 0) { s_shet = 0; s_ne = 0; }
    __syncthreads();
    const int value = st->value;
    if (value != NEG_INF && value != CHAIN_CORRUPT) {
        int shet = 0;
        for (int l = 1 + (int)threadIdx.x; l < L; l += (int)blockDim.x) {
            const uint32_t b = path[l];
            int i = 0, j = 0;                                         // destination cell at level l = predecessor recorded at l+1
            if (l < L - 1) { const uint32_t nb = path[l + 1]; i = (int)(nb & 0x7FFFu); j = (int)((nb >> 15) & 0x7FFFu); }
            const int pi = (int)(b & 0x7FFFu), pj = (int)((b >> 15) & 0x7FFFu);
            const int wu = (int)((b >> 30) & 1u), wv = (int)(b >> 31);
            const LevelDesc d = descs[l];
            const int u1 = d.a0 + pi, v1 = d.a0 + pj, u2 = d.b0 + i, v2 = d.b0 + j;
            if (d.delta_off >= 0) shet += score_symd(col, u1, v1, u2, v2);
            const int reps = (l == L - 1) ? 1 : 0;
            for (int q = 0; q < reps + wu; ++q) {
                const int e = atomicAdd(&s_ne, 1);
                if (e < cap_e) { edges[e] = l; edges[cap_e + e] = u1; edges[2 * cap_e + e] = u2; edges[3 * cap_e + e] = 0; }
            }
            for (int q = 0; q < reps + wv; ++q) {
                const int e = atomicAdd(&s_ne, 1);
                if (e < cap_e) { edges[e] = l; edges[cap_e + e] = v1; edges[2 * cap_e + e] = v2; edges[3 * cap_e + e] = 1; }
            }
        }
        if (shet) atomicAdd(&s_shet, shet);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        TraceOut o;
        o.value = value; o.s_het = s_shet; o.n_e = s_ne; o.overflow = s_ne > cap_e ? 1 : 0;
        *out = o;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t s) {
    if (int rc = b.ensure(bytes)) return rc;
    if (bytes) DG_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
    return DG_OK;
}

// Ask for `target` chunks (the latest request wins): starts the allocation thread if chunks are missing.  Chunks
// already mapped beyond the target are kept (hipFree of 8 GB costs ~0.1 s) unless pool_trim is called.  Returns at once.
static void pool_request(DpState &S, int device, size_t target) {
    std::unique_lock<std::mutex> lk(S.pool.mu);
    S.pool.target = std::max(target, S.pool.chunks.size());
    if (S.pool.running || S.pool.chunks.size() >= S.pool.target) return;
    if (S.pool.th.joinable()) { lk.unlock(); S.pool.th.join(); lk.lock(); }
    S.pool.running = true;
    S.pool.failed = false;
    DpState *Sp = &S;
    S.pool.th = std::thread([Sp, device]() {
        (void)hipSetDevice(device);
        for (;;) {
            size_t bytes;
            {
                std::unique_lock<std::mutex> lk2(Sp->pool.mu);
                Sp->pool.cv.wait(lk2, [&] { return !Sp->pool.paused; });
                if (Sp->pool.chunks.size() >= Sp->pool.target) { Sp->pool.running = false; Sp->pool.cv.notify_all(); return; }
                bytes = Sp->pool.chunk_units * 2;
            }
            void *q = nullptr;
            const hipError_t e = hipMalloc(&q, bytes);
            std::unique_lock<std::mutex> lk2(Sp->pool.mu);
            if (e != hipSuccess) { (void)hipGetLastError(); Sp->pool.failed = true; Sp->pool.running = false; Sp->pool.cv.notify_all(); return; }
            Sp->pool.chunks.push_back(q);
            Sp->pool.cv.notify_all();
            lk2.unlock();
            // HIP calls of other threads queue on a runtime lock this thread would otherwise win again at once
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    });
}
// hipMalloc calls queue behind the one in flight, so a caller with many small allocations to make (dp_load) holds
// the pool thread between chunks meanwhile
struct PoolPause {
    DpState &S;
    explicit PoolPause(DpState &s) : S(s) { std::unique_lock<std::mutex> lk(S.pool.mu); S.pool.paused = true; }
    ~PoolPause() { { std::unique_lock<std::mutex> lk(S.pool.mu); S.pool.paused = false; } S.pool.cv.notify_all(); }
};
// free chunks beyond `keep` (and stop asking for more than that)
static void pool_trim(DpState &S, size_t keep) {
    { std::unique_lock<std::mutex> lk(S.pool.mu); S.pool.target = std::min(S.pool.target, keep); }
    if (S.pool.th.joinable()) S.pool.th.join();               // it stops at the next chunk boundary
    std::unique_lock<std::mutex> lk(S.pool.mu);
    while (S.pool.chunks.size() > keep) { (void)hipFree(S.pool.chunks.back()); S.pool.chunks.pop_back(); }
    S.pool.running = false;
}
// drop every chunk (used before a differently sized pool or the segmented mode takes the memory)
static void pool_clear(DpState &S) { pool_trim(S, 0); }
// wait until chunk c exists; nullptr if the allocation failed
static void *pool_wait(DpState &S, size_t c) {
    std::unique_lock<std::mutex> lk(S.pool.mu);
    S.pool.cv.wait(lk, [&] { return S.pool.chunks.size() > c || S.pool.failed || !S.pool.running; });
    return S.pool.chunks.size() > c ? S.pool.chunks[c] : nullptr;
}

static double wall_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static int dp_load(dg_ctx *c, const dg_dp_graph *g) {
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    double tl0 = wall_s();
    auto lap = [&](const char *what) { if (dbg) { double t = wall_s(); fprintf(stderr, "[dipgenie_hip] load: %-22s %.3f s\n", what, t - tl0); tl0 = t; } };
    if (!g || !g->level_off || !g->out_off || !g->out_dst || !g->out_w || !g->hom_off || !g->het_off) {
        set_error("dg_dp_load_graph: null array"); return DG_ERR_ARG;
    }
    const int nV = g->n_vertices, L = g->n_levels, R = g->R;
    if (nV < 2 || L < 2 || R < 0 || R > 4096) { set_error("dg_dp_load_graph: bad sizes (V=%d L=%d R=%d)", nV, L, R); return DG_ERR_ARG; }
    if (g->level_off[0] != 0 || g->level_off[L] != nV) { set_error("level_off must span [0, n_vertices]"); return DG_ERR_ARG; }
    if (g->level_off[1] != 1) { set_error("level 0 must hold exactly the source vertex"); return DG_ERR_ARG; }
    if (!c->dp) c->dp = new DpState();
    DpState &S = *c->dp;
    graphs_clear(S);
    S.loaded = false;
    S.nV = nV; S.L = L; S.R = R; S.RP = R + 1;
    // Host-side table construction runs on a few std::threads over contiguous LEVEL ranges balanced by vertex count
    // (no OpenMP in this library: the caller may bring its own runtime).  Edges only go from level l to l + 1 and
    // vertex ids are level-sorted, so a range of source levels owns the in-edge lists of the next levels' vertices.
    const int NT = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)S.host_threads, (int64_t)std::thread::hardware_concurrency(), (int64_t)L / 64 + 1}));
    std::vector<int> lcut(NT + 1, L);                          // thread t owns levels [lcut[t], lcut[t+1])
    lcut[0] = 0;
    for (int t = 1; t < NT; ++t) {
        const int32_t want = (int32_t)((int64_t)nV * t / NT);
        lcut[t] = (int)(std::upper_bound(g->level_off, g->level_off + L + 1, want) - g->level_off) - 1;
        lcut[t] = std::min(std::max(lcut[t], lcut[t - 1]), L);
    }
    std::vector<std::string> terr(NT);
    std::vector<int> trc(NT, DG_OK);
    auto run_threads = [&](auto &&fn) {                        // fn(t): returns nothing; errors through tfail
        std::vector<std::thread> th;
        for (int t = 1; t < NT; ++t) th.emplace_back([&, t] { fn(t); });
        fn(0);
        for (auto &x : th) x.join();
    };
    auto tfail = [&](int t, int rc, const char *fmt, auto... a) {
        if (trc[t] != DG_OK) return;
        char buf[256];
        snprintf(buf, sizeof buf, fmt, a...);
        terr[t] = buf; trc[t] = rc;
    };
    auto first_error = [&]() -> int {
        for (int t = 0; t < NT; ++t) if (trc[t] != DG_OK) { set_error("%s", terr[t].c_str()); return trc[t]; }
        return DG_OK;
    };
    for (int l = 0; l < L; ++l)
        if (g->level_off[l + 1] <= g->level_off[l]) { set_error("level %d is empty", l); return DG_ERR_ARG; }
    std::vector<int32_t> level_of(nV);
    std::vector<int> tmax_k(NT, 0);
    run_threads([&](int t) {
        for (int l = lcut[t]; l < lcut[t + 1]; ++l) {
            tmax_k[t] = std::max(tmax_k[t], g->level_off[l + 1] - g->level_off[l]);
            for (int v = g->level_off[l]; v < g->level_off[l + 1]; ++v) level_of[v] = l;
        }
    });
    int max_k = 0;
    for (int t = 0; t < NT; ++t) max_k = std::max(max_k, tmax_k[t]);
    if (max_k > MAX_K) { set_error("level width %d exceeds the supported %d", max_k, MAX_K); return DG_ERR_UNSUPPORTED; }
    if (g->out_off[0] != 0) { set_error("out_off must start at 0"); return DG_ERR_ARG; }
    for (int t = 1; t <= NT; ++t) {                            // monotone at the range seams (inside: checked by the owner)
        const int v = lcut[t] < L ? g->level_off[lcut[t]] : nV;
        if (g->out_off[v] < 0) { set_error("out_off negative at %d", v); return DG_ERR_ARG; }
    }
    const int64_t E = g->out_off[nV];
    if (E < 0 || E >= (int64_t)1 << 31) { set_error("unsupported number of edges (%lld)", (long long)E); return DG_ERR_UNSUPPORTED; }
    // in-CSR: in-edges of every vertex sorted by (source position asc, adjacency order asc)
    std::vector<uint32_t> in_off((size_t)nV + 1, 0), in_edge((size_t)E);
    std::vector<int32_t> in_dst((size_t)E);
    run_threads([&](int t) {
        const int va = g->level_off[lcut[t]], vb = lcut[t + 1] < L ? g->level_off[lcut[t + 1]] : nV;
        for (int u = va; u < vb; ++u) {
            if (g->out_off[u + 1] < g->out_off[u] || g->out_off[u + 1] > E) { tfail(t, DG_ERR_ARG, "out_off not monotone at %d", u); return; }
            for (int64_t e = g->out_off[u]; e < g->out_off[u + 1]; ++e) {
                const int v = g->out_dst[e];
                if (v < 0 || v >= nV || level_of[v] != level_of[u] + 1) { tfail(t, DG_ERR_ARG, "edge %d->%d does not go to the next level", u, v); return; }
                if (g->out_w[e] > 1) { tfail(t, DG_ERR_ARG, "edge weight %d > 1", (int)g->out_w[e]); return; }
                in_off[v + 1]++;                               // v lies in a level this thread owns the sources of
            }
        }
    });
    if (int rc = first_error()) return rc;
    for (int v = 0; v < nV; ++v) in_off[v + 1] += in_off[v];
    run_threads([&](int t) {
        const int va = g->level_off[lcut[t]], vb = lcut[t + 1] < L ? g->level_off[lcut[t + 1]] : nV;
        // destinations of this range: the vertices of levels [lcut[t] + 1, lcut[t+1] + 1)
        const int da = lcut[t] + 1 <= L ? g->level_off[std::min(lcut[t] + 1, L)] : nV;
        const int db = lcut[t + 1] + 1 <= L ? g->level_off[std::min(lcut[t + 1] + 1, L)] : nV;
        std::vector<uint32_t> fill(in_off.begin() + da, in_off.begin() + db);
        for (int u = va; u < vb; ++u) {
            const uint32_t pos = (uint32_t)(u - g->level_off[level_of[u]]);
            for (int64_t e = g->out_off[u]; e < g->out_off[u + 1]; ++e) {
                const int v = g->out_dst[e];
                const uint32_t slot = fill[v - da]++;
                in_edge[slot] = pos | ((uint32_t)g->out_w[e] << 31);
                in_dst[slot] = v;
            }
        }
        for (int v = da; v < db; ++v)     // parallel edges must carry equal weights (always true for graphs built by
            for (uint32_t e = in_off[v] + 1; e < in_off[v + 1]; ++e)   // Approximator::solve)
                if ((in_edge[e] & 0x7FFFFFFFu) == (in_edge[e - 1] & 0x7FFFFFFFu) && in_edge[e] != in_edge[e - 1]) {
                    tfail(t, DG_ERR_UNSUPPORTED, "parallel edges with different weights into vertex %d: tie order would be schedule dependent", v);
                    return;
                }
    });
    if (int rc = first_error()) return rc;
    lap("validate + in-CSR");
    // colour lists must be sorted (the merges rely on it) and fit the uint16 delta
    if (g->hom_off[0] != 0 || g->het_off[0] != 0) { set_error("colour offsets must start at 0"); return DG_ERR_ARG; }
    std::vector<int64_t> tmax_list(NT, 0);
    std::vector<uint8_t> has_col(L, 0);
    run_threads([&](int t) {
        const int va = g->level_off[lcut[t]], vb = lcut[t + 1] < L ? g->level_off[lcut[t + 1]] : nV;
        for (int v = va; v < vb; ++v) {
            for (int pass = 0; pass < 2; ++pass) {
                const int64_t *off = pass ? g->het_off : g->hom_off;
                const int32_t *colv = pass ? g->het_col : g->hom_col;
                if (off[v + 1] < off[v]) { tfail(t, DG_ERR_ARG, "colour offsets not monotone at %d", v); return; }
                tmax_list[t] = std::max(tmax_list[t], off[v + 1] - off[v]);
                if (off[v + 1] > off[v]) has_col[level_of[v]] = 1;
                for (int64_t q = off[v] + 1; q < off[v + 1]; ++q)
                    if (colv[q] <= colv[q - 1]) { tfail(t, DG_ERR_ARG, "colour list of vertex %d is not sorted-unique", v); return; }
            }
        }
    });
    if (int rc = first_error()) return rc;
    int64_t max_list = 0;
    for (int t = 0; t < NT; ++t) max_list = std::max(max_list, tmax_list[t]);
    if (max_list * 4 > 65535) { set_error("colour lists too long for uint16 score deltas (%lld)", (long long)max_list); return DG_ERR_UNSUPPORTED; }

    lap("colour checks");
    // level descriptors: every thread builds the groups / dead columns / slot blocks of its levels into private
    // vectors with range-local offsets, a serial prefix over the ranges turns them into global ones
    S.descs.assign(L, LevelDesc{});
    S.cells = S.relaxations = S.edge_pairs = S.colour_entries = 0;
    S.total_units = 0; S.max_level_units = 0; S.max_level_cells = S.RP; S.delta_entries = DELTA_PAD;
    S.level_units.assign(L, 0);
    S.level_dmax.assign(L, 0);
    std::vector<uint32_t> rowrec((size_t)nV * 4, 0);
    struct Part {
        std::vector<int32_t> dtrans, dead_cols, heavy;
        std::vector<uint32_t> grp_begin, slots;
        std::vector<int64_t> dblk_first;
        int64_t cells = 0, units = 0, max_level_cells = 0, max_level_units = 0, delta_entries = 0, nblk = 0;
        uint64_t edge_pairs = 0, colour_entries = 0;
    };
    std::vector<Part> part(NT);
    run_threads([&](int t) {
        Part &P = part[t];
        const int va = g->level_off[lcut[t]], vb = lcut[t + 1] < L ? g->level_off[lcut[t + 1]] : nV;
        for (int v = va; v < vb; ++v) {
            const uint32_t e0 = in_off[v], dv = in_off[v + 1] - e0;
            rowrec[4 * (size_t)v] = e0; rowrec[4 * (size_t)v + 1] = dv;
            // in-edges 0 and 1 ride along; bit 16 of each says "the source vertex has exactly one in-edge itself", which
            // lets the chain walk skip the back-pointer load at the next level when row and column are both forced
            const int a0v = g->level_off[std::max(0, level_of[v] - 1)];
            for (uint32_t q = 0; q < 2; ++q) {
                uint32_t word = 0;
                if (dv > q) {
                    word = in_edge[e0 + q];
                    const int src = a0v + (int)(word & 0x7FFFu);
                    if (in_off[src + 1] - in_off[src] == 1) word |= 1u << 16;
                }
                rowrec[4 * (size_t)v + 2 + q] = word;
            }
        }
        auto &grp_begin = P.grp_begin; auto &dead_cols = P.dead_cols; auto &slots = P.slots;
        for (int l = std::max(1, lcut[t]); l < lcut[t + 1]; ++l) {
            LevelDesc &d = S.descs[l];
            d.a0 = g->level_off[l - 1]; d.k = g->level_off[l] - d.a0;
            d.b0 = g->level_off[l]; d.k2 = g->level_off[l + 1] - d.b0;
            d.in_base = in_off[d.b0];
            d.T = (int32_t)(in_off[d.b0 + d.k2] - d.in_base);
            // column groups: greedy runs of whole columns with <= 64 in-edges; a column with more gets its own group
            d.grp_first = (int32_t)grp_begin.size();
            d.dead_first = (int32_t)dead_cols.size();
            d.heavy_first = (int32_t)P.heavy.size();
            for (int c = 0; c < d.k2; ++c) if (in_off[d.b0 + c + 1] - in_off[d.b0 + c] > (uint32_t)COOP_MIN) P.heavy.push_back(c);
            d.n_heavy = (int32_t)P.heavy.size() - d.heavy_first;
            {
                uint32_t cur_size = 0;
                for (int c = 0; c < d.k2; ++c) {
                    const uint32_t e0 = in_off[d.b0 + c], dv = in_off[d.b0 + c + 1] - e0;
                    if (dv == 0) { dead_cols.push_back(c); continue; }
                    if (cur_size == 0 || cur_size + dv > 64 || dv > 64) { grp_begin.push_back(e0); cur_size = 0; }
                    cur_size += dv;
                    if (dv > 64) cur_size = 65;                      // force a new group after a giant column
                }
            }
            d.ngroups = (int32_t)grp_begin.size() - d.grp_first;
            grp_begin.push_back(d.in_base + (uint32_t)d.T);          // sentinel: end of the level's in-edges
            // 64-wide slot table of the fast kernel: one block per group; a giant column (in-degree > 64) takes
            // ceil(dv/64) consecutive blocks (first one tagged 15, the rest 14) and counts as that many "groups"
            d.slot_first = (int64_t)(slots.size() / 2);
            d.fast_ok = (d.T < (1 << 20)) ? 1 : 0;                   // the slot word keeps 20 bits of in-edge index
            uint32_t max_indeg = 0;
            for (int c = 0; c < d.k2; ++c) max_indeg = std::max(max_indeg, in_off[d.b0 + c + 1] - in_off[d.b0 + c]);
            S.level_dmax[l] = (int32_t)max_indeg;
            d.bp_wide = max_indeg > (uint32_t)BP_MAX_RANK ? 1 : 0;  // ranks do not fit 8 bits: wide words, generic kernel
            if (d.bp_wide) d.fast_ok = 0;
            int32_t n_blocks = 0;
            for (int gi = 0; gi < d.ngroups; ++gi) {
                const uint32_t gb0 = grp_begin[d.grp_first + gi], ge0 = grp_begin[d.grp_first + gi + 1];
                const bool giant = ge0 - gb0 > 64;
                if (giant && d.fast_ok) d.fast_ok = 2;               // the general sweep variant
                uint32_t maxdv = 1;
                if (!giant)
                    for (uint32_t e = gb0; e < ge0;) {
                        const int cc = in_dst[e] - d.b0;
                        const uint32_t dv = in_off[d.b0 + cc + 1] - in_off[d.b0 + cc];
                        maxdv = std::max(maxdv, dv);
                        e += dv;
                    }
                uint32_t steps = 0;
                while ((1u << steps) < std::min(maxdv, 64u)) ++steps;
                const uint32_t nb = giant ? (ge0 - gb0 + 63) / 64 : 1;
                const size_t s0 = slots.size();
                slots.resize(s0 + (size_t)nb * 128);
                uint32_t *sp = slots.data() + s0;
                for (uint32_t bq = 0; bq < nb; ++bq) {
                    const uint32_t tag = giant ? (bq == 0 ? 15u : 14u) : steps;
                    for (uint32_t q = 0; q < 64; ++q, sp += 2) {
                        const uint32_t e = gb0 + bq * 64 + q;
                        if (e < ge0) {
                            const uint32_t pv = in_edge[e];
                            sp[0] = (pv & 0x7FFFu) | ((pv >> 31) << 15) | ((uint32_t)(in_dst[e] - d.b0) << 16);
                            sp[1] = (e - d.in_base) | ((e - in_off[in_dst[e]]) << 20) | (tag << 28);   // in-edge, its rank in the column, tag
                        } else {
                            sp[0] = 0xFFFFFFFFu;
                            sp[1] = tag << 28;
                        }
                    }
                }
                n_blocks += (int32_t)nb;
            }
            if (n_blocks == 0) {                                     // level without in-edges: one all-padding block
                for (int q = 0; q < 64; ++q) { slots.push_back(0xFFFFFFFFu); slots.push_back(0); }
                n_blocks = 1;
            }
            d.nblocks = n_blocks;
            d.ndead = (int32_t)dead_cols.size() - d.dead_first;
            if (d.ngroups == 0) { d.ngroups = 1; grp_begin.push_back(d.in_base + (uint32_t)d.T); }   // level without in-edges: one empty group
            const int64_t ncell = (int64_t)d.k2 * d.k2 * S.RP;
            const int64_t units = ((d.bp_wide ? 2 * ncell : ncell) + 1) & ~(int64_t)1;
            S.level_units[l] = units;
            d.bp_off = P.units;                                      // range-local for now
            P.units += units;
            P.cells += ncell;
            P.max_level_cells = std::max(P.max_level_cells, ncell);
            P.max_level_units = std::max(P.max_level_units, units);
            P.edge_pairs += (uint64_t)d.T * (uint64_t)d.T;
            if (has_col[l - 1] || has_col[l]) {
                d.delta_off = P.delta_entries;                       // range-local for now
                P.delta_entries += (int64_t)d.T * d.T;
                P.dtrans.push_back(l);
                P.dblk_first.push_back(P.nblk);
                P.nblk += ((int64_t)d.T * d.T + DELTA_PER_BLOCK - 1) / DELTA_PER_BLOCK;
                int64_t f = 0;   // sum over in-edges of |col(src)| + |col(dst)|
                for (uint32_t e = d.in_base; e < d.in_base + (uint32_t)d.T; ++e) {
                    const int sv = d.a0 + (int)(in_edge[e] & 0x7FFFFFFFu), tv = in_dst[e];
                    f += (g->hom_off[sv + 1] - g->hom_off[sv]) + (g->het_off[sv + 1] - g->het_off[sv]) +
                         (g->hom_off[tv + 1] - g->hom_off[tv]) + (g->het_off[tv + 1] - g->het_off[tv]);
                }
                P.colour_entries += (uint64_t)(2 * (int64_t)d.T * f);
            } else {
                d.delta_off = -1;
            }
        }
    });
    // serial prefix over the ranges, then every range shifts its levels and copies its vectors into place
    std::vector<int64_t> b_grp(NT + 1, 0), b_dead(NT + 1, 0), b_slot(NT + 1, 0), b_cells(NT + 1, 0), b_units(NT + 1, 0), b_delta(NT + 1, DELTA_PAD), b_blk(NT + 1, 0), b_dt(NT + 1, 0), b_heavy(NT + 1, 0);
    for (int t = 0; t < NT; ++t) {
        const Part &P = part[t];
        b_grp[t + 1] = b_grp[t] + (int64_t)P.grp_begin.size();
        b_dead[t + 1] = b_dead[t] + (int64_t)P.dead_cols.size();
        b_slot[t + 1] = b_slot[t] + (int64_t)P.slots.size() / 2;
        b_cells[t + 1] = b_cells[t] + P.cells;
        b_units[t + 1] = b_units[t] + P.units;
        b_delta[t + 1] = b_delta[t] + P.delta_entries;
        b_blk[t + 1] = b_blk[t] + P.nblk;
        b_dt[t + 1] = b_dt[t] + (int64_t)P.dtrans.size();
        b_heavy[t + 1] = b_heavy[t] + (int64_t)P.heavy.size();
        S.max_level_cells = std::max(S.max_level_cells, P.max_level_cells);
        S.max_level_units = std::max(S.max_level_units, P.max_level_units);
        S.edge_pairs += P.edge_pairs;
        S.colour_entries += P.colour_entries;
    }
    S.total_units = b_units[NT];
    S.cells = (uint64_t)b_cells[NT];
    S.delta_entries = b_delta[NT];
    const int64_t nblk = b_blk[NT];
    if (b_grp[NT] >= (int64_t)1 << 31 || b_dead[NT] >= (int64_t)1 << 31) { set_error("group tables too large"); return DG_ERR_UNSUPPORTED; }
    std::vector<int32_t> dtrans((size_t)b_dt[NT]), dead_cols((size_t)b_dead[NT]), heavy_rows((size_t)b_heavy[NT] + 1);
    std::vector<uint32_t> grp_begin((size_t)b_grp[NT]);
    std::vector<uint32_t> slots((size_t)b_slot[NT] * 2);         // 2 words per slot, 64 slots per block
    std::vector<int64_t> dblk_first((size_t)b_dt[NT]);
    run_threads([&](int t) {
        Part &P = part[t];
        for (int l = std::max(1, lcut[t]); l < lcut[t + 1]; ++l) {
            LevelDesc &d = S.descs[l];
            d.grp_first += (int32_t)b_grp[t];
            d.dead_first += (int32_t)b_dead[t];
            d.heavy_first += (int32_t)b_heavy[t];
            d.slot_first += b_slot[t];
            d.bp_off += b_units[t];
            if (d.delta_off >= 0) d.delta_off += b_delta[t];
        }
        std::copy(P.grp_begin.begin(), P.grp_begin.end(), grp_begin.begin() + b_grp[t]);
        std::copy(P.dead_cols.begin(), P.dead_cols.end(), dead_cols.begin() + b_dead[t]);
        std::copy(P.heavy.begin(), P.heavy.end(), heavy_rows.begin() + b_heavy[t]);
        std::copy(P.slots.begin(), P.slots.end(), slots.begin() + 2 * b_slot[t]);
        std::copy(P.dtrans.begin(), P.dtrans.end(), dtrans.begin() + b_dt[t]);
        for (size_t q = 0; q < P.dblk_first.size(); ++q) dblk_first[(size_t)b_dt[t] + q] = P.dblk_first[q] + b_blk[t];
        Part().dtrans.swap(P.dtrans); std::vector<uint32_t>().swap(P.slots);
    });
    S.relaxations = S.edge_pairs * (uint64_t)S.RP;
    S.n_delta_blocks = nblk;
    if (nblk >= (int64_t)1 << 31) { set_error("delta grid too large"); return DG_ERR_UNSUPPORTED; }

    lap("descs + groups + slots");
    // memory budget, lattice chunking / segmentation
    // score-delta windows (see DpState::dwin_t)
    S.dtrans_host = dtrans;
    S.dblk_first_host = dblk_first;
    S.dblk_first_host.push_back(nblk);
    S.dwin_t.assign(1, 0);
    S.level_win.assign(L, -1);
    {
        int64_t acc = 0, max_win = 0;
        for (size_t t = 0; t < dtrans.size(); ++t) {
            const int64_t n = (int64_t)S.descs[dtrans[t]].T * S.descs[dtrans[t]].T;
            if (acc > 0 && acc + n > S.delta_cap_entries) { S.dwin_t.push_back((int32_t)t); max_win = std::max(max_win, acc); acc = 0; }
            acc += n;
            S.level_win[dtrans[t]] = (int32_t)S.dwin_t.size() - 1;
        }
        max_win = std::max(max_win, acc);
        S.dwin_t.push_back((int32_t)dtrans.size());
        S.delta_buf_entries = DELTA_PAD + max_win;
    }
    const size_t st_bytes = (size_t)S.max_level_cells * 4 * 2, dl_bytes = (size_t)S.delta_buf_entries * 2;
    size_t free_b = 0, total_b = 0;
    DG_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t pool_bytes;
    { std::unique_lock<std::mutex> lk(S.pool.mu); pool_bytes = S.pool.chunks.size() * S.pool.chunk_units * 2; }
    const size_t have = free_b + pool_bytes + S.d_bp.bytes + S.d_delta.bytes + S.d_val[0].bytes + S.d_val[1].bytes + S.d_ckpt.bytes;
    const size_t fixed = st_bytes + dl_bytes + 64 * (size_t)nV + ((size_t)2 << 30);     // state, delta, tables, slack
    if (fixed > have) {
        set_error("graph needs %.1f GB of HBM for state/delta/tables but only %.1f GB is free", fixed / 1e9, have / 1e9);
        return DG_ERR_OOM;
    }
    // Levels are packed into equal chunks (a level never straddles two).  If all chunks fit they stay resident;
    // otherwise a segment = as many consecutive chunks as fit (pool chunks are reused by every segment) and the run
    // goes checkpoint + recompute.  segment_cells (tests) caps the chunk size and forces one chunk per segment.
    size_t chunk_units = S.chunk_units_cfg;
    if (S.segment_cells > 0) chunk_units = std::min(chunk_units, ((size_t)S.segment_cells + 1) & ~(size_t)1);
    if ((size_t)S.max_level_units > chunk_units) chunk_units = (size_t)S.max_level_units;
    S.chunk_begin.assign(1, 1);
    {
        size_t acc = 0;
        for (int l = 1; l < L; ++l) {
            const size_t nu = (size_t)S.level_units[l];
            if (acc > 0 && acc + nu > chunk_units) { S.chunk_begin.push_back(l); acc = 0; }
            acc += nu;
        }
        S.chunk_begin.push_back(L);
    }
    const size_t n_chunks = S.chunk_begin.size() - 1;
    const bool tiny = n_chunks == 1 && (size_t)S.total_units < S.chunk_units_cfg / 8 && S.segment_cells == 0;   // one exact allocation
    const size_t chunk_bytes = chunk_units * 2;
    const size_t resident_bytes = tiny ? (size_t)S.total_units * 2 : n_chunks * chunk_bytes;
    const bool segmented = (resident_bytes + fixed > have || S.segment_cells > 0) && n_chunks > 1;
    if (!segmented && resident_bytes + fixed > have) {
        set_error("back-pointer lattice of %.1f GB (one level alone needs %.1f GB) does not fit the %.1f GB of free HBM", resident_bytes / 1e9,
                  S.max_level_units * 2 / 1e9, (have - fixed) / 1e9);
        return DG_ERR_OOM;
    }
    // chunks per segment: the most that fit beside the checkpoints (state in front of every segment)
    size_t group = n_chunks;
    S.seg_begin.assign(1, 1);
    S.ckpt_off.assign(1, 0);
    int64_t ckpt_cells = 0;
    if (segmented) {
        group = S.segment_cells > 0 ? 1 : std::max<size_t>(1, std::min(n_chunks - 1, (have - fixed) / chunk_bytes));
        for (;; --group) {
            S.seg_begin.assign(1, 1);
            S.ckpt_off.assign(1, 0);
            ckpt_cells = 0;
            for (size_t cb = group; cb < n_chunks; cb += group) {
                const int l = S.chunk_begin[cb];
                S.seg_begin.push_back(l);
                S.ckpt_off.push_back(ckpt_cells);
                ckpt_cells += (int64_t)S.descs[l].k * S.descs[l].k * S.RP;            // state of level l-1
            }
            if (group * chunk_bytes + (size_t)ckpt_cells * 4 + fixed <= have || group == 1) break;
        }
        if (group * chunk_bytes + (size_t)ckpt_cells * 4 + fixed > have) {
            set_error("segmented lattice needs %.1f GB (+%.1f GB checkpoints) but only %.1f GB of HBM is free", group * chunk_bytes / 1e9,
                      ckpt_cells * 4 / 1e9, (have - fixed) / 1e9);
            return DG_ERR_OOM;
        }
        if (dbg)
            fprintf(stderr, "[dipgenie_hip] lattice %.1f GB does not fit: %zu segments of <= %zu chunks of %.1f GB, checkpoints %.2f GB\n",
                    S.total_units * 2 / 1e9, S.seg_begin.size(), group, chunk_bytes / 1e9, ckpt_cells * 4 / 1e9);
    }
    S.seg_begin.push_back(L);
    S.seg_chunks = (int)group;
    const size_t bp_bytes = tiny ? resident_bytes : group * chunk_bytes, ck_bytes = (size_t)ckpt_cells * 4;
    S.d_bp.release();
    if (tiny) {
        pool_clear(S);
        if (int rc = S.d_bp.ensure((size_t)S.total_units * 2)) return rc;
    } else {
        if (chunk_units != S.pool.chunk_units) { pool_clear(S); S.pool.chunk_units = chunk_units; pool_bytes = 0; }
        // surplus chunks of an over-estimated reservation stay unless the other buffers need their room
        const size_t mapped = pool_bytes / chunk_bytes;
        if (mapped > group && free_b < fixed + ck_bytes) pool_trim(S, group);
        pool_request(S, c->device, group);                      // returns at once; dp_run waits for the chunks
        if (dbg) fprintf(stderr, "[dipgenie_hip] lattice: %zu chunks of %.1f GB (%zu resident at a time), %zu mapped so far\n", n_chunks, chunk_bytes / 1e9, group, mapped);
    }
    PoolPause pause(S);                                         // until the allocations below are done
    lap("plan lattice");
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
    if (int rc = upload(S.d_grp, grp_begin.data(), 4 * grp_begin.size(), s)) return rc;
    if (int rc = upload(S.d_dead, dead_cols.data(), 4 * dead_cols.size(), s)) return rc;
    if (int rc = upload(S.d_heavy, heavy_rows.data(), 4 * heavy_rows.size(), s)) return rc;
    if (int rc = upload(S.d_rowrec, rowrec.data(), 4 * rowrec.size(), s)) return rc;
    if (int rc = upload(S.d_slots, slots.data(), 4 * slots.size(), s)) return rc;
    if (int rc = S.d_eflag.ensure(in_dst.size() + 16)) return rc;
    if (int rc = S.d_eself.ensure(2 * in_dst.size() + 16)) return rc;
    {
        ColourCsr colf{S.d_hom_off.as<int64_t>(), S.d_het_off.as<int64_t>(), S.d_hom_col.as<int32_t>(), S.d_het_col.as<int32_t>()};
        hipLaunchKernelGGL(dp_edge_flags_kernel, dim3((unsigned)std::max(L - 1, 1)), dim3(64), 0, s, S.d_descs.as<LevelDesc>(), L, S.d_in_edge.as<uint32_t>(),
                           S.d_in_dst.as<int32_t>(), colf, S.d_eflag.as<uint8_t>(), S.d_eself.as<uint16_t>());
    }
    lap("table uploads");
    if (int rc = S.d_delta.ensure(dl_bytes)) return rc;
    DG_HIP(hipMemsetAsync(S.d_delta.p, 0, 2 * DELTA_PAD, s));
    if (int rc = S.d_ckpt.ensure(ck_bytes)) return rc;
    if (int rc = S.d_chain.ensure(sizeof(ChainState))) return rc;
    S.pad_front = 2 * (int64_t)max_k;
    const size_t pad_bytes = 4 * (size_t)(S.pad_front + 33 * (int64_t)max_k);
    if (int rc = S.d_val[0].ensure(st_bytes / 2 + pad_bytes)) return rc;
    if (int rc = S.d_val[1].ensure(st_bytes / 2 + pad_bytes)) return rc;
    DG_HIP(hipMemsetAsync(S.d_val[0].p, 0, S.d_val[0].bytes, s));
    DG_HIP(hipMemsetAsync(S.d_val[1].p, 0, S.d_val[1].bytes, s));
    if (int rc = S.d_digest.ensure(8 * (size_t)L)) return rc;
    if (int rc = S.d_trace.ensure(sizeof(TraceOut))) return rc;
    if (int rc = S.d_ctrl.ensure(sizeof(TeamCtl) * TEAM_CTL_SLOTS)) return rc;
    S.state_alloc_bytes = st_bytes / 2 + pad_bytes;
    S.all_fast = true;
    for (int l = 1; l < L; ++l) if (!S.descs[l].fast_ok || (int64_t)S.descs[l].k2 * S.descs[l].nblocks * 5 >= ((int64_t)1 << 31)) S.all_fast = false;
    S.cap = 2 * (R + 8);                               // edge records of both paths
    if (int rc = S.d_edges.ensure(4 * 4 * (size_t)S.cap)) return rc;
    if (int rc = S.d_path.ensure(4 * (size_t)L)) return rc;
    DG_HIP(hipStreamSynchronize(s));      // host staging vectors die here
    lap("allocs + sync");
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
    const int rc_sel = S.RP <= 8 ? 8 : (S.RP <= 19 ? 19 : 33);
    const int nchunk = (S.RP + rc_sel - 1) / rc_sel;
    const bool small_state = S.state_alloc_bytes < ((size_t)1 << 31);   // 32-bit buffer offsets
    SweepArgs A;
    A.descs = descs; A.in_off = S.d_in_off.as<uint32_t>(); A.in_edge = S.d_in_edge.as<uint32_t>();
    A.grp_begin = S.d_grp.as<uint32_t>(); A.in_dst = S.d_in_dst.as<int32_t>(); A.dead_cols = S.d_dead.as<int32_t>();
    A.delta = A.delta_zero = S.d_delta.as<uint16_t>();
    A.buf0 = S.d_val[0].as<int32_t>() + S.pad_front; A.buf1 = S.d_val[1].as<int32_t>() + S.pad_front;
    A.bp = nullptr; A.digest = S.d_digest.as<unsigned long long>(); A.RP = S.RP;
    FastArgs F;
    F.rowrec = S.d_rowrec.as<uint4>(); F.slots = S.d_slots.as<uint2>(); F.in_edge = A.in_edge; F.dead_cols = A.dead_cols;
    F.delta = F.delta_zero = A.delta; F.bp = nullptr; F.digest = A.digest; F.RP = S.RP;
    F.base0 = S.d_val[0].as<int32_t>(); F.base1 = S.d_val[1].as<int32_t>();
    F.pad_bytes = (int)(4 * S.pad_front);
    F.buf_bytes = (uint32_t)std::min<size_t>(std::min(S.d_val[0].bytes, S.d_val[1].bytes), 0x7FFFFFFFu);
    int64_t n_launch = 0;
    double host_enqueue_s = 0;                          // host time spent issuing the sweep's launches (DG_DEBUG)
    uint32_t team_err = 0;
    bool team_used = false, team_failed = false;
    int n_team_launch = 0;

    // Sweeps destination levels [lb, le).  bp_biased = lattice pointer minus the offset of level lb's first cell
    // (so the kernels keep using the global LevelDesc::bp_off), or nullptr for a value-only pass.
    // (re)computes the score deltas of window w into d_delta and points the sweep arguments at it
    const int n_win = (int)S.dwin_t.size() - 1;
    auto load_window = [&](int w) {
        const int t0 = S.dwin_t[w], t1 = S.dwin_t[w + 1];
        if (t1 > t0) {
            const int64_t b0 = S.dblk_first_host[t0], b1 = S.dblk_first_host[t1];
            const int64_t base_off = S.descs[S.dtrans_host[t0]].delta_off;          // global entry offset of the window's first matrix
            uint16_t *biased = S.d_delta.as<uint16_t>() + DELTA_PAD - base_off;
            hipLaunchKernelGGL(dp_delta_kernel, dim3((unsigned)(b1 - b0)), dim3(256), 0, s, descs, S.d_dtrans.as<int32_t>(),
                               S.d_dblk_first.as<int64_t>(), (int)S.dtrans_host.size(), S.d_in_edge.as<uint32_t>(), S.d_in_dst.as<int32_t>(), col,
                               biased, b0, S.d_eflag.as<uint8_t>(), S.d_eself.as<uint16_t>());
            A.delta = F.delta = biased;
        }
        S.cur_win = w;
    };
    auto warm_tables = [&](int q0, int q1) {             // graph tables of destination levels [q0, q1) -> Infinity Cache
        const LevelDesc &da = S.descs[q0], &db = S.descs[q1 - 1];
        WarmRanges W{};
        auto put = [&](int q, const void *base, int64_t b0, int64_t b1) {       // byte range [b0, b1) behind base
            const uintptr_t a = ((uintptr_t)base + (uintptr_t)b0) & ~(uintptr_t)15, e = ((uintptr_t)base + (uintptr_t)b1) & ~(uintptr_t)15;
            W.p[q] = (const char *)a; W.n16[q] = e > a ? (long long)((e - a) >> 4) : 0;
        };
        put(0, F.rowrec, 16 * (int64_t)da.b0, 16 * ((int64_t)db.b0 + db.k2));
        put(1, F.slots, 8 * da.slot_first, 8 * (db.slot_first + (int64_t)db.nblocks * 64));
        put(2, F.in_edge, 4 * (int64_t)da.in_base, 4 * ((int64_t)db.in_base + db.T));
        int64_t d0 = -1, d1 = -1;
        for (int q = q0; q < q1; ++q) {
            const LevelDesc &dq = S.descs[q];
            if (dq.delta_off < 0 || S.level_win[q] != S.cur_win) continue;
            const int64_t e = dq.delta_off + (int64_t)dq.T * dq.T;
            d0 = d0 < 0 ? dq.delta_off : std::min(d0, dq.delta_off);
            d1 = std::max(d1, e);
        }
        if (d0 >= 0) put(3, F.delta, 2 * d0, 2 * d1);
        long long tot = 0;
        for (int q = 0; q < 4; ++q) tot += W.n16[q];
        if (tot <= 0) return;
        const unsigned grid = (unsigned)std::min<long long>((tot + 255) / 256, 2048);
        hipLaunchKernelGGL(dp_warm_tables_kernel, dim3(grid), dim3(256), 0, s, W);
    };
    auto sweep_range = [&](int lb, int le, uint16_t *bp_biased) -> int {
        A.bp = bp_biased; F.bp = bp_biased;
        // runs of narrow levels may go to the one-XCD team kernel (one launch per run, optional); every other level
        // gets one whole-chip launch
        const bool team_ok = S.use_team && !S.want_digest && small_state && !team_failed && n_win == 1;
        S.schedule.clear();
        for (int l = lb; l < le;) {
            auto narrow = [&](int q) { const LevelDesc &d = S.descs[q]; return team_ok && d.fast_ok == 1 && (int64_t)d.k2 * d.nblocks <= S.team_max_tasks; };
            int e = l;
            if (narrow(l)) { while (e < le && narrow(e)) ++e; }
            const bool is_team = e - l >= S.team_min_levels;
            if (!is_team) e = std::max(e, l + 1);
            if (!is_team && !S.schedule.empty() && !S.schedule.back().team) S.schedule.back().end = e;
            else S.schedule.push_back({l, e, is_team});
            l = e;
        }
        for (const auto &seg : S.schedule) {
            if (seg.team) {
                TeamCtl *ctl = S.d_ctrl.as<TeamCtl>() + (n_team_launch % TEAM_CTL_SLOTS);
                DG_HIP(hipMemsetAsync(ctl, 0, sizeof(TeamCtl), s));
                const dim3 grid((unsigned)S.team_grid);
#define DG_TEAM(RCV) hipLaunchKernelGGL((dp_team_kernel<RCV, false>), grid, dim3(512), 0, s, F, descs, seg.begin, seg.end, ctl)
                if (rc_sel == 19 && getenv("DG_TEAM_PROF")) hipLaunchKernelGGL((dp_team_kernel<19, true>), grid, dim3(512), 0, s, F, descs, seg.begin, seg.end, ctl);
                else if (rc_sel == 8) DG_TEAM(8); else if (rc_sel == 19) DG_TEAM(19); else DG_TEAM(33);
#undef DG_TEAM
                team_used = true;
                ++n_team_launch;
                ++n_launch;
                if (n_team_launch % TEAM_CTL_SLOTS == 0) {      // ctl slots are recycled: check the finished ones first
                    std::vector<TeamCtl> hc(TEAM_CTL_SLOTS);
                    DG_HIP(hipMemcpyAsync(hc.data(), S.d_ctrl.p, sizeof(TeamCtl) * TEAM_CTL_SLOTS, hipMemcpyDeviceToHost, s));
                    DG_HIP(hipStreamSynchronize(s));
                    for (auto &h : hc) if (h.error) team_err = h.error;
                    if (team_err) return DG_OK;
                }
                continue;
            }
            // Issuing a level costs the host 3-4.5 us (hipLaunchKernelGGL), the GPU 2-3 us on narrow levels: batches of levels are
            // captured once into hipGraphs and replayed on later passes over the same resident graph (option graph_batch).
            // Measured: MHC_4 (3.5 k cells per level) 390 -> 358 ms per sweep, also on the capturing pass; MHC-24 (265 k cells per
            // level, GPU-bound at 4.6 us) 651 -> 658 ms.  -1 picks 1,000-level batches for graphs below 32 k cells per level.
            const int64_t gb = S.graph_batch >= 0 ? S.graph_batch : ((int64_t)(S.cells / (uint64_t)std::max(S.L, 1)) < 32768 ? 1000 : 0);
            // A stream that cannot be captured (e.g. a caller-provided legacy stream) or a failed instantiation switches the
            // context back to plain launches for good; the batch at hand is then issued again, plainly.
            for (int l0 = seg.begin; l0 < seg.end;) {
                const bool use_graph = gb > 0 && n_win == 1 && S.sync_every == 0 && !S.graph_failed;
                const int l1 = use_graph ? (int)std::min<int64_t>((int64_t)l0 + gb, seg.end) : seg.end;
                hipGraphExec_t *slot = nullptr;
                bool capturing = false;
                if (use_graph) {
                    slot = &S.graphs[std::make_tuple(l0, l1, (const void *)bp_biased)];
                    if (*slot) { DG_HIP(hipGraphLaunch(*slot, s)); n_launch += l1 - l0; l0 = l1; continue; }
                    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
                    else { (void)hipGetLastError(); S.graph_failed = true; continue; }
                }
                const int64_t n_launch_before = n_launch;
                for (int l = l0; l < l1; ++l) {
                    LevelDesc &d = S.descs[l];
                    if (S.level_win[l] >= 0 && S.level_win[l] != S.cur_win) load_window(S.level_win[l]);
                    if (S.warm_ahead > 0 && (l - lb) % S.warm_ahead == 0) {
                        // tables of the batch after this one (and, at the start of a range, of this one too)
                        const int q0 = l == lb ? l : (int)std::min<int64_t>(l + S.warm_ahead, le), q1 = (int)std::min<int64_t>(l + 2 * S.warm_ahead, le);
                        if (q1 > q0) warm_tables(q0, q1);
                    }
                    // tiny levels end sooner with write-back stores (3.6 vs 4.2 us per level on MHC_4), big ones with
                    // non-temporal ones that keep the once-written lattice out of the L2
                    d.bp_nt = (int64_t)d.k2 * d.k2 * S.RP >= S.bp_nt_min_cells ? 1 : 0;
                    if (d.fast_ok && small_state && S.RP <= 65535 && S.use_fast) {
                        // A lone wave retires ~1 instruction per 4-8 cycles, so the RC-fold unrolled task is the level's
                        // critical path: while the chip has idle wave slots, give each wave fewer recombination counts.
                        const int64_t base = (int64_t)d.k2 * d.nblocks;
                        int rc = rc_sel;
                        bool coop = false;
                        if (S.adaptive_rc == 1) {                             // first rule: smallest RC whose waves fit a budget
                            static const int cand[5] = {1, 2, 4, 8, 16};
                            for (int q = 0; q < 5; ++q)
                                if (cand[q] < rc_sel && base * ((S.RP + cand[q] - 1) / cand[q]) <= S.chip_waves) { rc = cand[q]; break; }
                        } else if (S.adaptive_rc >= 2) {
                            // Cost model fitted on MHC-24 (R = 18) and the 100-walk chr22-style panel (R = 32):
                            //   T(RC) = max(1, W / cap) * (t0 + dmax * RC * tg) + W * tw,   W = tasks * ceil(RP / RC) waves.
                            // First factor: rounds of resident waves; second: a wave's dependent chain (the row with the
                            // largest in-degree walks dmax in-edges with RC gathers each); last: per-wave issue overhead.
                            // Cooperative variant (RC <= 4, lean levels): rows above COOP_MIN in-edges are walked by four waves, so
                            // the chain is a quarter (at least COOP_MIN) while four extra workgroup slots per heavy row are launched.
                            const int cand[11] = {1, 2, 3, 4, 5, 6, 8, 10, 11, 16, rc_sel};
                            const bool coop_ok = S.use_coop && S.adaptive_rc >= 3 && d.n_heavy > 0 && d.k2 + 4 * d.n_heavy <= 65535;
                            const double dmax = (double)std::max(1, S.level_dmax[l]);
                            double best = 1e300;
                            for (int pass = (coop_ok && S.use_coop == 2) ? 1 : 0; pass < (coop_ok ? 2 : 1); ++pass) {     // coop = 2 (tests): whenever possible
                                for (int q = 0; q < 11; ++q) {
                                    if (cand[q] > rc_sel || (q < 10 && cand[q] == rc_sel)) continue;
                                    if (S.adaptive_rc == 2 && (cand[q] == 3 || cand[q] == 5 || cand[q] == 6 || cand[q] == 10 || cand[q] == 11)) continue;   // 3: all sizes
                                    if (pass == 1 && cand[q] > 4) continue;
                                    const double rows = pass ? (double)d.k2 + 4.0 * d.n_heavy : (double)d.k2;
                                    const double chain = pass ? std::max((double)COOP_MIN, std::ceil(dmax / 4.0)) + 1.0 : dmax;
                                    const double W = rows * d.nblocks * ((S.RP + cand[q] - 1) / cand[q]);
                                    const double T = std::max(1.0, W / (double)S.rc_cap) * ((double)S.rc_t0_ns + (pass ? (double)S.coop_cost_ns : 0.0) + chain * cand[q] * (double)S.rc_tg_ps * 1e-3) +
                                                     W * (double)S.rc_tw_ps * 1e-3;
                                    if (T <= best) { best = T; rc = cand[q]; coop = pass == 1; }   // ties: the larger RC (fewer waves)
                                }
                            }
                        }
                        const int nch = (S.RP + rc - 1) / rc;
                        const int wpb = coop ? 4 : (int)S.waves_per_block;    // waves (= slot blocks) per workgroup
                        const dim3 grid((unsigned)((d.nblocks + wpb - 1) / wpb), (unsigned)nch, (unsigned)(d.k2 + (coop ? 4 * d.n_heavy : 0)));
                        const int32_t *hv = S.d_heavy.as<int32_t>();
    #define DG_FAST(RCV, DG) do { if (d.fast_ok == 2) hipLaunchKernelGGL((dp_sweep_fast_kernel<RCV, DG, true>), grid, dim3(64 * wpb), 0, s, F, d, l, hv); \
                                  else hipLaunchKernelGGL((dp_sweep_fast_kernel<RCV, DG, false>), grid, dim3(64 * wpb), 0, s, F, d, l, hv); } while (0)
    #define DG_COOP(RCV, DG) do { if (d.fast_ok == 2) hipLaunchKernelGGL((dp_sweep_fast_kernel<RCV, DG, true, true>), grid, dim3(256), 0, s, F, d, l, hv); \
                                  else hipLaunchKernelGGL((dp_sweep_fast_kernel<RCV, DG, false, true>), grid, dim3(256), 0, s, F, d, l, hv); } while (0)
    #define DG_FAST_RC(DG) do { if (coop) { switch (rc) { case 1: DG_COOP(1, DG); break; case 2: DG_COOP(2, DG); break; case 3: DG_COOP(3, DG); break; \
                                                        default: DG_COOP(4, DG); break; } break; } \
                                switch (rc) { case 1: DG_FAST(1, DG); break; case 2: DG_FAST(2, DG); break; case 3: DG_FAST(3, DG); break; \
                                            case 4: DG_FAST(4, DG); break; case 5: DG_FAST(5, DG); break; case 6: DG_FAST(6, DG); break; \
                                            case 8: DG_FAST(8, DG); break; case 10: DG_FAST(10, DG); break; case 11: DG_FAST(11, DG); break; \
                                            case 16: DG_FAST(16, DG); break; case 19: DG_FAST(19, DG); break; \
                                            default: DG_FAST(33, DG); break; } } while (0)
                        if (S.want_digest) DG_FAST_RC(true); else DG_FAST_RC(false);
    #undef DG_COOP
    #undef DG_FAST_RC
    #undef DG_FAST
                    } else {
                        const int64_t ntask = (int64_t)d.k2 * d.ngroups * nchunk;
                        const unsigned grid = (unsigned)std::min<int64_t>((ntask + 3) / 4, S.max_blocks);
    #define DG_SWEEP(RCV, DG) hipLaunchKernelGGL((dp_sweep_kernel<RCV, DG>), dim3(grid), dim3(256), 0, s, A, l)
                        if (S.want_digest) { if (rc_sel == 8) DG_SWEEP(8, true); else if (rc_sel == 19) DG_SWEEP(19, true); else DG_SWEEP(33, true); }
                        else { if (rc_sel == 8) DG_SWEEP(8, false); else if (rc_sel == 19) DG_SWEEP(19, false); else DG_SWEEP(33, false); }
    #undef DG_SWEEP
                    }
                    ++n_launch;
                    // profiling aid: rocprofv3 --pmc crashes when ~10^5 dispatches are queued without a drain
                    if (S.sync_every > 0 && n_launch % S.sync_every == 0) DG_HIP(hipStreamSynchronize(s));
                }
                if (capturing) {
                    hipGraph_t cg = nullptr;
                    const bool ok = hipStreamEndCapture(s, &cg) == hipSuccess && cg && hipGraphInstantiate(slot, cg, nullptr, nullptr, 0) == hipSuccess;
                    if (cg) (void)hipGraphDestroy(cg);
                    if (!ok) {                                          // nothing of this batch has run: issue it again without a graph
                        (void)hipGetLastError();
                        *slot = nullptr; S.graph_failed = true; n_launch = n_launch_before;
                        continue;
                    }
                    DG_HIP(hipGraphLaunch(*slot, s));
                }
                l0 = l1;
            }
        }
        return DG_OK;
    };
    auto state_ptr = [&](int level) { return S.d_val[level & 1].as<int32_t>() + S.pad_front; };
    auto level_cells = [&](int level) -> size_t {      // state size of a level (level 0: the source, k = 1)
        const int64_t k = level == 0 ? 1 : S.descs[level].k2;
        return (size_t)(k * k * S.RP);
    };
    const int n_seg = (int)S.seg_begin.size() - 1;
    auto warm_rows = [&](int lb, int le) {              // row records of destination levels [lb, le), at most ~200 MB worth
        const int64_t v0 = S.descs[lb].b0, v1 = (int64_t)S.descs[le - 1].b0 + S.descs[le - 1].k2;
        const int64_t n = std::min<int64_t>(v1 - v0, (int64_t)12 << 20);
        if (n > 0 && S.warm_rows) hipLaunchKernelGGL(dp_warm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S.d_rowrec.as<uint4>() + (v1 - n), n);
    };

    // Launches issued while the pool thread is still mapping chunks would each queue behind a multi-GB hipMalloc,
    // so there is nothing to overlap: wait for every chunk this run uses first.
    const int n_chunks_all = (int)S.chunk_begin.size() - 1;
    std::vector<uint16_t *> pool_base;
    if (!S.d_bp.p) {
        const double tw0 = wall_s();
        const size_t need = (size_t)std::min(S.seg_chunks, n_chunks_all);
        if (!pool_wait(S, need - 1)) { set_error("back-pointer lattice: hipMalloc of a %.1f GB chunk failed", S.pool.chunk_units * 2 / 1e9); return DG_ERR_OOM; }
        std::unique_lock<std::mutex> lk(S.pool.mu);
        for (size_t q = 0; q < need; ++q) pool_base.push_back((uint16_t *)S.pool.chunks[q]);
        if (getenv("DG_DEBUG")) fprintf(stderr, "[dipgenie_hip] run: waited %.3f s for lattice chunks\n", wall_s() - tw0);
    }
    // Sweeps the chunks [c0, c1) with back-pointers into pool chunks 0 .. c1-c0-1 (or the one exact buffer), then walks
    // them last first; from_sink = this is the walk that starts at the sink.
    auto sweep_and_walk = [&](int c0, int c1, bool from_sink, bool mark_forward_end) -> int {
        std::vector<uint16_t *> biased(c1 - c0);
        for (int ch = c0; ch < c1; ++ch) {
            const int lb = S.d_bp.p ? 1 : S.chunk_begin[ch], le = S.d_bp.p ? S.L : S.chunk_begin[ch + 1];
            uint16_t *base = S.d_bp.p ? S.d_bp.as<uint16_t>() : pool_base[ch - c0];
            biased[ch - c0] = base - S.descs[lb].bp_off;
            const double th0 = wall_s();
            if (int rc = sweep_range(lb, le, biased[ch - c0])) return rc;
            host_enqueue_s += wall_s() - th0;
        }
        if (mark_forward_end) DG_HIP(hipEventRecord(S.ev[2], s));
        for (int ch = c1 - 1; ch >= c0; --ch) {
            const int lb = S.d_bp.p ? 1 : S.chunk_begin[ch], le = S.d_bp.p ? S.L : S.chunk_begin[ch + 1];
            warm_rows(lb, le);
            const int32_t *fv = from_sink && ch == c1 - 1 ? state_ptr(S.L - 1) : (const int32_t *)nullptr;
            if (S.chain_spec)
                hipLaunchKernelGGL(dp_trace_chain_spec_kernel, dim3(1), dim3(64), 0, s, descs, le - 1, lb, S.RP, S.R, biased[ch - c0], fv, S.d_rowrec.as<uint4>(),
                                   S.d_in_edge.as<uint32_t>(), S.d_path.as<uint32_t>(), S.d_chain.as<ChainState>());
            else
                hipLaunchKernelGGL(dp_trace_chain_kernel, dim3(1), dim3(64), 0, s, descs, le - 1, lb, S.RP, S.R, biased[ch - c0], fv, S.d_rowrec.as<uint4>(),
                                   S.d_in_edge.as<uint32_t>(), S.d_path.as<uint32_t>(), S.d_chain.as<ChainState>());
        }
        return DG_OK;
    };
retry_forward:
    n_launch = 0; team_used = false; team_err = 0;
    DG_HIP(hipEventRecord(S.ev[0], s));
    S.cur_win = -1;
    if (n_win == 1 && S.n_delta_blocks > 0) load_window(0);     // everything fits: computed once, up front (delta_ms)
    DG_HIP(hipEventRecord(S.ev[1], s));
    if (S.want_digest) DG_HIP(hipMemsetAsync(S.d_digest.p, 0, 8 * (size_t)S.L, s));
    hipLaunchKernelGGL(dp_init_kernel, dim3(1), dim3(256 * ((S.RP + 255) / 256)), 0, s, state_ptr(0), S.RP);
    if (n_seg == 1) {
        // whole lattice resident: one sweep with back-pointers, then the chain walk chunk by chunk, last first
        if (int rc = sweep_and_walk(0, S.d_bp.p ? 1 : n_chunks_all, true, true)) return rc;
    } else {
        // pass 1: values only, keeping the state in front of every segment
        const bool dig = S.want_digest;
        for (int sg = 0; sg < n_seg; ++sg) {
            if (sg > 0)
                DG_HIP(hipMemcpyAsync(S.d_ckpt.as<int32_t>() + S.ckpt_off[sg], state_ptr(S.seg_begin[sg] - 1),
                                      4 * level_cells(S.seg_begin[sg] - 1), hipMemcpyDeviceToDevice, s));
            if (int rc = sweep_range(S.seg_begin[sg], S.seg_begin[sg + 1], nullptr)) return rc;
        }
        DG_HIP(hipEventRecord(S.ev[2], s));                 // (the re-sweeps below are booked under traceback_ms)
        // pass 2: last segment first -- restore its input state, re-sweep its chunks with back-pointers, walk them
        S.want_digest = 0;                                  // digests were accumulated in pass 1
        for (int sg = n_seg - 1; sg >= 0; --sg) {
            const int lb = S.seg_begin[sg];
            if (sg > 0)
                DG_HIP(hipMemcpyAsync(state_ptr(lb - 1), S.d_ckpt.as<int32_t>() + S.ckpt_off[sg], 4 * level_cells(lb - 1), hipMemcpyDeviceToDevice, s));
            else
                hipLaunchKernelGGL(dp_init_kernel, dim3(1), dim3(256 * ((S.RP + 255) / 256)), 0, s, state_ptr(0), S.RP);
            const int c0 = sg * S.seg_chunks, c1 = std::min(n_chunks_all, c0 + S.seg_chunks);
            if (int rc = sweep_and_walk(c0, c1, sg == n_seg - 1, false)) { S.want_digest = dig; return rc; }
        }
        S.want_digest = dig;
    }
    hipLaunchKernelGGL(dp_trace_finish_kernel, dim3(1), dim3(1024), 0, s, descs, S.L, S.d_path.as<uint32_t>(), col, S.cap,
                       S.d_edges.as<int32_t>(), S.d_chain.as<ChainState>(), S.d_trace.as<TraceOut>());
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
    std::vector<TeamCtl> ctl_host(TEAM_CTL_SLOTS);
    memset(ctl_host.data(), 0, sizeof(TeamCtl) * TEAM_CTL_SLOTS);
    if (team_used) DG_HIP(hipMemcpyAsync(ctl_host.data(), S.d_ctrl.p, sizeof(TeamCtl) * TEAM_CTL_SLOTS, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    for (auto &h : ctl_host) if (h.error) team_err = h.error;
    if (team_used && team_err) {        // a team could not form or a spin timed out: redo with per-level launches
        fprintf(stderr, "[dipgenie_hip] team kernel reported code %u; falling back to per-level launches\n", team_err);
        team_failed = true;
        S.team_fallbacks++;
        goto retry_forward;
    }
    S.last_team_size = team_used ? (int)ctl_host[0].team_count[(ctl_host[0].leader_xcc_plus1 - 1) & 7] : 0;
    if (getenv("DG_DEBUG")) fprintf(stderr, "[dipgenie_hip] run: host issued %lld sweep launches in %.1f ms (%.2f us each)\n", (long long)n_launch, 1e3 * host_enqueue_s, 1e6 * host_enqueue_s / (double)std::max<int64_t>(n_launch, 1));
    if (team_used && getenv("DG_DEBUG"))
        fprintf(stderr, "[dipgenie_hip] %d team launches; team size %d of %u WGs; first team: %.3f ms, shader clock %.0f MHz\n", n_team_launch,
                S.last_team_size, ctl_host[0].registered, ctl_host[0].t_real / 1e5, ctl_host[0].t_real ? 100.0 * ctl_host[0].t_cycles / ctl_host[0].t_real : 0.0);
    if (team_used && getenv("DG_TEAM_PROF")) {
        const TeamCtl &h = ctl_host[0];
        fprintf(stderr, "[dipgenie_hip] slot0 cycles: desc %llu tasks %llu drain %llu sync1 %llu barrier %llu total %llu | tables %llu relax %llu reduce %llu store %llu\n",
                h.phase[0], h.phase[1], h.phase[2], h.phase[3], h.phase[4], h.phase[5], h.tphase[0], h.tphase[1], h.tphase[2], h.tphase[3]);
    }
    DG_HIP(hipEventElapsedTime(&S.timing.delta_ms, S.ev[0], S.ev[1]));
    DG_HIP(hipEventElapsedTime(&S.timing.forward_ms, S.ev[1], S.ev[2]));
    DG_HIP(hipEventElapsedTime(&S.timing.traceback_ms, S.ev[2], S.ev[3]));
    DG_HIP(hipEventElapsedTime(&S.timing.total_ms, S.ev[0], S.ev[3]));
    S.timing.n_forward_launches = n_launch;
    if (to.value == CHAIN_CORRUPT) { set_error("back-pointer lattice is corrupt: the chain walk left its level (a level was not swept?)"); return DG_ERR_STATE; }
    if (to.overflow || to.n_e > S.cap) { set_error("traceback edge list overflow (%d > %d)", to.n_e, S.cap); return DG_ERR_STATE; }
    res->value = to.value; res->s_het = to.s_het;
    res->cells = S.cells; res->relaxations = S.relaxations;
    // records arrive in arbitrary order: path order = ascending level (the two records of the last level are equal)
    std::vector<int> order(to.n_e);
    for (int q = 0; q < to.n_e; ++q) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return edges[a] < edges[b]; });
    int n1 = 0, n2 = 0;
    for (int q : order) {
        const int from = edges[S.cap + q], tov = edges[2 * S.cap + q];
        if (edges[3 * S.cap + q] == 0) { if (n1 < res->cap && res->p1_from && res->p1_to) { res->p1_from[n1] = from; res->p1_to[n1] = tov; } ++n1; }
        else { if (n2 < res->cap && res->p2_from && res->p2_to) { res->p2_from[n2] = from; res->p2_to[n2] = tov; } ++n2; }
    }
    res->n_p1 = n1; res->n_p2 = n2;
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
extern "C" int dg_dp_prealloc(dg_ctx *c, int64_t bytes) {
    if (int rc = dgi::bind(c)) return rc;
    if (!c->dp) c->dp = new dgi::DpState();
    dgi::DpState &S = *c->dp;
    if (S.pool.chunk_units != S.chunk_units_cfg) { dgi::pool_clear(S); S.pool.chunk_units = S.chunk_units_cfg; }
    const size_t chunk_bytes = S.pool.chunk_units * 2;
    if (S.pool.cap_chunks == 0) {          // first call only: later ones may arrive while chunks are being mapped
        size_t free_b = 0, total_b = 0;
        DG_HIP(hipMemGetInfo(&free_b, &total_b));
        S.pool.cap_chunks = std::max<size_t>(1, (size_t)(0.6 * (double)free_b) / chunk_bytes);
    }
    const size_t want_chunks = bytes > 0 ? std::min(((size_t)bytes + chunk_bytes - 1) / chunk_bytes, S.pool.cap_chunks) : S.pool.cap_chunks;
    dgi::pool_request(S, c->device, want_chunks);   // whole chunks; returns immediately
    return DG_OK;
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
    dgi::graphs_clear(*c->dp);
    if (!strcmp(key, "digest")) c->dp->want_digest = v;
    else if (!strcmp(key, "fast")) c->dp->use_fast = v;
    else if (!strcmp(key, "team")) c->dp->use_team = v;
    else if (!strcmp(key, "team_grid")) c->dp->team_grid = v > 0 ? v : 256;
    else if (!strcmp(key, "team_max_tasks")) c->dp->team_max_tasks = v;
    else if (!strcmp(key, "team_min_levels")) c->dp->team_min_levels = v;
    else if (!strcmp(key, "adaptive_rc")) c->dp->adaptive_rc = v;
    else if (!strcmp(key, "segment_cells")) c->dp->segment_cells = v;
    else if (!strcmp(key, "sync_every")) c->dp->sync_every = v;
    else if (!strcmp(key, "coop")) c->dp->use_coop = v;
    else if (!strcmp(key, "coop_cost_ns")) c->dp->coop_cost_ns = v;
    else if (!strcmp(key, "chain_spec")) c->dp->chain_spec = v;
    else if (!strcmp(key, "delta_cap_entries")) c->dp->delta_cap_entries = v > 0 ? v : (int64_t)4 << 30;   // takes effect at the next load
    else if (!strcmp(key, "rc_cap")) c->dp->rc_cap = v > 0 ? v : 16384;
    else if (!strcmp(key, "rc_t0_ns")) c->dp->rc_t0_ns = v;
    else if (!strcmp(key, "rc_tg_ps")) c->dp->rc_tg_ps = v;
    else if (!strcmp(key, "rc_tw_ps")) c->dp->rc_tw_ps = v;
    else if (!strcmp(key, "warm_rows")) c->dp->warm_rows = v;
    else if (!strcmp(key, "graph_batch")) c->dp->graph_batch = v < -1 ? -1 : v;
    else if (!strcmp(key, "warm_ahead")) c->dp->warm_ahead = v < 0 ? 0 : v;
    else if (!strcmp(key, "bp_nt_min_cells")) c->dp->bp_nt_min_cells = v;
    else if (!strcmp(key, "host_threads")) c->dp->host_threads = v < 1 ? 1 : v;
    else if (!strcmp(key, "lattice_chunk_cells")) {          // size of one lattice chunk (in 16-bit back-pointer units = cells on ordinary levels; default 2^32 = 8 GB)
        if (v < 1) { dgi::set_error("lattice_chunk_cells must be positive"); return DG_ERR_ARG; }
        dgi::pool_clear(*c->dp);
        c->dp->chunk_units_cfg = c->dp->pool.chunk_units = ((size_t)v + 1) & ~(size_t)1;
    }
    else if (!strcmp(key, "waves_per_block")) c->dp->waves_per_block = (v >= 1 && v <= 4) ? v : 4;
    else if (!strcmp(key, "chip_waves")) c->dp->chip_waves = v > 0 ? v : 8192;
    else if (!strcmp(key, "max_blocks")) c->dp->max_blocks = v > 0 ? v : 2048;
    else { dgi::set_error("unknown option %s", key); return DG_ERR_ARG; }
    return DG_OK;
}
