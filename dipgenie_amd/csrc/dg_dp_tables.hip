// dg_dp_load_graph: validates the levelized graph, gets the sweep's tables built (on the device: dg_dp_build.hip; with option
// host_tables = 1 on the host, below -- the two must produce the same bytes) and plans the back-pointer lattice.
//
//   in-CSR            in-edges of every vertex sorted by (source position asc, adjacency order asc): rank order inside
//                     a list IS the reference's tie order (pred_i asc / pred_j asc, approximator.cpp:657-659)
//   row records       {first in-edge, in-degree, in-edge 0, in-edge 1} per vertex (16 B): 97 % of the rows need no more
//   row in-edge matrix  per level with fan-in rows: in-edge words padded to the level's largest in-degree, addressed
//                     from kernel arguments alone (LevelDesc::rowx_*)
//   slot table        64 records per column group (run of <= 64 in-edges covering whole destination columns)
//   level descriptors 120 B each, passed by value to the level's launch
// plus the plan of the back-pointer lattice (chunks / segments) and of the score-delta windows.
//
// Host-side construction runs on a few std::threads over contiguous LEVEL ranges balanced by vertex count (no OpenMP
// in this library: the caller may bring its own runtime).  Edges only go from level l to l + 1 and vertex ids are
// level-sorted, so a range of source levels owns the in-edge lists of the next levels' vertices.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "dg_dp.hpp"

namespace dgi {

namespace {

constexpr int64_t ROWX_MAX_LEVEL_CELLS = (int64_t)1 << 21;   // bigger levels are throughput-bound: one more load round is noise there
constexpr int64_t ROWX_BUDGET_WORDS = (int64_t)1 << 30;      // 4 GB of row in-edge matrices at most

struct Part {                                                // tables of one thread's level range, range-local offsets
    std::vector<int32_t> dtrans, dead_cols, heavy;
    std::vector<uint32_t> grp_begin, slots, rowx;
    std::vector<int64_t> dblk_first;
    int64_t cells = 0, units = 0, max_level_cells = 0, max_level_units = 0, delta_entries = 0, nblk = 0;
    uint64_t edge_pairs = 0, colour_entries = 0;
};

struct Builder {
    const dg_dp_graph *g;
    DpState &S;
    int nV, L, NT;
    int64_t E = 0;
    int max_k = 0;
    std::vector<int> lcut;                                   // thread t owns levels [lcut[t], lcut[t+1])
    std::vector<std::string> terr;
    std::vector<int> trc;
    std::vector<int32_t> level_of;
    std::vector<uint32_t> in_off, in_edge;
    std::vector<int32_t> in_dst;
    std::vector<uint8_t> has_col;
    std::vector<uint32_t> rowrec;
    // merged tables
    std::vector<int32_t> dtrans, dead_cols, heavy_rows;
    std::vector<uint32_t> grp_begin, slots, rowx;
    std::vector<int64_t> dblk_first;

    Builder(const dg_dp_graph *g_, DpState &S_) : g(g_), S(S_), nV(g_->n_vertices), L(g_->n_levels) {
        NT = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)S.host_threads, (int64_t)std::thread::hardware_concurrency(), (int64_t)L / 64 + 1}));
        lcut.assign(NT + 1, L);
        lcut[0] = 0;
        for (int t = 1; t < NT; ++t) {
            const int32_t want = (int32_t)((int64_t)nV * t / NT);
            lcut[t] = (int)(std::upper_bound(g->level_off, g->level_off + L + 1, want) - g->level_off) - 1;
            lcut[t] = std::min(std::max(lcut[t], lcut[t - 1]), L);
        }
        terr.resize(NT);
        trc.assign(NT, DG_OK);
    }
    template <class F> void run_threads(F &&fn) {
        std::vector<std::thread> th;
        for (int t = 1; t < NT; ++t) th.emplace_back([&fn, t] { fn(t); });
        fn(0);
        for (auto &x : th) x.join();
    }
    template <class... A> void tfail(int t, int rc, const char *fmt, A... a) {
        if (trc[t] != DG_OK) return;
        char buf[256];
        snprintf(buf, sizeof buf, fmt, a...);
        terr[t] = buf; trc[t] = rc;
    }
    int first_error() {
        for (int t = 0; t < NT; ++t) if (trc[t] != DG_OK) { set_error("%s", terr[t].c_str()); return trc[t]; }
        return DG_OK;
    }
    int range_v0(int t) const { return g->level_off[lcut[t]]; }
    int range_v1(int t) const { return lcut[t + 1] < L ? g->level_off[lcut[t + 1]] : nV; }

    int validate_and_index();
    int build_in_csr();
    int check_colours();
    int build_level_tables();
};

int Builder::validate_and_index() {
    for (int l = 0; l < L; ++l)
        if (g->level_off[l + 1] <= g->level_off[l]) { set_error("level %d is empty", l); return DG_ERR_ARG; }
    level_of.resize(nV);
    std::vector<int> tmax_k(NT, 0);
    run_threads([&](int t) {
        for (int l = lcut[t]; l < lcut[t + 1]; ++l) {
            tmax_k[t] = std::max(tmax_k[t], g->level_off[l + 1] - g->level_off[l]);
            for (int v = g->level_off[l]; v < g->level_off[l + 1]; ++v) level_of[v] = l;
        }
    });
    for (int t = 0; t < NT; ++t) max_k = std::max(max_k, tmax_k[t]);
    if (max_k > MAX_K) { set_error("level width %d exceeds the supported %d", max_k, MAX_K); return DG_ERR_UNSUPPORTED; }
    if (g->out_off[0] != 0) { set_error("out_off must start at 0"); return DG_ERR_ARG; }
    for (int t = 1; t <= NT; ++t) {                            // monotone at the range seams (inside: checked by the owner)
        const int v = lcut[t] < L ? g->level_off[lcut[t]] : nV;
        if (g->out_off[v] < 0) { set_error("out_off negative at %d", v); return DG_ERR_ARG; }
    }
    E = g->out_off[nV];
    if (E < 0 || E >= (int64_t)1 << 31) { set_error("unsupported number of edges (%lld)", (long long)E); return DG_ERR_UNSUPPORTED; }
    return DG_OK;
}

int Builder::build_in_csr() {
    in_off.assign((size_t)nV + 1, 0);
    in_edge.resize((size_t)E);
    in_dst.resize((size_t)E);
    run_threads([&](int t) {
        for (int u = range_v0(t); u < range_v1(t); ++u) {
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
        // destinations of this range: the vertices of levels [lcut[t] + 1, lcut[t+1] + 1)
        const int da = g->level_off[std::min(lcut[t] + 1, L)], db = g->level_off[std::min(lcut[t + 1] + 1, L)];
        std::vector<uint32_t> fill(in_off.begin() + da, in_off.begin() + db);
        for (int u = range_v0(t); u < range_v1(t); ++u) {
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
    return first_error();
}

int Builder::check_colours() {   // colour lists must be sorted (the merges rely on it) and fit the uint16 delta
    if (g->hom_off[0] != 0 || g->het_off[0] != 0) { set_error("colour offsets must start at 0"); return DG_ERR_ARG; }
    std::vector<int64_t> tmax_list(NT, 0);
    has_col.assign(L, 0);
    run_threads([&](int t) {
        for (int v = range_v0(t); v < range_v1(t); ++v) {
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
    return DG_OK;
}

// Level descriptors: every thread builds the groups / dead columns / slot blocks / row matrices of its levels into private
// vectors with range-local offsets, a serial prefix over the ranges turns them into global ones.
int Builder::build_level_tables() {
    S.descs.assign(L, LevelDesc{});
    S.cells = S.relaxations = S.edge_pairs = S.colour_entries = 0;
    S.total_units = 0; S.max_level_units = 0; S.max_level_cells = S.RP; S.delta_entries = DELTA_PAD;
    S.level_units.assign(L, 0);
    S.level_dmax.assign(L, 0);
    rowrec.assign((size_t)nV * 4, 0);
    std::vector<Part> part(NT);
    run_threads([&](int t) {
        Part &P = part[t];
        for (int v = range_v0(t); v < range_v1(t); ++v) {
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
        auto &gb = P.grp_begin; auto &dead = P.dead_cols; auto &slots_p = P.slots;
        for (int l = std::max(1, lcut[t]); l < lcut[t + 1]; ++l) {
            LevelDesc &d = S.descs[l];
            d.a0 = g->level_off[l - 1]; d.k = g->level_off[l] - d.a0;
            d.b0 = g->level_off[l]; d.k2 = g->level_off[l + 1] - d.b0;
            d.in_base = in_off[d.b0];
            d.T = (int32_t)(in_off[d.b0 + d.k2] - d.in_base);
            auto indeg = [&](int c) { return in_off[d.b0 + c + 1] - in_off[d.b0 + c]; };
            d.grp_first = (int32_t)gb.size();
            d.dead_first = (int32_t)dead.size();
            d.heavy_first = (int32_t)P.heavy.size();
            uint32_t max_indeg = 0;
            for (int c = 0; c < d.k2; ++c) {
                max_indeg = std::max(max_indeg, indeg(c));
                if (indeg(c) > (uint32_t)COOP_MIN) P.heavy.push_back(c);
            }
            d.n_heavy = (int32_t)P.heavy.size() - d.heavy_first;
            for (int q = 0; q < HEAVY_INLINE; ++q) d.heavy_in[q] = q < d.n_heavy ? (int16_t)P.heavy[(size_t)d.heavy_first + q] : (int16_t)-1;
            S.level_dmax[l] = (int32_t)max_indeg;
            d.dmax = (int32_t)max_indeg;
            // column groups: greedy runs of whole columns with <= 64 in-edges; a column with more gets its own group
            {
                uint32_t cur_size = 0;
                for (int c = 0; c < d.k2; ++c) {
                    const uint32_t e0 = in_off[d.b0 + c], dv = indeg(c);
                    if (dv == 0) { dead.push_back(c); continue; }
                    if (cur_size == 0 || cur_size + dv > 64 || dv > 64) { gb.push_back(e0); cur_size = 0; }
                    cur_size += dv;
                    if (dv > 64) cur_size = 65;                      // force a new group after a giant column
                }
            }
            d.ngroups = (int32_t)gb.size() - d.grp_first;
            gb.push_back(d.in_base + (uint32_t)d.T);                 // sentinel: end of the level's in-edges
            // 64-wide slot table of the fast kernel: one block per group; a giant column (in-degree > 64) takes
            // ceil(dv/64) consecutive blocks (first one tagged 15, the rest 14) and counts as that many "groups"
            d.slot_first = (int64_t)(slots_p.size() / 2);
            d.fast_ok = (d.T < (1 << 20)) ? 1 : 0;                   // the slot word keeps 20 bits of in-edge index
            d.bp_wide = max_indeg > (uint32_t)BP_MAX_RANK ? 1 : 0;  // ranks do not fit 8 bits: wide words, generic kernel
            if (d.bp_wide) d.fast_ok = 0;
            int32_t n_blocks = 0;
            for (int gi = 0; gi < d.ngroups; ++gi) {
                const uint32_t gb0 = gb[d.grp_first + gi], ge0 = gb[d.grp_first + gi + 1];
                const bool giant = ge0 - gb0 > 64;
                if (giant && d.fast_ok) d.fast_ok = 2;               // the general sweep variant
                uint32_t maxdv = 1;
                if (!giant)
                    for (uint32_t e = gb0; e < ge0;) {
                        const uint32_t dv = indeg(in_dst[e] - d.b0);
                        maxdv = std::max(maxdv, dv);
                        e += dv;
                    }
                uint32_t steps = 0;
                while ((1u << steps) < std::min(maxdv, 64u)) ++steps;
                const uint32_t nb = giant ? (ge0 - gb0 + 63) / 64 : 1;
                const size_t s0 = slots_p.size();
                slots_p.resize(s0 + (size_t)nb * 128);
                uint32_t *sp = slots_p.data() + s0;
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
                for (int q = 0; q < 64; ++q) { slots_p.push_back(0xFFFFFFFFu); slots_p.push_back(0); }
                n_blocks = 1;
            }
            d.nblocks = n_blocks;
            d.ndead = (int32_t)dead.size() - d.dead_first;
            if (d.ngroups == 0) { d.ngroups = 1; gb.push_back(d.in_base + (uint32_t)d.T); }   // level without in-edges: one empty group
            const int64_t ncell = (int64_t)d.k2 * d.k2 * S.RP;
            // row in-edge matrix (LevelDesc::rowx_*): fan-in rows on latency-bound levels of the lean variant
            d.rowx_off = (int64_t)P.rowx.size();
            d.rowx_stride = 0;
            if (S.use_rowx && d.fast_ok == 1 && max_indeg > 2 && max_indeg <= (uint32_t)ROWX_MAX && ncell <= ROWX_MAX_LEVEL_CELLS) {
                d.rowx_stride = (int32_t)max_indeg;
                P.rowx.resize(P.rowx.size() + (size_t)d.k2 * max_indeg, 0u);
                uint32_t *rx = P.rowx.data() + d.rowx_off;
                for (int c = 0; c < d.k2; ++c)
                    for (uint32_t q = 0; q < indeg(c); ++q) rx[(size_t)c * max_indeg + q] = in_edge[in_off[d.b0 + c] + q];
            }
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
    std::vector<int64_t> b_grp(NT + 1, 0), b_dead(NT + 1, 0), b_slot(NT + 1, 0), b_cells(NT + 1, 0), b_units(NT + 1, 0), b_delta(NT + 1, DELTA_PAD),
        b_blk(NT + 1, 0), b_dt(NT + 1, 0), b_heavy(NT + 1, 0), b_rowx(NT + 1, 0);
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
        b_rowx[t + 1] = b_rowx[t] + (int64_t)P.rowx.size();
        S.max_level_cells = std::max(S.max_level_cells, P.max_level_cells);
        S.max_level_units = std::max(S.max_level_units, P.max_level_units);
        S.edge_pairs += P.edge_pairs;
        S.colour_entries += P.colour_entries;
    }
    S.total_units = b_units[NT];
    {   // the lean chain walk (dg_dp_trace.hip) addresses a level with 32-bit offsets and 24-bit multiplies
        int kmax = 1;
        bool wide = false;
        for (int l = 1; l < L; ++l) { kmax = std::max(kmax, S.descs[l].k2); wide |= S.descs[l].bp_wide != 0; }
        S.lean_chain = S.use_lean_chain && !wide && S.max_level_cells < ((int64_t)1 << 30) && (int64_t)S.RP * kmax < ((int64_t)1 << 24) && nV < (1 << 27);
    }
    S.cells = (uint64_t)b_cells[NT];
    S.delta_entries = b_delta[NT];
    S.n_delta_blocks = b_blk[NT];
    S.relaxations = S.edge_pairs * (uint64_t)S.RP;
    if (b_grp[NT] >= (int64_t)1 << 31 || b_dead[NT] >= (int64_t)1 << 31) { set_error("group tables too large"); return DG_ERR_UNSUPPORTED; }
    if (S.n_delta_blocks >= (int64_t)1 << 31) { set_error("delta grid too large"); return DG_ERR_UNSUPPORTED; }
    const int64_t rowx_words = std::min(b_rowx[NT], ROWX_BUDGET_WORDS);   // matrices past the budget are dropped (their levels take the slower path)
    dtrans.resize((size_t)b_dt[NT]); dead_cols.resize((size_t)b_dead[NT]); heavy_rows.resize((size_t)b_heavy[NT] + 1);
    grp_begin.resize((size_t)b_grp[NT]);
    slots.resize((size_t)b_slot[NT] * 2);                         // 2 words per slot, 64 slots per block
    rowx.assign((size_t)rowx_words + 4, 0u);
    dblk_first.resize((size_t)b_dt[NT]);
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
            d.rowx_off += b_rowx[t];
            if (d.rowx_stride > 0 && d.rowx_off + (int64_t)d.k2 * d.rowx_stride > rowx_words) d.rowx_stride = 0;
        }
        std::copy(P.grp_begin.begin(), P.grp_begin.end(), grp_begin.begin() + b_grp[t]);
        std::copy(P.dead_cols.begin(), P.dead_cols.end(), dead_cols.begin() + b_dead[t]);
        std::copy(P.heavy.begin(), P.heavy.end(), heavy_rows.begin() + b_heavy[t]);
        std::copy(P.slots.begin(), P.slots.end(), slots.begin() + 2 * b_slot[t]);
        std::copy(P.dtrans.begin(), P.dtrans.end(), dtrans.begin() + b_dt[t]);
        if (b_rowx[t] < rowx_words)
            std::copy(P.rowx.begin(), P.rowx.begin() + std::min<int64_t>((int64_t)P.rowx.size(), rowx_words - b_rowx[t]), rowx.begin() + b_rowx[t]);
        for (size_t q = 0; q < P.dblk_first.size(); ++q) dblk_first[(size_t)b_dt[t] + q] = P.dblk_first[q] + b_blk[t];
        std::vector<uint32_t>().swap(P.slots); std::vector<uint32_t>().swap(P.rowx);
    });
    return DG_OK;
}

// score-delta windows (see DpState::dwin_t)
void plan_delta_windows(DpState &S, const std::vector<int32_t> &dtrans, const std::vector<int64_t> &dblk_first) {
    S.dtrans_host = dtrans;
    S.dblk_first_host = dblk_first;
    S.dblk_first_host.push_back(S.n_delta_blocks);
    S.dwin_t.assign(1, 0);
    S.level_win.assign(S.L, -1);
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

// Memory budget, lattice chunking / segmentation.  Levels are packed into equal chunks (a level never straddles two).
// If all chunks fit they stay resident; otherwise a segment = as many consecutive chunks as fit (pool chunks are reused
// by every segment) and the run goes checkpoint + recompute.  segment_cells (tests) caps the chunk size and forces one
// chunk per segment.  On success: bp_bytes / ck_bytes = bytes of lattice resident at a time / of checkpoints.
int plan_lattice(dg_ctx *c, DpState &S, size_t st_bytes, size_t dl_bytes, size_t table_bytes, size_t &bp_bytes, size_t &ck_bytes, bool dbg) {
    const int L = S.L;
    size_t free_b = 0, total_b = 0;
    DG_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t pool_bytes;
    { std::unique_lock<std::mutex> lk(S.pool.mu); pool_bytes = S.pool.chunks.size() * S.pool.chunk_units * 2; }
    const size_t have = free_b + pool_bytes + S.d_bp.bytes + S.d_delta.bytes + S.d_ring.bytes + S.d_ckpt.bytes;
    const size_t fixed = st_bytes + dl_bytes + table_bytes + ((size_t)2 << 30);     // state, delta, tables, slack
    if (fixed > have) {
        set_error("graph needs %.1f GB of HBM for state/delta/tables but only %.1f GB is free", fixed / 1e9, have / 1e9);
        return DG_ERR_OOM;
    }
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
        // Beyond HBM every level is swept twice whatever the segment size, but a segment's chunks must be mapped before its re-sweep
        // starts and mapping runs at ~85 GB/s (the 5 Mbp x 100-walk panel waited 3.5 s for 33 chunks = all of HBM): stay near what
        // the background thread has mapped by now -- more, smaller segments cost one tiny checkpoint each
        if (S.segment_cells == 0) group = std::min(group, std::max<size_t>(8, pool_bytes / chunk_bytes + 2));
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
    bp_bytes = tiny ? resident_bytes : group * chunk_bytes;
    ck_bytes = (size_t)ckpt_cells * 4;
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
    return DG_OK;
}

int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t s) {
    if (int rc = b.ensure(bytes)) return rc;
    if (bytes) DG_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
    return DG_OK;
}

}  // namespace

int dp_load(dg_ctx *c, const dg_dp_graph *g) {
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
    if (!c->dp) c->dp = new DpState(c->device);
    DpState &S = *c->dp;
    graphs_clear(S);
    S.loaded = false;
    S.nV = nV; S.L = L; S.R = R; S.RP = R + 1; S.rp_active = S.RP;
    hipStream_t s = c->stream;
    PoolPause pause(S);                                         // the pool thread maps no chunk while this function allocates
    std::vector<int32_t> dtrans;
    std::vector<int64_t> dblk_first;
    int max_k = 1;
    if (!S.host_tables) {
        if (int rc = dp_build_tables_device(c, g, S, dtrans, dblk_first, max_k)) return rc;
    } else {
        Builder B(g, S);
        if (int rc = B.validate_and_index()) return rc;
        if (int rc = B.build_in_csr()) return rc;
        lap("validate + in-CSR");
        if (int rc = B.check_colours()) return rc;
        lap("colour checks");
        if (int rc = B.build_level_tables()) return rc;
        lap("descs + groups + slots");
        if (int rc = upload(S.d_descs, S.descs.data(), sizeof(LevelDesc) * L, s)) return rc;
        if (int rc = upload(S.d_in_off, B.in_off.data(), 4 * B.in_off.size(), s)) return rc;
        if (int rc = upload(S.d_in_edge, B.in_edge.data(), 4 * B.in_edge.size(), s)) return rc;
        if (int rc = upload(S.d_in_dst, B.in_dst.data(), 4 * B.in_dst.size(), s)) return rc;
        if (int rc = upload(S.d_hom_off, g->hom_off, 8 * ((size_t)nV + 1), s)) return rc;
        if (int rc = upload(S.d_het_off, g->het_off, 8 * ((size_t)nV + 1), s)) return rc;
        if (int rc = upload(S.d_hom_col, g->hom_col, 4 * (size_t)g->hom_off[nV], s)) return rc;
        if (int rc = upload(S.d_het_col, g->het_col, 4 * (size_t)g->het_off[nV], s)) return rc;
        if (int rc = upload(S.d_dtrans, B.dtrans.data(), 4 * B.dtrans.size(), s)) return rc;
        if (int rc = upload(S.d_dblk_first, B.dblk_first.data(), 8 * B.dblk_first.size(), s)) return rc;
        if (int rc = upload(S.d_grp, B.grp_begin.data(), 4 * B.grp_begin.size(), s)) return rc;
        if (int rc = upload(S.d_dead, B.dead_cols.data(), 4 * B.dead_cols.size(), s)) return rc;
        if (int rc = upload(S.d_heavy, B.heavy_rows.data(), 4 * B.heavy_rows.size(), s)) return rc;
        if (int rc = upload(S.d_rowrec, B.rowrec.data(), 4 * B.rowrec.size(), s)) return rc;
        if (int rc = upload(S.d_rowx, B.rowx.data(), 4 * B.rowx.size(), s)) return rc;
        if (int rc = upload(S.d_slots, B.slots.data(), 4 * B.slots.size(), s)) return rc;
        DG_HIP(hipStreamSynchronize(s));                        // the staging vectors die with B
        S.n_grp = (int64_t)B.grp_begin.size(); S.n_dead = (int64_t)B.dead_cols.size(); S.n_heavy_rows = (int64_t)B.heavy_rows.size() - 1;
        S.n_slot_records = (int64_t)B.slots.size() / 2; S.n_rowx_words = (int64_t)B.rowx.size() - 4;
        dtrans.swap(B.dtrans); dblk_first.swap(B.dblk_first);
        max_k = B.max_k;
        lap("table uploads");
    }
    S.n_dtrans = (int64_t)dtrans.size();
    S.n_edges = g->out_off[nV];
    plan_delta_windows(S, dtrans, dblk_first);
    const size_t n_edges = (size_t)g->out_off[nV];
    const size_t st_bytes = (size_t)S.max_level_cells * 4 * 2, dl_bytes = (size_t)S.delta_buf_entries * 2;
    size_t bp_bytes = 0, ck_bytes = 0;
    // (the tables are allocated already; what follows: edge flags + self scores, digests, the path)
    if (int rc = plan_lattice(c, S, st_bytes, dl_bytes, 3 * n_edges + 16 * (size_t)L + (1 << 20), bp_bytes, ck_bytes, dbg)) return rc;
    lap("plan lattice");
    if (int rc = S.d_eflag.ensure(n_edges + 16)) return rc;
    if (int rc = S.d_eself.ensure(2 * n_edges + 16)) return rc;
    delta_launch_edge_flags(S, s);
    if (int rc = S.d_delta.ensure(dl_bytes)) return rc;
    DG_HIP(hipMemsetAsync(S.d_delta.p, 0, 2 * DELTA_PAD, s));
    if (int rc = S.d_ckpt.ensure(ck_bytes)) return rc;
    if (int rc = S.d_chain.ensure(128)) return rc;                 // ChainState, and ChainSync at +64
    if (int rc = S.d_pfctl.ensure(256)) return rc;                  // PfCtl of the L2 table prefetcher
    DG_HIP(hipMemsetAsync(S.d_pfctl.p, 0, 256, s));
    DG_HIP(hipMemsetAsync(S.d_chain.p, 0, 128, s)); S.chain_seq = 0;
    S.pad_front = 2 * (int64_t)max_k;
    const size_t pad_bytes = 4 * (size_t)(S.pad_front + 33 * (int64_t)max_k);
    S.state_alloc_bytes = (((size_t)S.max_level_cells * 4 + pad_bytes) + 255) & ~(size_t)255;   // one slot
    if (int rc = S.d_ring.ensure(S.state_alloc_bytes * 2)) return rc;          // the two ping-pong slots
    DG_HIP(hipMemsetAsync(S.d_ring.p, 0, S.state_alloc_bytes * 2, s));
    if (int rc = S.d_digest.ensure(8 * (size_t)L)) return rc;
#ifdef DG_SWEEP_PROBE
    if (int rc = S.d_probe.ensure(64 * (size_t)L)) return rc;
#endif
    if (int rc = S.d_trace.ensure(sizeof(TraceOut))) return rc;
    S.cap = 2 * (R + 8);                               // edge records of both paths
    if (int rc = S.d_edges.ensure(4 * 4 * (size_t)S.cap)) return rc;
    if (int rc = S.d_path.ensure(8 * (size_t)L)) return rc;
    lap("allocs");
    for (auto &e : S.ev) if (!e) DG_HIP(hipEventCreate(&e));
    memset(&S.timing, 0, sizeof S.timing);
    S.timing.edge_pairs = S.edge_pairs;
    S.timing.colour_entries = S.colour_entries;
    S.timing.state_bytes = st_bytes; S.timing.bp_bytes = bp_bytes; S.timing.delta_bytes = dl_bytes;
    S.loaded = true;
    return DG_OK;
}

// FNV-1a over every table of the resident graph (bp_nt is a launch-time field): the device and the host construction must agree
int dp_table_digest(dg_ctx *c, uint64_t *out, int n) {
    DpState *Sp = c->dp;
    if (!Sp || !Sp->loaded) { set_error("dg_dp_get_table_digest: no graph loaded"); return DG_ERR_STATE; }
    if (!out || n < 12) { set_error("dg_dp_get_table_digest: need 12 words"); return DG_ERR_ARG; }
    DpState &S = *Sp;
    DG_HIP(hipStreamSynchronize(c->stream));
    std::vector<unsigned char> h;
    auto fnv = [&](const DevBuf &b, size_t bytes, uint64_t &dst) -> int {
        h.resize(bytes);
        if (bytes) DG_HIP(hipMemcpy(h.data(), b.p, bytes, hipMemcpyDeviceToHost));
        uint64_t x = 1469598103934665603ULL;
        for (size_t i = 0; i < bytes; ++i) { x ^= h[i]; x *= 1099511628211ULL; }
        dst = x;
        return DG_OK;
    };
    {
        std::vector<LevelDesc> d(S.L);
        DG_HIP(hipMemcpy(d.data(), S.d_descs.p, sizeof(LevelDesc) * (size_t)S.L, hipMemcpyDeviceToHost));
        uint64_t x = 1469598103934665603ULL;
        for (auto &q : d) { q.bp_nt = 0; const unsigned char *p = (const unsigned char *)&q; for (size_t i = 0; i < sizeof q; ++i) { x ^= p[i]; x *= 1099511628211ULL; } }
        out[0] = x;
    }
    if (int rc = fnv(S.d_in_off, 4 * ((size_t)S.nV + 1), out[1])) return rc;
    if (int rc = fnv(S.d_in_edge, 4 * (size_t)S.n_edges, out[2])) return rc;
    if (int rc = fnv(S.d_in_dst, 4 * (size_t)S.n_edges, out[3])) return rc;
    if (int rc = fnv(S.d_dtrans, 4 * (size_t)S.n_dtrans, out[4])) return rc;
    if (int rc = fnv(S.d_dblk_first, 8 * (size_t)S.n_dtrans, out[5])) return rc;
    if (int rc = fnv(S.d_grp, 4 * (size_t)S.n_grp, out[6])) return rc;
    if (int rc = fnv(S.d_dead, 4 * (size_t)S.n_dead, out[7])) return rc;
    if (int rc = fnv(S.d_heavy, 4 * (size_t)S.n_heavy_rows, out[8])) return rc;
    if (int rc = fnv(S.d_rowrec, 16 * (size_t)S.nV, out[9])) return rc;
    if (int rc = fnv(S.d_rowx, 4 * (size_t)S.n_rowx_words, out[10])) return rc;
    if (int rc = fnv(S.d_slots, 8 * (size_t)S.n_slot_records, out[11])) return rc;
    return DG_OK;
}

}  // namespace dgi
