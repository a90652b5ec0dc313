// oracle/oracle_dp.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
//
// CPU restatement of DipGenie's diploid pair-of-paths DP level loop:
//   inter_size_union2x2 / symdiff_size_union2x2   /root/reference/src/approximator.cpp:269-311
//   dp_entry / EdgeNode                           /root/reference/src/approximator.cpp:381-410
//   level loop (reset, count, prefix, fill, relax, roll, compact)   :532-716
//   sink read-out + materialize_edges             :757-764, 774-785
// Scatter form, single thread, loop order (r, i, j, adjacency(u1), adjacency(v1)) -- i.e. exactly
// what the reference executes with -t1; its output is thread-count independent.
// Pointers are replaced by indices into one node arena; the every-1000-levels pool compaction
// (:476-530, :710-713) is restated as an index mark-compact (it does not change any result).
#include "oracle.h"
#include <algorithm>
#include <cstring>
#include <limits>
#include <vector>

extern "C" int orc_inter_union2x2(const int32_t *A, int na, const int32_t *B, int nb,
                                  const int32_t *C, int nc, const int32_t *D, int nd) {
    int i = 0, j = 0, k = 0, m = 0, cnt = 0;                        // :271
    while (i < na || j < nb || k < nc || m < nd) {
        int x = std::numeric_limits<int>::max();
        if (i < na) x = std::min(x, A[i]);
        if (j < nb) x = std::min(x, B[j]);
        if (k < nc) x = std::min(x, C[k]);
        if (m < nd) x = std::min(x, D[m]);
        bool inL = false, inR = false;
        while (i < na && A[i] == x) { inL = true; ++i; }
        while (j < nb && B[j] == x) { inL = true; ++j; }
        while (k < nc && C[k] == x) { inR = true; ++k; }
        while (m < nd && D[m] == x) { inR = true; ++m; }
        if (inL && inR) ++cnt;                                      // :285
    }
    return cnt;
}

extern "C" int orc_symdiff_union2x2(const int32_t *A, int na, const int32_t *B, int nb,
                                    const int32_t *C, int nc, const int32_t *D, int nd) {
    int i = 0, j = 0, k = 0, m = 0, cnt = 0;                        // :294
    while (i < na || j < nb || k < nc || m < nd) {
        int x = std::numeric_limits<int>::max();
        if (i < na) x = std::min(x, A[i]);
        if (j < nb) x = std::min(x, B[j]);
        if (k < nc) x = std::min(x, C[k]);
        if (m < nd) x = std::min(x, D[m]);
        bool inL = false, inR = false;
        while (i < na && A[i] == x) { inL = true; ++i; }
        while (j < nb && B[j] == x) { inL = true; ++j; }
        while (k < nc && C[k] == x) { inR = true; ++k; }
        while (m < nd && D[m] == x) { inR = true; ++m; }
        if (inL ^ inR) ++cnt;                                       // :308
    }
    return cnt;
}

namespace {
struct EdgeNode { int from, to; int64_t prev; };                    // :381-388 (prev = index, -1 = null)
struct Entry {                                                      // :390-410
    int pred_i, pred_j, value, s_het;
    int64_t p1_tail, p2_tail;
    int p1_count, p2_count;
};
const int IMAX = std::numeric_limits<int>::max();
}

extern "C" int orc_dp_solve_diploid(const orc_dp_graph *g, orc_dp_result *res, uint64_t *level_digest) {
    const int L = g->n_levels, R = g->R;
    const int32_t NEG_INF = std::numeric_limits<int32_t>::min() / 4;  // :413
    if (L < 1) return -1;
    // pos_in_level (:372-379): ids are level-sorted, so pos = v - level_off[level(v)]
    std::vector<int> pos(g->n_vertices);
    for (int l = 0; l < L; ++l)
        for (int v = g->level_off[l]; v < g->level_off[l + 1]; ++v) pos[v] = v - g->level_off[l];

    auto homp = [&](int v) { return g->hom_col + g->hom_off[v]; };
    auto homn = [&](int v) { return (int)(g->hom_off[v + 1] - g->hom_off[v]); };
    auto hetp = [&](int v) { return g->het_col + g->het_off[v]; };
    auto hetn = [&](int v) { return (int)(g->het_off[v + 1] - g->het_off[v]); };

    std::vector<EdgeNode> pool1, pool2;
    std::vector<Entry> cur, nxt;
    Entry e0{IMAX, IMAX, 0, 0, -1, -1, 0, 0};
    cur.assign((size_t)(R + 1), e0);                                 // :534-535 (level 0 has k=1)
    if (g->level_off[1] - g->level_off[0] != 1) return -2;

    std::vector<int> score_deltas, s_hets;
    std::vector<size_t> base;
    uint64_t cells = 0, relax = 0;

    auto compact = [&](std::vector<EdgeNode> &pool, bool first) {   // :476-530 restated on indices
        std::vector<int64_t> remap(pool.size(), -1);
        std::vector<EdgeNode> np;
        std::vector<int64_t> stack;
        for (auto &e : cur) {
            int64_t &tail = first ? e.p1_tail : e.p2_tail;
            if (tail < 0) continue;
            stack.clear();
            int64_t n = tail;
            while (n >= 0 && remap[n] < 0) { stack.push_back(n); n = pool[n].prev; }
            int64_t prev_new = (n >= 0) ? remap[n] : -1;
            for (size_t t = stack.size(); t-- > 0;) {
                int64_t o = stack[t];
                np.push_back({pool[o].from, pool[o].to, prev_new});
                prev_new = (int64_t)np.size() - 1;
                remap[o] = prev_new;
            }
            tail = remap[tail];
        }
        pool.swap(np);
    };

    for (int l = 0; l + 1 < L; ++l) {                                // :537-540
        const int a0 = g->level_off[l], k = g->level_off[l + 1] - a0;
        const int b0 = g->level_off[l + 1], k2 = g->level_off[l + 2] - b0;
        const size_t szN = (size_t)(R + 1) * k2 * k2;
        Entry reset{IMAX, IMAX, NEG_INF, 0, -1, -1, 0, 0};           // :565-576
        nxt.assign(szN, reset);
        cells += szN;

        base.assign((size_t)k * k + 1, 0);                           // :579-601
        size_t total = 0;
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                base[(size_t)i * k + j] = total;
                total += (size_t)(g->out_off[a0 + i + 1] - g->out_off[a0 + i]) *
                         (size_t)(g->out_off[a0 + j + 1] - g->out_off[a0 + j]);
            }
        base[(size_t)k * k] = total;
        relax += total * (uint64_t)(R + 1);
        score_deltas.resize(total);
        s_hets.resize(total);

        for (int i = 0; i < k; ++i)                                  // :604-624 fill
            for (int j = 0; j < k; ++j) {
                const int u1 = a0 + i, v1 = a0 + j;
                size_t out = base[(size_t)i * k + j];
                for (int64_t eu = g->out_off[u1]; eu < g->out_off[u1 + 1]; ++eu)
                    for (int64_t ev = g->out_off[v1]; ev < g->out_off[v1 + 1]; ++ev) {
                        const int u2 = g->out_dst[eu], v2 = g->out_dst[ev];
                        int inter = orc_inter_union2x2(homp(u1), homn(u1), homp(v1), homn(v1),
                                                       homp(u2), homn(u2), homp(v2), homn(v2));
                        int symd = orc_symdiff_union2x2(hetp(u1), hetn(u1), hetp(v1), hetn(v1),
                                                        hetp(u2), hetn(u2), hetp(v2), hetn(v2));
                        s_hets[out] = symd;
                        score_deltas[out] = inter + symd;
                        ++out;
                    }
            }

        for (int r = 0; r <= R; ++r)                                 // :627-701 relaxation
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    const Entry src = cur[((size_t)r * k + i) * k + j];
                    if (src.value == NEG_INF) continue;              // :633
                    const int u1 = a0 + i, v1 = a0 + j;
                    size_t idx = base[(size_t)i * k + j];
                    for (int64_t eu = g->out_off[u1]; eu < g->out_off[u1 + 1]; ++eu) {
                        const int u2 = g->out_dst[eu], wu = g->out_w[eu];
                        const int iu2 = pos[u2];
                        for (int64_t ev = g->out_off[v1]; ev < g->out_off[v1 + 1]; ++ev) {
                            const int v2 = g->out_dst[ev], wv = g->out_w[ev];
                            const int jv2 = pos[v2];
                            const int r2 = r + wu + wv;
                            if (r2 > R) { ++idx; continue; }         // :647
                            Entry &dst = nxt[((size_t)r2 * k2 + iu2) * k2 + jv2];
                            const int cand = src.value + score_deltas[idx];
                            if (cand > dst.value ||                  // :657-659
                                (cand == dst.value && i < dst.pred_i) ||
                                (cand == dst.value && i == dst.pred_i && j < dst.pred_j)) {
                                dst.value = cand;
                                dst.s_het = src.s_het + s_hets[idx];
                                dst.pred_i = i; dst.pred_j = j;
                                dst.p1_tail = src.p1_tail; dst.p2_tail = src.p2_tail;
                                dst.p1_count = src.p1_count; dst.p2_count = src.p2_count;
                                if (wu > 0) {                        // :673-677
                                    pool1.push_back({u1, u2, dst.p1_tail});
                                    dst.p1_tail = (int64_t)pool1.size() - 1; ++dst.p1_count;
                                }
                                if (wv > 0) {                        // :678-682
                                    pool2.push_back({v1, v2, dst.p2_tail});
                                    dst.p2_tail = (int64_t)pool2.size() - 1; ++dst.p2_count;
                                }
                                if (l + 1 == L - 1) {                // :684-692 final edges, unconditional
                                    pool1.push_back({u1, u2, dst.p1_tail});
                                    dst.p1_tail = (int64_t)pool1.size() - 1; ++dst.p1_count;
                                    pool2.push_back({v1, v2, dst.p2_tail});
                                    dst.p2_tail = (int64_t)pool2.size() - 1; ++dst.p2_count;
                                }
                            }
                            ++idx;
                        }
                    }
                }
        cur.swap(nxt);                                               // :706
        if (((l + 1) % 1000) == 0) { compact(pool1, true); compact(pool2, false); }   // :710-713
        if (level_digest) {
            uint64_t d = 0;
            for (size_t t = 0; t < cur.size(); ++t)
                if (cur[t].value != NEG_INF)                         // value term + predecessor term (the :657-659 tie-break of EVERY cell)
                    d += (uint64_t)(uint32_t)(cur[t].value + 1) * (uint64_t)(t + 1) +
                         0x9E3779B97F4A7C15ULL * ((((uint64_t)cur[t].pred_i << 15) | (uint64_t)cur[t].pred_j) + 1ULL) * (uint64_t)(t + 1);
            level_digest[l + 1] = d;
        }
    }

    const int k_sink = g->level_off[L] - g->level_off[L - 1];        // :730
    const Entry &sink = cur[((size_t)R * k_sink + 0) * k_sink + 0];  // :774-775 (best_r = R)
    res->value = sink.value;
    res->s_het = sink.s_het;
    res->cells = cells;
    res->relaxations = relax;
    auto materialize = [&](const std::vector<EdgeNode> &pool, int64_t tail, int32_t *from, int32_t *to) {   // :757-764
        std::vector<std::pair<int, int>> out;
        for (int64_t c = tail; c >= 0; c = pool[c].prev) out.emplace_back(pool[c].from, pool[c].to);
        std::reverse(out.begin(), out.end());
        int n = 0;
        for (auto &p : out) { if (n < res->cap) { from[n] = p.first; to[n] = p.second; } ++n; }
        return n;
    };
    res->n_p1 = materialize(pool1, sink.p1_tail, res->p1_from, res->p1_to);
    res->n_p2 = materialize(pool2, sink.p2_tail, res->p2_from, res->p2_to);
    return 0;
}

extern "C" int orc_dp_haploid(const orc_hap_graph *g, int32_t *dp, int32_t *back_vtx, int32_t *back_r) {   // :44-72
    const int n = g->n_vertices, R = g->R;
    const size_t N = (size_t)n * (R + 1);
    for (size_t t = 0; t < N; ++t) { dp[t] = 0; back_vtx[t] = -1; back_r[t] = -1; }                       // :50-52
    auto idx = [&](int v, int r) { return (size_t)v * (R + 1) + r; };
    for (int u = 0; u < n; ++u)                                                                            // :55
        for (int r = 0; r <= R; ++r)
            for (int64_t e = g->out_off[u]; e < g->out_off[u + 1]; ++e) {
                const int v = g->out_dst[e], w = g->out_w[e];
                // :60 -- the reference compares in size_t (int + size_t > int); every value is >= 0, so int compares equal
                if (r + w <= R && (size_t)dp[idx(u, r)] + (size_t)g->n_colours[v] > (size_t)dp[idx(v, r + w)]) {
                    dp[idx(v, r + w)] = dp[idx(u, r)] + g->n_colours[v];
                    back_vtx[idx(v, r + w)] = u;
                    back_r[idx(v, r + w)] = r;
                }
            }
    return 0;
}
