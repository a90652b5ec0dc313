// Pipeline::build_levelized_fast -- the diploid route from Anchor_hits to the levelized DP graph in one fused, threaded
// construction.  It produces exactly what the literal route of pipeline.cpp produces by running, one after the other,
//   Approximator::solve's graph construction           /root/reference/src/approximator.cpp:1017-1246
//   ExpandedGraph::topologically_reorder                /root/reference/src/ExpandedGraph.hpp:29-102
//   ExpandedGraph::strict_bfs_levelize_and_reorder      /root/reference/src/ExpandedGraph.hpp:269-409
//   the HOM / HET colour split of the DP prologue       /root/reference/src/approximator.cpp:431-453
// (same vertex numbering, same adjacency order, same colour lists: tests compare the two routes' .dpg dumps byte by byte),
// without a push log, without the two whole-graph permutations and without materialising the intermediate graphs:
//
//  A. the pre-order graph G0 in the reference's numbering (source, chain vertices in (haplotype, step) order, sink,
//     recombination vertices in discovery order, anchor super-nodes in Anchor_hits order) as a CSR written in place: the
//     adjacency of a vertex is [chain edge | weight-1 edges | start->super edges | overlap edges] = its push order
//     (SURVEY.md Appendix A.5); everything a haplotype's loop iterations push goes to that haplotype's own vertices, so
//     haplotypes fill their rows in parallel;
//  B. one Kahn pass (FIFO, sink last: ExpandedGraph.hpp:37-62) that also relaxes the longest-path levels.  A FIFO Kahn
//     order is sorted by longest-path depth, so the later stable sort by (level, id) of the levelizer keeps it: the final id
//     of a real vertex is its Kahn position plus the number of dummies on lower levels (checked, not assumed: any other
//     input takes the literal route);
//  C. dummy vertices (edges spanning more than one level, ExpandedGraph.hpp:326-352) get their rank inside their level by a
//     two-pass count in creation order (source ascending, edge order, step);
//  D. the final arrays (out-CSR, HOM / HET colour CSR, haplotype, original-vertex lists) are written once, in parallel,
//     from closed-form offsets (prefix sums over the Kahn order).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <thread>

#include "pipeline.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace dg {

namespace {
struct Lap {
    bool on;
    double t;
    Lap() : on(getenv("DG_DEBUG") != nullptr), t(now_s()) {}
    void operator()(const char *w) { if (on) { const double n = now_s(); fprintf(stderr, "[dg::fast] %-20s %.3f s\n", w, n - t); t = n; } }
};
struct Proxy { int32_t startExp, endExp, idx; };             // what the (startExp, endExp) sort of approximator.cpp:1203 looks at
}  // namespace

bool Pipeline::build_levelized_fast(ExpandedGraph &g, std::vector<std::vector<AnchorRec>> &anchorsByHap, std::vector<uint8_t> &color_homo_bv) {
    Lap lap;
    const int H = (int)paths.size();
    const int NT = std::max(1, std::min(opt.threads, 64));
    if (H == 0) return false;
    int64_t N64 = 0;
    std::vector<int64_t> base(H + 1, 0);
    for (int h = 0; h < H; ++h) { if (paths[h].empty()) return false; base[h + 1] = base[h] + (int64_t)paths[h].size(); }
    N64 = base[H];
    if (N64 + 2 >= (int64_t)1 << 30) return false;
    const int32_t N = (int32_t)N64, sink = N + 1;
    auto cid = [&](int h, size_t i) { return (int32_t)(1 + base[h] + (int64_t)i); };

    // ---- A1: vertex_to_expanded_map (approximator.cpp:1023, :1035: the last occurrence wins) ----
    uvec<int32_t> v2e((size_t)n_vtx * H);
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int64_t q = 0; q < (int64_t)v2e.size(); ++q) v2e[q] = -1;
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h)
        for (size_t i = 0; i < paths[h].size(); ++i) v2e[(size_t)paths[h][i] * H + h] = cid(h, i);

    // ---- A2: recombination events (approximator.cpp:1051-1095) per haplotype, in (step, adjacency) order ----
    std::vector<int64_t> wslot_off((size_t)n_vtx + 1, 0);            // vertex_w_uv[u][j] flattened
    for (size_t u = 0; u < adj_list.size(); ++u) wslot_off[u + 1] = wslot_off[u] + (int64_t)adj_list[u].size();
    struct Ev { int32_t slot, src; };
    std::vector<uvec<Ev>> events(H);
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        auto &ev = events[h];
        ev.reserve(paths[h].size() / 2 + 16);
        const size_t n = paths[h].size();
        for (size_t i = 0; i < n; ++i) {
            const uint32_t u = paths[h][i];
            const auto &au = adj_list[u];
            const int32_t src = v2e[(size_t)u * H + h];
            for (size_t j = 0; j < au.size(); ++j)
                if (i == n - 1 || au[j] != paths[h][i + 1]) ev.push_back(Ev{(int32_t)(wslot_off[u] + (int64_t)j), src});
        }
    }
    // discovery order of the w_uv vertices: first touch in (haplotype, step, adjacency) order -- serial, one compare per event
    std::vector<int32_t> w_of_slot((size_t)wslot_off[n_vtx], -1), slot_of_w;
    for (int h = 0; h < H; ++h)
        for (const Ev &e : events[h])
            if (w_of_slot[e.slot] < 0) { w_of_slot[e.slot] = (int32_t)slot_of_w.size(); slot_of_w.push_back(e.slot); }
    const int32_t W = (int32_t)slot_of_w.size(), w_base = N + 2;
    std::vector<int32_t> slot_vtx((size_t)wslot_off[n_vtx]);         // slot -> target original vertex v of (u, j)
    for (size_t u = 0; u < adj_list.size(); ++u)
        for (size_t j = 0; j < adj_list[u].size(); ++j) slot_vtx[wslot_off[u] + (int64_t)j] = (int32_t)adj_list[u][j];
    lap("v2e + recomb events");

    // ---- A3: anchors -> records per haplotype (approximator.cpp:1114-1176), in Anchor_hits order ----
    const int64_t n_occ = (int64_t)occs.size();
    const int s_base = w_base + W;                                   // first super-node id
    const int NC = std::max(1, std::min<int>(NT * 4, (int)std::max<int64_t>(1, n_occ / 4096)));
    auto c_lo = [&](int c) { return n_occ * c / NC; };
    uvec<int32_t> sE((size_t)n_occ), eE((size_t)n_occ);
    std::vector<int64_t> ch_newa(NC + 1, 0), ch_super(NC + 1, 0), ch_pool(NC + 1, 0);
    std::vector<int64_t> ch_hap((size_t)(NC + 1) * H, 0);
    bool bad_occ = false;
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1) reduction(|| : bad_occ)
    for (int c = 0; c < NC; ++c) {
        int64_t newa = 0, nsup = 0, npool = 0;
        int64_t *hc = ch_hap.data() + (size_t)(c + 1) * H;
        for (int64_t p = c_lo(c); p < c_lo(c + 1); ++p) {
            const Occ &o = occs[p];
            if (o.h < 0 || o.h >= H || o.len == 0) { bad_occ = true; continue; }
            const int32_t s = v2e[(size_t)vpool[o.off] * H + o.h], e = v2e[(size_t)vpool[o.off + o.len - 1] * H + o.h];
            if (s < 0 || e < 0) bad_occ = true;
            sE[p] = s; eE[p] = e;
            newa += p == 0 || occs[p - 1].a != o.a;
            if (s != e) { ++nsup; npool += o.len; }
            ++hc[o.h];
        }
        ch_newa[c + 1] = newa; ch_super[c + 1] = nsup; ch_pool[c + 1] = npool;
    }
    if (bad_occ) return false;
    for (int c = 0; c < NC; ++c) {
        ch_newa[c + 1] += ch_newa[c]; ch_super[c + 1] += ch_super[c]; ch_pool[c + 1] += ch_pool[c];
        for (int h = 0; h < H; ++h) ch_hap[(size_t)(c + 1) * H + h] += ch_hap[(size_t)c * H + h];
    }
    const int n_colours = (int)ch_newa[NC];
    const int64_t S64 = ch_super[NC];
    if ((int64_t)s_base + S64 >= (int64_t)1 << 30) return false;
    const int32_t S = (int32_t)S64, n0 = s_base + S;
    sum.n_colours = n_colours;
    std::vector<int32_t> color_to_anchor(n_colours);
    anchorsByHap.assign(H, {});
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) anchorsByHap[h].resize((size_t)ch_hap[(size_t)NC * H + h]);
    // per G0 vertex: haplotype, original-vertex list (chain vertex: its path step; super-node: the anchor's vertex list)
    uvec<int32_t> hap0(n0);
    uvec<uint32_t> ooff0(n0), olen0(n0);
    uvec<int32_t> opool((size_t)N + (size_t)ch_pool[NC]);
    uvec<int32_t> sup_end(S), sup_hap(S);                            // super-node -> endExp, haplotype
    hap0[0] = 0; ooff0[0] = 0; olen0[0] = 0; hap0[sink] = 0; ooff0[sink] = 0; olen0[sink] = 0;
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h)
        for (size_t i = 0; i < paths[h].size(); ++i) {
            const int32_t c = cid(h, i);
            hap0[c] = h; ooff0[c] = (uint32_t)(c - 1); olen0[c] = 1; opool[c - 1] = (int32_t)paths[h][i];
        }
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t w = 0; w < W; ++w) { hap0[w_base + w] = -1; ooff0[w_base + w] = 0; olen0[w_base + w] = 0; }
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int c = 0; c < NC; ++c) {
        int64_t colour = ch_newa[c] - 1, sidx = ch_super[c], pool = (int64_t)N + ch_pool[c];
        std::vector<int64_t> hpos(ch_hap.begin() + (size_t)c * H, ch_hap.begin() + (size_t)(c + 1) * H);
        for (int64_t p = c_lo(c); p < c_lo(c + 1); ++p) {
            const Occ &o = occs[p];
            if (p == 0 || occs[p - 1].a != o.a) { ++colour; color_to_anchor[colour] = o.a; }
            int32_t node = sE[p];
            if (sE[p] != eE[p]) {
                node = s_base + (int32_t)sidx;
                sup_end[sidx] = eE[p]; sup_hap[sidx] = o.h;
                hap0[node] = -1; ooff0[node] = (uint32_t)pool; olen0[node] = o.len;
                std::copy(vpool.begin() + o.off, vpool.begin() + o.off + o.len, opool.begin() + pool);
                pool += o.len; ++sidx;
            }
            AnchorRec &r = anchorsByHap[o.h][(size_t)hpos[o.h]++];
            r.startOrg = vpool[o.off]; r.endOrg = vpool[o.off + o.len - 1]; r.startExp = sE[p]; r.endExp = eE[p];
            r.colours = ColourList{(int)colour}; r.nodeID = node;
        }
    }
    lap("anchor records");

    // ---- A4: per-vertex out-degree of everything but the overlap edges; start->super edges per haplotype in Anchor_hits order ----
    uvec<int32_t> deg0(n0);                                          // becomes the cursor of the fill below
    uvec<int32_t> n_ov(n0);                                          // overlap edges per source
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t v = 0; v < n0; ++v) { deg0[v] = 0; n_ov[v] = 0; }
    // ---- A5: per-haplotype sweep (approximator.cpp:1193-1246): overlap edges + containment colour propagation ----
    struct OvEdge { int32_t src, dst; };
    std::vector<uvec<OvEdge>> ov_edges(H);
    uvec<int32_t> ncol0(n0), nhom0(n0);                              // colours / HOM colours per G0 vertex
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t v = 0; v < n0; ++v) { ncol0[v] = 0; nhom0[v] = 0; }
    std::vector<uvec<int32_t>> sup_edges(H);                         // (startExp, super) pairs, flattened, in Anchor_hits order
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        auto &vec = anchorsByHap[h];
        for (const Ev &e : events[h]) ++deg0[e.src];                  // weight-1 edges
        for (const AnchorRec &r : vec)
            if (r.startExp != r.endExp) { ++deg0[r.startExp]; sup_edges[h].push_back(r.startExp); sup_edges[h].push_back(r.nodeID); }
        if (vec.empty()) continue;
        // std::sort is not stable: the permutation depends on the comparisons alone, so sorting light proxies with the same
        // comparator from the same initial order reproduces what sorting the records themselves would give
        std::vector<Proxy> px(vec.size());
        for (size_t q = 0; q < vec.size(); ++q) px[q] = Proxy{vec[q].startExp, vec[q].endExp, (int32_t)q};
        std::sort(px.begin(), px.end(), [](const Proxy &a, const Proxy &b) {
            if (a.startExp != b.startExp) return a.startExp < b.startExp;
            else return a.endExp < b.endExp;
        });
        {
            std::vector<AnchorRec> sorted(vec.size());
            for (size_t q = 0; q < vec.size(); ++q) sorted[q] = std::move(vec[px[q].idx]);
            vec.swap(sorted);
        }
        std::vector<AnchorRec *> stk;
        auto &ov = ov_edges[h];
        for (auto &anc : vec) {
            while (!stk.empty() && stk.back()->endExp < anc.startExp) stk.pop_back();
            if (!stk.empty() && anc.startExp <= stk.back()->endExp && stk.back()->nodeID != anc.nodeID) {
                ov.push_back(OvEdge{stk.back()->nodeID, anc.nodeID});
                ++n_ov[stk.back()->nodeID];
            }
            for (int i = (int)stk.size() - 1; i >= 0; --i) {
                if (anc.endExp <= stk[i]->endExp) {
                    for (int c : anc.colours)
                        if (std::find(stk[i]->colours.begin(), stk[i]->colours.end(), c) == stk[i]->colours.end())
                            stk[i]->colours.push_back(c);
                } else break;
            }
            stk.push_back(&anc);
        }
        for (const auto &anc : vec) ncol0[anc.nodeID] += (int32_t)anc.colours.size();   // (with repeats; made unique below)
    }
    lap("sweep");

    // ---- A6: G0 CSR offsets ----
    uvec<int64_t> off0((size_t)n0 + 1);
    // fan-out of the w_uv vertices (:1082-1090): every haplotype that holds v, in haplotype order
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t w = 0; w < W; ++w) {
        const int32_t v = slot_vtx[slot_of_w[w]];
        int32_t d = 0;
        for (int hh = 0; hh < H; ++hh) d += v2e[(size_t)v * H + hh] >= 0;
        deg0[w_base + w] = d;
    }
    {
        // degrees: source H, chain 1 + (weight-1) + (start->super) + overlap, sink 0, w_uv fan-out, super 1 + overlap
        int64_t run = 0;
        for (int32_t v = 0; v < n0; ++v) {
            int64_t d;
            if (v == 0) d = H;
            else if (v <= N) d = 1 + (int64_t)deg0[v] + n_ov[v];
            else if (v == sink) d = 0;
            else if (v < s_base) d = deg0[v];
            else d = 1 + (int64_t)n_ov[v];
            off0[v] = run;
            run += d;
        }
        off0[n0] = run;
        if (run >= (int64_t)1 << 31) return false;
    }
    const int64_t E0 = off0[n0];
    uvec<int32_t> dst0((size_t)E0);
    uvec<uint8_t> w0((size_t)E0);
    uvec<int32_t> indeg(n0);
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t v = 0; v < n0; ++v) indeg[v] = 0;
    // ---- A7: fill, every haplotype its own rows (cursor = deg0, reset to the position after the chain edge) ----
    for (int h = 0; h < H; ++h) { dst0[off0[0] + h] = cid(h, 0); w0[off0[0] + h] = 0; }
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const size_t n = paths[h].size();
        for (size_t i = 0; i < n; ++i) {                               // chain edge first (:1041 / :1045)
            const int32_t c = cid(h, i);
            dst0[off0[c]] = i + 1 < n ? c + 1 : sink; w0[off0[c]] = 0;
            deg0[c] = 1;
        }
        for (const Ev &e : events[h]) {                                // weight-1 edges in event order (:1078)
            const int64_t o = off0[e.src] + deg0[e.src]++;
            dst0[o] = w_base + w_of_slot[e.slot]; w0[o] = 1;
        }
        const auto &se = sup_edges[h];
        for (size_t q = 0; q < se.size(); q += 2) {                    // start -> super (:1148)
            const int64_t o = off0[se[q]] + deg0[se[q]]++;
            dst0[o] = se[q + 1]; w0[o] = 0;
        }
    }
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t s = 0; s < S; ++s) { const int32_t v = s_base + s; dst0[off0[v]] = sup_end[s]; w0[off0[v]] = 0; deg0[v] = 1; }   // super -> end (:1149)
#pragma omp parallel for num_threads(NT) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h)
        for (const OvEdge &e : ov_edges[h]) {                          // overlap edges, appended after everything else (:1222)
            const int64_t o = off0[e.src] + deg0[e.src]++;
            dst0[o] = e.dst; w0[o] = 0;
        }
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t w = 0; w < W; ++w) {
        const int32_t v = slot_vtx[slot_of_w[w]];
        int64_t o = off0[w_base + w];
        for (int hh = 0; hh < H; ++hh) { const int32_t ve = v2e[(size_t)v * H + hh]; if (ve >= 0) { dst0[o] = ve; w0[o] = 0; ++o; } }
    }
    { uvec<int32_t>().swap(v2e); }
    // (nothing below needs the fill's inputs: at chr22 scale they are several GB each)
    { std::vector<uvec<Ev>>().swap(events); std::vector<uvec<int32_t>>().swap(sup_edges); std::vector<uvec<OvEdge>>().swap(ov_edges); uvec<int32_t>().swap(n_ov); uvec<int32_t>().swap(deg0);
      uvec<int32_t>().swap(sE); uvec<int32_t>().swap(eE); }
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int64_t e = 0; e < E0; ++e) {
#pragma omp atomic
        ++indeg[dst0[e]];
    }
    lap("G0 CSR");

    // ---- B: Kahn (FIFO; the sink only when nothing else is left) + longest-path levels: one thread, the chain is serial ...
    uvec<int32_t> order(n0), lvl(n0);
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t v = 0; v < n0; ++v) lvl[v] = 0;
    int kahn_rc = 0;                                                 // 1: not exactly one source, 2: cycle
    auto kahn = [&]() {
        int32_t n_src = 0;
        size_t tail = 0;
        for (int32_t v = 0; v < n0; ++v)
            if (indeg[v] == 0) { if (off0[v + 1] > off0[v]) ++n_src; if (v != sink) order[tail++] = v; }
        if (n_src != 1) { kahn_rc = 1; return; }                      // (the literal route reports it as the reference does)
        bool sink_ready = indeg[sink] == 0;
        size_t head = 0;
        while (head < tail || sink_ready) {
            int32_t u;
            if (head < tail) u = order[head++];
            else { u = sink; sink_ready = false; order[tail++] = sink; ++head; }
            const int32_t lu = lvl[u] + 1;
            for (int64_t e = off0[u]; e < off0[u + 1]; ++e) {
                const int32_t v = dst0[e];
                if (lvl[v] < lu) lvl[v] = lu;
                if (--indeg[v] == 0) { if (v == sink) sink_ready = true; else order[tail++] = v; }
            }
        }
        if ((int32_t)tail != n0) kahn_rc = 2;                         // cycle: the literal route throws
    };
    std::thread kahn_thread;
    struct JoinGuard { std::thread &t; ~JoinGuard() { if (t.joinable()) t.join(); } } kahn_guard{kahn_thread};   // an allocation below may throw: never unwind past a joinable thread
    const int NTC = NT > 1 ? NT - 1 : 1;                             // threads of the colour work beside it
    if (NT > 1) kahn_thread = std::thread(kahn); else kahn();
    wait_fit();                                                      // the grid fit has been running beside everything above: homo_bv from here on
    color_homo_bv.assign(n_colours, 0);                              // approximator.cpp:1283-1290
    for (int c = 0; c < n_colours; ++c) if (homo_bv[color_to_anchor[c]]) color_homo_bv[c] = 1;
    // ---- ... while the others build the colour CSR of G0 (:1240-1245: per node, the sorted-unique union of its records' lists).
    // Node ids of different haplotypes are disjoint: count (done in the sweep), scatter, then sort + unique the few nodes that
    // hold more than one colour.
    uvec<int64_t> coff_raw((size_t)n0 + 1);
    { int64_t run = 0; for (int32_t v = 0; v < n0; ++v) { coff_raw[v] = run; run += ncol0[v]; } coff_raw[n0] = run; }
    uvec<int32_t> craw((size_t)coff_raw[n0]);
#pragma omp parallel for num_threads(NTC) schedule(static)
    for (int32_t v = 0; v < n0; ++v) ncol0[v] = 0;                   // now: the scatter cursor
#pragma omp parallel for num_threads(NTC) schedule(dynamic, 1)
    for (int h = 0; h < H; ++h)
        for (const auto &anc : anchorsByHap[h])
            for (int c : anc.colours) craw[coff_raw[anc.nodeID] + ncol0[anc.nodeID]++] = c;
    std::vector<int64_t> cc(NTC + 1, 0);
    auto v_lo = [&](int c) { return (int32_t)((int64_t)n0 * c / NTC); };
#pragma omp parallel for num_threads(NTC) schedule(static, 1)
    for (int c = 0; c < NTC; ++c) {
        int64_t tot = 0;
        for (int32_t v = v_lo(c); v < v_lo(c + 1); ++v) {
            int32_t *a = craw.data() + coff_raw[v];
            int32_t n = ncol0[v];
            if (n > 1) { std::sort(a, a + n); n = (int32_t)(std::unique(a, a + n) - a); ncol0[v] = n; }
            int32_t nh = 0;
            for (int32_t q = 0; q < n; ++q) nh += color_homo_bv[a[q]] == 1;
            nhom0[v] = nh;
            tot += n;
        }
        cc[c + 1] = tot;
    }
    for (int c = 0; c < NTC; ++c) cc[c + 1] += cc[c];
    uvec<int64_t> coff0((size_t)n0 + 1);
    uvec<int32_t> cpool0((size_t)cc[NTC]);
#pragma omp parallel for num_threads(NTC) schedule(static, 1)
    for (int c = 0; c < NTC; ++c) {
        int64_t o = cc[c];
        for (int32_t v = v_lo(c); v < v_lo(c + 1); ++v) {
            coff0[v] = o;
            const int32_t *a = craw.data() + coff_raw[v];
            for (int32_t q = 0; q < ncol0[v]; ++q) cpool0[o++] = a[q];
        }
    }
    coff0[n0] = cc[NTC];
    { uvec<int32_t>().swap(craw); uvec<int64_t>().swap(coff_raw); }
    lap("colour CSR");
    if (kahn_thread.joinable()) kahn_thread.join();
    if (kahn_rc != 0) return false;
    { uvec<int32_t>().swap(indeg); }
    const int32_t max_level = lvl[order[n0 - 1]];
    const int32_t L = max_level + 1;
    lap("kahn + levels");
    // the Kahn order must be sorted by level (it is, for a FIFO queue; a deeper vertex after the sink would break it)
    bool sorted = true;
#pragma omp parallel for num_threads(NT) schedule(static) reduction(&& : sorted)
    for (int32_t t = 1; t < n0; ++t) sorted = sorted && lvl[order[t - 1]] <= lvl[order[t]];
    if (!sorted) return false;
    std::vector<int32_t> rs((size_t)L + 1, 0);                       // first Kahn position of every level
    {
        for (int32_t t = 0; t < n0; ++t) ++rs[(size_t)lvl[order[t]] + 1];
        for (int32_t l = 0; l < L; ++l) rs[l + 1] += rs[l];
    }
    // ---- C: dummies.  Creation order = (Kahn position of the source, edge order, step); rank inside the level = creation order ----
    const int NK = NT;
    auto t_lo = [&](int c) { return (int32_t)((int64_t)n0 * c / NK); };
    std::vector<int64_t> dbase_c(NK + 1, 0);                         // dummies created before chunk c
    uvec<int32_t> hist((size_t)NK * L);
#pragma omp parallel for num_threads(NT) schedule(static, 1)
    for (int c = 0; c < NK; ++c) {
        int32_t *hc = hist.data() + (size_t)c * L;
        std::fill(hc, hc + L, 0);
        int64_t nd = 0;
        for (int32_t t = t_lo(c); t < t_lo(c + 1); ++t) {
            const int32_t u = order[t], lu = lvl[u];
            for (int64_t e = off0[u]; e < off0[u + 1]; ++e) {
                const int32_t lv = lvl[dst0[e]];
                for (int32_t l = lu + 1; l < lv; ++l) ++hc[l];
                if (lv - lu - 1 > 0) nd += lv - lu - 1;
            }
        }
        dbase_c[c + 1] = nd;
    }
    for (int c = 0; c < NK; ++c) dbase_c[c + 1] += dbase_c[c];
    const int64_t n_dummy = dbase_c[NK];
    const int64_t n1l = (int64_t)n0 + n_dummy;
    if (n1l >= INT32_MAX) throw std::runtime_error("expanded graph too large");
    const int32_t n1 = (int32_t)n1l;
    std::vector<int64_t> dbefore((size_t)L + 1, 0);                  // dummies on lower levels
    {
        std::vector<int32_t> nd_level(L, 0);
#pragma omp parallel for num_threads(NT) schedule(static)
        for (int32_t l = 0; l < L; ++l) {
            int32_t run = 0;
            for (int c = 0; c < NK; ++c) { int32_t &x = hist[(size_t)c * L + l]; const int32_t n = x; x = run; run += n; }
            nd_level[l] = run;
        }
        for (int32_t l = 0; l < L; ++l) dbefore[l + 1] = dbefore[l] + nd_level[l];
    }
    g = ExpandedGraph();
    g.n = n1;
    g.level_off.resize((size_t)L + 1);
    for (int32_t l = 0; l <= L; ++l) g.level_off[l] = (int32_t)(rs[l] + dbefore[l]);
    if (be.hint_dp_soon && opt.ploidy == 2) {                        // level widths are final: the exact lattice size
        double cells = 0;
        for (int l = 1; l < L; ++l) { const double kw = (double)(g.level_off[l + 1] - g.level_off[l]); cells += kw * kw; }
        be.hint_dp_soon(be.ctx, (int64_t)std::min(9.0e18, cells * (opt.R + 1)));
    }
    // prefix sums over the Kahn order: out-degree, HOM / HET colour counts
    uvec<int64_t> PD((size_t)n0 + 1), PH((size_t)n0 + 1), PT((size_t)n0 + 1);
    {
        std::vector<int64_t> cd(NK + 1, 0), chm(NK + 1, 0), cht(NK + 1, 0);
#pragma omp parallel for num_threads(NT) schedule(static, 1)
        for (int c = 0; c < NK; ++c) {
            int64_t d = 0, hm = 0, ht = 0;
            for (int32_t t = t_lo(c); t < t_lo(c + 1); ++t) { const int32_t u = order[t]; d += off0[u + 1] - off0[u]; hm += nhom0[u]; ht += ncol0[u] - nhom0[u]; }
            cd[c + 1] = d; chm[c + 1] = hm; cht[c + 1] = ht;
        }
        for (int c = 0; c < NK; ++c) { cd[c + 1] += cd[c]; chm[c + 1] += chm[c]; cht[c + 1] += cht[c]; }
#pragma omp parallel for num_threads(NT) schedule(static, 1)
        for (int c = 0; c < NK; ++c) {
            int64_t d = cd[c], hm = chm[c], ht = cht[c];
            for (int32_t t = t_lo(c); t < t_lo(c + 1); ++t) {
                PD[t] = d; PH[t] = hm; PT[t] = ht;
                const int32_t u = order[t];
                d += off0[u + 1] - off0[u]; hm += nhom0[u]; ht += ncol0[u] - nhom0[u];
            }
        }
        PD[n0] = cd[NK]; PH[n0] = chm[NK]; PT[n0] = cht[NK];
    }
    const int64_t E1 = E0 + n_dummy;
    if (E1 >= (int64_t)1 << 31) return false;
    lap("dummy ranks + prefixes");

    // ---- D: final arrays.  Real vertex at Kahn position t on level l: id t + dbefore[l], first out-edge PD[t] + dbefore[l];
    //         dummy of rank r on level l: id rs[l+1] + dbefore[l] + r, its one out-edge PD[rs[l+1]] + dbefore[l] + r. ----
    g.adj_off.resize((size_t)n1 + 1); g.adj_dst.resize((size_t)E1); g.adj_w.resize((size_t)E1);
    g.haplotype.resize(n1); g.orig_off.resize(n1); g.orig_len.resize(n1);
    g.hom_off.resize((size_t)n1 + 1); g.het_off.resize((size_t)n1 + 1);
    g.hom_col.resize((size_t)PH[n0]); g.het_col.resize((size_t)PT[n0]);
    g.orig_pool.swap(opool);
    uvec<int32_t> fid(n0);                                           // G0 id -> final id
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t t = 0; t < n0; ++t) fid[order[t]] = (int32_t)(t + dbefore[lvl[order[t]]]);
#pragma omp parallel for num_threads(NT) schedule(static)
    for (int32_t l = 0; l < L; ++l) {                                // the dummies' offsets: constant steps inside a level
        const int64_t nd = dbefore[l + 1] - dbefore[l], f0 = rs[l + 1] + dbefore[l], e0 = PD[rs[l + 1]] + dbefore[l];
        const int64_t hm = PH[rs[l + 1]], ht = PT[rs[l + 1]];
        for (int64_t r = 0; r < nd; ++r) { g.adj_off[f0 + r] = e0 + r; g.adj_w[e0 + r] = 0; g.hom_off[f0 + r] = hm; g.het_off[f0 + r] = ht; }
    }
    g.adj_off[n1] = E1; g.hom_off[n1] = PH[n0]; g.het_off[n1] = PT[n0];
#pragma omp parallel for num_threads(NT) schedule(static, 1)
    for (int c = 0; c < NK; ++c) {
        int32_t *hc = hist.data() + (size_t)c * L;                   // next rank of this chunk's dummies on every level
        for (int32_t t = t_lo(c); t < t_lo(c + 1); ++t) {
            const int32_t u = order[t], lu = lvl[u];
            const int64_t db = dbefore[lu], f = t + db;
            int64_t eo = PD[t] + db;
            g.adj_off[f] = eo;
            g.haplotype[f] = hap0[u]; g.orig_off[f] = ooff0[u]; g.orig_len[f] = olen0[u];
            int64_t ph = PH[t], pt = PT[t];
            g.hom_off[f] = ph; g.het_off[f] = pt;
            for (int64_t q = coff0[u]; q < coff0[u + 1]; ++q) { const int32_t col = cpool0[q]; if (color_homo_bv[col] == 1) g.hom_col[ph++] = col; else g.het_col[pt++] = col; }
            for (int64_t e = off0[u]; e < off0[u + 1]; ++e, ++eo) {
                const int32_t v = dst0[e], lv = lvl[v];
                g.adj_w[eo] = w0[e];
                if (lv - lu - 1 <= 0) { g.adj_dst[eo] = fid[v]; continue; }
                int64_t slot = eo;                                     // where the next hop is written
                for (int32_t l = lu + 1; l < lv; ++l) {
                    const int64_t r = hc[l]++, fd = rs[l + 1] + dbefore[l] + r;
                    g.adj_dst[slot] = (int32_t)fd;
                    g.haplotype[fd] = hap0[u]; g.orig_off[fd] = ooff0[u]; g.orig_len[fd] = olen0[u];
                    slot = PD[rs[l + 1]] + dbefore[l] + r;
                }
                g.adj_dst[slot] = fid[v];
            }
        }
    }
    g.colours_split = true;
    lap("emit");
    return true;
}

}  // namespace dg
