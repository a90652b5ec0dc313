#include "pipeline.hpp"

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <limits>
#include <numeric>
#include <queue>
#include <stdexcept>
#include <string_view>
#include <unordered_map>
#include <unordered_set>

#include "seq_reader.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace dg {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void Pipeline::stamp(const char *name, double t0) {
    double dt = now_s() - t0;
    static std::mutex mu;                                              // (the sharded CLI reads the GFA and the reads on two threads)
    std::lock_guard<std::mutex> lk(mu);
    sum.stage_s.emplace_back(name, dt);
    if (!opt.quiet) fprintf(stderr, "[dg::stage] %-28s %.3f s\n", name, dt);
}

// ======================================================================================
// Solver::read_gfa  (solver.cpp:27-227)
// ======================================================================================
void Pipeline::read_gfa_from(const GfaGraph &g) {
    // graph: one vertex per segment, forward-strand arcs only (:60-91); walks -> paths, named sample.hap (:108-125)
    n_vtx = g.n_seg();
    node_seq.assign(g.seg_seq.begin(), g.seg_seq.end());
    adj_list.assign(n_vtx, {});
    for (uint32_t seg = 0; seg < n_vtx; ++seg)
        for (uint32_t arc : g.arcs[(size_t)2 * seg]) adj_list[seg].push_back(arc >> 1);
    num_walks = (uint32_t)g.walks.size();
    paths.assign(num_walks, {});
    hap_id2name.assign(num_walks, std::string());
    for (uint32_t w = 0; w < num_walks; ++w) {
        hap_id2name[w] = g.walks[w].sample + "." + std::to_string(g.walks[w].hap);
        paths[w].reserve(g.walks[w].v.size());
        for (uint32_t oriented : g.walks[w].v) {
            if (oriented & 1) exit(1);                                 // a reverse-strand step ends the run silently (:116-119)
            paths[w].push_back(oriented >> 1);
        }
    }
    // MSA-like column of every vertex (:127-171): seeded with the earliest step index at which any walk visits it
    // (vertices on no walk are parked one column past the largest seed), then raised until column[next] > column[prev]
    // holds along every walk.  The least such assignment is a longest-path labelling of the walk-step graph: one pass
    // in topological order when that graph is acyclic; otherwise the reference's sweep-until-stable loop, cap included.
    const int32_t nv = (int32_t)n_vtx;
    constexpr int64_t UNSEEN = std::numeric_limits<int64_t>::max() / 4;
    std::vector<int64_t> column(nv, UNSEEN);
    for (const auto &walk : paths)
        for (size_t step = 0; step < walk.size(); ++step) column[walk[step]] = std::min(column[walk[step]], (int64_t)step);
    {
        int64_t last_seed = -1;
        for (int64_t c : column) if (c != UNSEEN) last_seed = std::max(last_seed, c);
        for (int64_t &c : column) if (c == UNSEEN) c = last_seed + 1;
    }
    bool labelled = false;
    {
        std::vector<int64_t> succ_off((size_t)nv + 1, 0);
        for (const auto &walk : paths) for (size_t t = 1; t < walk.size(); ++t) succ_off[walk[t - 1] + 1]++;
        for (int32_t v = 0; v < nv; ++v) succ_off[v + 1] += succ_off[v];
        std::vector<int32_t> succ((size_t)succ_off[nv]), pending(nv, 0), order;
        std::vector<int64_t> cursor(succ_off.begin(), succ_off.end() - 1);
        for (const auto &walk : paths)
            for (size_t t = 1; t < walk.size(); ++t) { succ[cursor[walk[t - 1]]++] = (int32_t)walk[t]; pending[walk[t]]++; }
        order.reserve(nv);
        for (int32_t v = 0; v < nv; ++v) if (!pending[v]) order.push_back(v);
        std::vector<int64_t> raised(column);
        for (size_t at = 0; at < order.size(); ++at) {
            const int32_t u = order[at];
            for (int64_t e = succ_off[u]; e < succ_off[u + 1]; ++e) {
                raised[succ[e]] = std::max(raised[succ[e]], raised[u] + 1);
                if (--pending[succ[e]] == 0) order.push_back(succ[e]);
            }
        }
        if ((int32_t)order.size() == nv) { column.swap(raised); labelled = true; }
    }
    for (int sweep = 0, cap = std::max(10, nv); !labelled && sweep < cap; ++sweep) {   // cyclic step graph (:158-171)
        labelled = true;
        for (const auto &walk : paths)
            for (size_t t = 1; t < walk.size(); ++t)
                if (column[walk[t]] <= column[walk[t - 1]]) { column[walk[t]] = column[walk[t - 1]] + 1; labelled = false; }
    }
    // top_order_map (:174-199) = rank of the vertex in (column, id) order; adjacency lists (:216-223) sorted by (dense
    // column rank, id).  Both are total orders, so one 64-bit key per vertex serves either sort.
    std::vector<int32_t> by_column(nv);
    std::iota(by_column.begin(), by_column.end(), 0);
    std::stable_sort(by_column.begin(), by_column.end(), [&](int32_t x, int32_t y) { return column[x] < column[y]; });   // ids ascending inside a column
    top_order_map.assign(nv, -1);
    std::vector<uint64_t> sort_key(nv);
    int64_t n_columns = -1, last_column = std::numeric_limits<int64_t>::min();
    for (int32_t rank = 0; rank < nv; ++rank) {
        const int32_t v = by_column[rank];
        top_order_map[v] = rank;
        if (column[v] != last_column) { ++n_columns; last_column = column[v]; }
        sort_key[v] = ((uint64_t)n_columns << 32) | (uint32_t)v;
    }
    for (auto &targets : adj_list)
        std::sort(targets.begin(), targets.end(), [&](uint32_t x, uint32_t y) { return sort_key[x] < sort_key[y]; });
}

int Pipeline::load_graph(std::string &err) {
    double t0 = now_s();
    GfaGraph g;
    if (!read_gfa_file(opt.gfa_file, g, err)) return -1;
    const double t1 = now_s();
    read_gfa_from(g);
    if (getenv("DG_DEBUG")) fprintf(stderr, "[dg::gfa] file %.3f s, read_gfa (adjacency, paths, column order) %.3f s\n", t1 - t0, now_s() - t1);
    stamp("gfa_read+read_gfa", t0);
    return 0;
}

int Pipeline::load_reads(std::string &err) {                           // solver.cpp:230-245
    double t0 = now_s();
    reads.clear();
    if (!read_sequences(opt.reads_file, reads, err)) return -1;
    stamp("read_ip_reads", t0);
    return 0;
}

// ======================================================================================
// Solver::compute_and_classify_anchors  (solver.cpp:449-887)
// ======================================================================================
int Pipeline::compute_and_classify_anchors(std::string &err) {
    const int k = opt.k;
    // Order of the stages: the reference sketches the haplotypes first, then the reads, then joins, then fits (solver.cpp:449-887).
    // Nothing in the haplotype index depends on the reads and the fit needs only the reads' multiplicity histogram, so the reads are
    // sketched FIRST and the fit (0.15 s of host arithmetic on MHC-24) runs on its thread beside the haplotype index and the anchor
    // join instead of in front of the graph stage.  The log lines keep the reference's order.
    double t0 = now_s();
    // ---- read sketches: Read_hashes / Sp_R / kmer_count (solver.cpp:526-555, 711-732) ----
    std::vector<uint64_t> sp_hash;     // sorted distinct read-minimizer hashes; id = rank (:541-546)
    std::vector<int32_t> sp_count;     // number of reads containing it (== kmer_count)
    if (spectrum_injected) {                                           // the read-sharded ranks' merged spectrum (dist_sketch.py)
        for (size_t q = 1; q < inj_sp_hash.size(); ++q)
            if (inj_sp_hash[q] <= inj_sp_hash[q - 1]) { err = "injected spectrum: hashes must be strictly ascending"; return -1; }
        if (inj_sp_count.size() != inj_sp_hash.size()) { err = "injected spectrum: one count per hash"; return -1; }
        sp_hash = inj_sp_hash; sp_count = inj_sp_count;
    } else {
        std::vector<int64_t> off(reads.size() + 1, 0);
        for (size_t r = 0; r < reads.size(); ++r) off[r + 1] = off[r] + (int64_t)reads[r].second.size();
        std::string bases;
        bases.reserve((size_t)off.back());
        for (auto &rd : reads) bases += rd.second;
        uint64_t *hh = nullptr; int32_t *cc = nullptr; int64_t n = 0;
        int rc = be.sketch_reads(be.ctx, bases.data(), off.data(), (int64_t)reads.size(), k, opt.w, &hh, &cc, &n);
        if (rc != 0) { err = std::string("sketch_reads failed: ") + (be.last_error ? be.last_error() : "?"); return -1; }
        sp_hash.assign(hh, hh + n);
        sp_count.assign(cc, cc + n);
        be.free_buf(hh); be.free_buf(cc);
    }
    count_sp_r = (int32_t)sp_hash.size();
    sum.spectrum = count_sp_r;
    stamp("compute_hashes+Sp_R", t0);
    // ---- multiplicity histogram, fit, classify (:745-879) ----
    std::map<int32_t, int32_t> kmer_freq;                              // :745-750
    for (int32_t c : sp_count) kmer_freq[c] += 1;
    if (spectrum_injected && !inj_hist.empty()) {                      // the ranks' all-reduced Hist_kmer must be the histogram of the counts they sent
        std::vector<int64_t> mine(inj_hist.size(), 0);
        for (auto &kv : kmer_freq) mine[std::min<size_t>((size_t)std::max(kv.first, 0), mine.size() - 1)] += kv.second;
        if (mine != inj_hist) { err = "injected multiplicity histogram does not match the injected counts"; return -1; }
    }
    std::vector<HistBin> hist;
    int max_mult = 0;
    for (auto &kv : kmer_freq) { hist.push_back({(int)kv.first, (double)kv.second}); max_mult = std::max(max_mult, (int)kv.first); }
    // The grid fit (serial in the reference, :785) needs nothing but the histogram, and nothing before the colour split of the
    // graph stage needs its result (homo_bv): it runs on a thread of its own beside the first phases of that stage
    // (wait_fit() joins it and prints its two lines).
    auto job = [this, hist, max_mult, threads = std::max(1, opt.threads / 2)]() {
        sum.fit = kg_fit(hist, /*max_copy=*/10, max_mult, threads);
        const KGParams &P = sum.fit.P;
        std::vector<int8_t> label(max_mult + 1, -1);
        homo_bv.assign(count_sp_r, 0);                                 // :830-879
        fit_n_hom = 0;
        for (int32_t id = 0; id < count_sp_r; ++id) {
            int m = fit_sp_count[id];
            if (m <= 0 || m > max_mult) continue;                      // (0: solver.cpp:845; the rest cannot come from the device path, and injected spectra are checked in dgr_inject_spectrum)
            if (label[m] < 0) label[m] = kg_is_hom(P, m) ? 1 : 0;
            homo_bv[id] = (uint8_t)label[m];
            fit_n_hom += label[m];
        }
    };
    fit_sp_count.swap(sp_count);
    fit_pending = true;
    if (opt.threads > 1 && !getenv("DG_FIT_INLINE")) fit_thread = std::thread(job); else job();
    // ---- haplotype sketches (index_kmers, solver.cpp:277-363) ----
    t0 = now_s();
    struct HapIndex { std::vector<uint64_t> hash; std::vector<uint32_t> voff; std::vector<int32_t> v; };
    std::vector<HapIndex> kmer_index(num_walks);
    sum.minimizers_per_hap.assign(num_walks, 0);
    double t_sketch = 0;
    int rc_sketch = 0;
    std::string err_sketch;
    // Device path: the backend keeps every haplotype's minimizers and vertex lists on the device
    // (dg_anchor_*) and later returns the finished occurrence list; the host path below does the same with the
    // position lists of be.sketch_haplotype.  Both end in the same `occs` / `vpool` (tests/golden/anchors.json).
    bool dev_anchors = be.anchor_begin && be.anchor_add_haplotype && be.anchor_finish && !opt.host_anchors;
    for (const HapSketch &hs : inj_hap) if (hs.set && !be.anchor_add_haplotype_sketched) dev_anchors = false;   // (a backend without the import: host index)
    auto host_index = [&]() {
    // The backend calls are issued by one thread, back to back (a ctx is not thread-safe); the position -> vertex-span
    // mapping of a finished haplotype (:343-357) runs as a task on the other threads meanwhile.
#pragma omp parallel num_threads(opt.threads)
#pragma omp single
    for (uint32_t h = 0; h < num_walks && rc_sketch == 0; ++h) {
        std::string hap;                                               // :283-285
        auto *seg_start_p = new std::vector<int64_t>(paths[h].size() + 1, 0);
        {
            std::vector<int64_t> &seg_start = *seg_start_p;
            size_t tot = 0;
            for (size_t i = 0; i < paths[h].size(); ++i) { seg_start[i] = (int64_t)tot; tot += node_seq[paths[h][i]].size(); }
            seg_start[paths[h].size()] = (int64_t)tot;
            hap.reserve(tot);
            for (size_t i = 0; i < paths[h].size(); ++i) hap += node_seq[paths[h][i]];
        }
        uint64_t *hh = nullptr; int64_t *pp = nullptr; int64_t n = 0;
        const double ts0 = now_s();
        int rc = 0;
        if (h < inj_hap.size() && inj_hap[h].set) {                    // sketched by another rank: same (malloc'ed) hand-off as the backend's
            n = (int64_t)inj_hap[h].hash.size();
            hh = (uint64_t *)malloc(8 * (size_t)(n + 1)); pp = (int64_t *)malloc(8 * (size_t)(n + 1));
            std::copy(inj_hap[h].hash.begin(), inj_hap[h].hash.end(), hh);
            std::copy(inj_hap[h].pos.begin(), inj_hap[h].pos.end(), pp);
        } else {
            rc = be.sketch_haplotype(be.ctx, hap.data(), (int64_t)hap.size(), k, opt.w, &hh, &pp, &n);
        }
        t_sketch += now_s() - ts0;
        if (rc != 0) {
            rc_sketch = rc;
            err_sketch = std::string("sketch_haplotype failed: ") + (be.last_error ? be.last_error() : "?");
            delete seg_start_p;
            break;
        }
        sum.minimizers_per_hap[h] = n;
#pragma omp task firstprivate(h, hh, pp, n, seg_start_p)
        {
            const std::vector<int64_t> &seg_start = *seg_start_p;
            HapIndex &ix = kmer_index[h];
            ix.hash.assign(hh, hh + n);
            ix.voff.reserve(n + 1);
            ix.voff.push_back(0);
            std::vector<int32_t> uniq;
            size_t seg = 0;
            for (int64_t m = 0; m < n; ++m) {                          // :343-357 position -> vertex span
                int64_t p = pp[m];
                // positions are non-decreasing; seg = index of the path step containing base p
                if (seg_start[seg] > p) seg = 0;
                while (seg + 1 < seg_start.size() - 1 && seg_start[seg + 1] <= p) ++seg;
                uniq.clear();
                size_t s2 = seg;
                for (;;) {
                    int32_t vtx = (int32_t)paths[h][s2];
                    if (seg_start[s2 + 1] > seg_start[s2] &&           // empty segments contribute no base
                        std::find(uniq.begin(), uniq.end(), vtx) == uniq.end()) uniq.push_back(vtx);
                    if (seg_start[s2 + 1] >= p + k) break;
                    ++s2;
                }
                std::sort(uniq.begin(), uniq.end(), [&](int32_t a, int32_t b) { return top_order_map[a] < top_order_map[b]; });
                ix.v.insert(ix.v.end(), uniq.begin(), uniq.end());
                ix.voff.push_back((uint32_t)ix.v.size());
            }
            if (h < inj_hap.size() && inj_hap[h].set) { free(hh); free(pp); } else { be.free_buf(hh); be.free_buf(pp); }
            delete seg_start_p;
        }
    }
    };
    if (dev_anchors) {
        if (be.anchor_begin(be.ctx, (int32_t)num_walks, (int32_t)n_vtx, top_order_map.data(), k, opt.w) != 0) {
            err = std::string("anchor_begin failed: ") + (be.last_error ? be.last_error() : "?"); return -1;
        }
        // the next haplotype's string and step arrays are assembled on a helper thread while the device works on this one
        struct HapInput { std::string hap; std::vector<int64_t> seg_start; std::vector<int32_t> step_vtx; size_t tot = 0; };
        auto assemble = [this](uint32_t h) {
            HapInput in;
            const size_t ns = paths[h].size();
            in.seg_start.assign(ns + 1, 0);
            in.step_vtx.assign(paths[h].begin(), paths[h].end());
            size_t tot = 0;
            for (size_t i = 0; i < ns; ++i) { in.seg_start[i] = (int64_t)tot; tot += node_seq[paths[h][i]].size(); }
            in.seg_start[ns] = (int64_t)tot;
            in.tot = tot;
            if (!(h < inj_hap.size() && inj_hap[h].set)) { in.hap.reserve(tot); for (size_t i = 0; i < ns; ++i) in.hap += node_seq[paths[h][i]]; }
            return in;
        };
        // (several helpers: one assembly takes longer than the device needs for a haplotype -- 6 ms against 2 on MHC-24)
        const uint32_t depth = (uint32_t)std::max(1, std::min(opt.threads - 1, 6));
        std::vector<std::future<HapInput>> ahead(depth);
        for (uint32_t q = 0; q < depth && q < num_walks; ++q) ahead[q] = std::async(std::launch::async, assemble, q);
        for (uint32_t h = 0; h < num_walks; ++h) {
            HapInput cur = ahead[h % depth].get();
            if (h + depth < num_walks) ahead[h % depth] = std::async(std::launch::async, assemble, h + depth);
            const std::string &hap = cur.hap;
            const std::vector<int64_t> &seg_start = cur.seg_start;
            const std::vector<int32_t> &step_vtx = cur.step_vtx;
            const size_t ns = paths[h].size(), tot = cur.tot;
            int64_t n = 0;
            const double ts0 = now_s();
            if (h < inj_hap.size() && inj_hap[h].set) {                // sketched by another rank
                n = (int64_t)inj_hap[h].hash.size();
                if (be.anchor_add_haplotype_sketched(be.ctx, (int32_t)h, (int64_t)tot, inj_hap[h].hash.data(), inj_hap[h].pos.data(), n, step_vtx.data(),
                                                     seg_start.data(), (int64_t)ns) != 0) {
                    err = std::string("anchor_add_haplotype_sketched failed: ") + (be.last_error ? be.last_error() : "?"); return -1;
                }
            } else if (be.anchor_add_haplotype(be.ctx, (int32_t)h, hap.data(), (int64_t)hap.size(), step_vtx.data(), seg_start.data(), (int64_t)ns, &n) != 0) {
                err = std::string("anchor_add_haplotype failed: ") + (be.last_error ? be.last_error() : "?"); return -1;
            }
            t_sketch += now_s() - ts0;
            sum.minimizers_per_hap[h] = n;
        }
    } else {
        host_index();
    }
    if (rc_sketch != 0) { err = err_sketch; return -1; }
    if (!opt.quiet) {
        std::cerr << "Number of Minimizers" << std::endl;              // :467-474
        for (uint32_t h = 0; h < num_walks; ++h) fprintf(stderr, "%s : %d\n", hap_id2name[h].c_str(), (int)sum.minimizers_per_hap[h]);
    }
    if (getenv("DG_DEBUG")) fprintf(stderr, "[dg::index] backend sketch calls %.3f s of %.3f s\n", t_sketch, now_s() - t0);
    stamp("index_kmers", t0);

    if (!opt.quiet) fprintf(stderr, "[M::%s] Indexed reads with spectrum size: %d\n", __func__, count_sp_r);   // :558

    // ---- compute_anchors (solver.cpp:415-446, 560-575): hap minimizers whose hash is in Sp_R ----
    t0 = now_s();
    // Once the device work of this stage is done the device side may reserve the DP lattice: mapping 100+ GB takes
    // seconds during which every other HIP call of the process queues behind the allocation, so it must start where
    // only host work follows -- before the host join, after the device join.
    auto hint_lattice = [&]() {
        if (!(opt.ploidy == 2 && be.hint_dp_soon)) return;
        size_t max_path = 0;
        for (auto &pw : paths) max_path = std::max(max_path, pw.size());
        // Generous on purpose (levels ~ 2.5 x path steps, width ~ 5 x walks: chain + recombination + dummy vertices):
        // reserving too much costs nothing once the exact figure (diploid(), below) stops it, too little stalls the DP.
        const double kk = 5.0 * (double)num_walks;
        be.hint_dp_soon(be.ctx, (int64_t)std::min(9.0e18, 2.5 * (double)max_path * kk * kk * (opt.R + 1)));
    };
    if (!dev_anchors) hint_lattice();
    const bool dbg_a = getenv("DG_DEBUG") != nullptr;
    double tla = now_s();
    auto lap_a = [&](const char *w) { if (dbg_a) { double t = now_s(); fprintf(stderr, "[dg::anchors] %-18s %.3f s\n", w, t - tla); tla = t; } };
    auto host_join = [&]() {
    struct Raw { int32_t h; uint32_t m; };   // minimizer m of haplotype h
    std::vector<int64_t> bucket_off((size_t)count_sp_r + 1, 0);
    std::vector<std::vector<int32_t>> ids(num_walks);
    for (uint32_t h = 0; h < num_walks; ++h) ids[h].resize(kmer_index[h].hash.size());
    {
        // (haplotype, block of minimizers) work items: 24 whole haplotypes do not balance over 16+ threads
        const size_t BLK = 1 << 15;
        // Sp_R is sorted: a table over the top bits of the hash (about two keys per slot) replaces most of the binary search
        // (19 dependent cache misses per probe at 5 x 10^5 keys) by one table read and a search over a handful of keys
        int tb = 1;
        while (tb < 28 && ((size_t)1 << tb) < sp_hash.size() / 2) ++tb;
        std::vector<uint32_t> top(((size_t)1 << tb) + 1);
        {
            const int64_t nt = (int64_t)1 << tb;
#pragma omp parallel for num_threads(opt.threads) schedule(static)
            for (int64_t q = 0; q <= nt; ++q) {
                const uint64_t lo_key = q == nt ? ~(uint64_t)0 : (uint64_t)q << (64 - tb);
                top[q] = q == nt ? (uint32_t)sp_hash.size() : (uint32_t)(std::lower_bound(sp_hash.begin(), sp_hash.end(), lo_key) - sp_hash.begin());
            }
        }
        std::vector<std::pair<uint32_t, size_t>> items;
        for (uint32_t h = 0; h < num_walks; ++h)
            for (size_t m0 = 0; m0 < kmer_index[h].hash.size(); m0 += BLK) items.emplace_back(h, m0);
#pragma omp parallel for num_threads(opt.threads) schedule(dynamic, 1)
        for (int64_t it = 0; it < (int64_t)items.size(); ++it) {
            const uint32_t h = items[it].first;
            const auto &ix = kmer_index[h];
            const size_t m1 = std::min(ix.hash.size(), items[it].second + BLK);
            for (size_t m = items[it].second; m < m1; ++m) {
                const uint64_t key = ix.hash[m];
                const size_t slot = (size_t)(key >> (64 - tb));
                auto itp = std::lower_bound(sp_hash.begin() + top[slot], sp_hash.begin() + top[slot + 1], key);
                ids[h][m] = (itp != sp_hash.begin() + top[slot + 1] && *itp == key) ? (int32_t)(itp - sp_hash.begin()) : -1;
            }
        }
    }
    lap_a("dictionary lookup");
    // stable counting sort of the (h, m) sequence by id: one histogram per haplotype, offsets in (id, h) order,
    // every haplotype then scatters its own minimizers -- (h asc, minimizer order asc) inside every id
    std::vector<Raw> raw;
    if ((size_t)num_walks * (size_t)count_sp_r > ((size_t)1 << 29)) {   // histograms would not fit comfortably: serial sort
        for (uint32_t h = 0; h < num_walks; ++h)
            for (int32_t id : ids[h]) if (id >= 0) bucket_off[id + 1]++;
        for (int32_t r = 0; r < count_sp_r; ++r) bucket_off[r + 1] += bucket_off[r];
        raw.resize((size_t)bucket_off[count_sp_r]);
        std::vector<int64_t> fill(bucket_off.begin(), bucket_off.end() - 1);
        for (uint32_t h = 0; h < num_walks; ++h)
            for (size_t m = 0; m < ids[h].size(); ++m)
                if (ids[h][m] >= 0) raw[fill[ids[h][m]]++] = Raw{(int32_t)h, (uint32_t)m};
    } else {
        const size_t NS = (size_t)count_sp_r;
        std::vector<int32_t> cnt((size_t)num_walks * NS, 0);           // cnt[h][id]
#pragma omp parallel for num_threads(opt.threads) schedule(dynamic, 1)
        for (int32_t h = 0; h < (int32_t)num_walks; ++h) {
            int32_t *c = cnt.data() + (size_t)h * NS;
            for (int32_t id : ids[h]) if (id >= 0) c[id]++;
        }
        int64_t run = 0;
        for (size_t r = 0; r < NS; ++r) {                              // exclusive prefix in (id, h) order
            bucket_off[r] = run;
            for (uint32_t h = 0; h < num_walks; ++h) { int32_t &c = cnt[(size_t)h * NS + r]; const int32_t n = c; c = (int32_t)(run - bucket_off[r]); run += n; }
        }
        bucket_off[NS] = run;
        raw.resize((size_t)run);
#pragma omp parallel for num_threads(opt.threads) schedule(dynamic, 1)
        for (int32_t h = 0; h < (int32_t)num_walks; ++h) {
            int32_t *c = cnt.data() + (size_t)h * NS;                  // now: offset of (id, h) inside bucket id
            for (size_t m = 0; m < ids[h].size(); ++m) {
                const int32_t id = ids[h][m];
                if (id >= 0) raw[(size_t)(bucket_off[id] + c[id]++)] = Raw{h, (uint32_t)m};
            }
        }
    }
    lap_a("bucket by id");
    // ---- shared-anchor filter (:590-633) + occurrence sort (:641-663) ----
    occs.clear(); vpool.clear();
    const float thr = opt.threshold * num_walks;                       // float * uint32 -> float (:618)
    // ids are independent (the reference runs this loop under OpenMP too, :593): contiguous id chunks balanced by
    // occurrence count, each with private output, concatenated in id order afterwards
    const int n_chunks = std::max(1, opt.threads * 4);
    std::vector<int32_t> chunk_lo(n_chunks + 1, count_sp_r);
    chunk_lo[0] = 0;
    for (int c = 1; c < n_chunks; ++c) {
        const int64_t want = (int64_t)raw.size() * c / n_chunks;
        chunk_lo[c] = (int32_t)(std::lower_bound(bucket_off.begin(), bucket_off.end(), want) - bucket_off.begin());
        chunk_lo[c] = std::min(std::max(chunk_lo[c], chunk_lo[c - 1]), count_sp_r);
    }
    std::vector<std::vector<Occ>> occs_c(n_chunks);
    std::vector<std::vector<int32_t>> vpool_c(n_chunks);
#pragma omp parallel for num_threads(opt.threads) schedule(dynamic, 1)
    for (int c = 0; c < n_chunks; ++c) {
        std::string arena;                                             // keys "v0_v1_..._" back to back (:600-603)
        std::vector<uint32_t> koff;
        std::vector<int32_t> order, grp, byhap;
        auto key = [&](int32_t t) { return std::string_view(arena.data() + koff[t], koff[t + 1] - koff[t]); };
        auto &occs_l = occs_c[c];
        auto &vpool_l = vpool_c[c];
        for (int32_t r = chunk_lo[c]; r < chunk_lo[c + 1]; ++r) {
            const int64_t b = bucket_off[r], e = bucket_off[r + 1];
            if (b == e) continue;
            const int32_t n = (int32_t)(e - b);
            // The filter (:615-622) only asks whether some vertex path occurs >= thr times; equal keys <=> equal vertex
            // lists, so the lists themselves are grouped first (any order consistent with equality serves for counting).
            // Most ids are dropped here -- every haplotype carries the k-mer on the same path -- without a key being built.
            if ((float)n >= thr) {
                auto lst = [&](int32_t t) {
                    const Raw &o = raw[b + t];
                    const auto &ix = kmer_index[o.h];
                    return std::pair<const int32_t *, uint32_t>(ix.v.data() + ix.voff[o.m], ix.voff[o.m + 1] - ix.voff[o.m]);
                };
                bool dropped = false;
                const auto first = lst(0);
                int32_t same = 1;
                while (same < n) { const auto q = lst(same); if (q.second != first.second || !std::equal(q.first, q.first + q.second, first.first)) break; ++same; }
                if (same == n) dropped = true;                          // the common case: one path, n >= thr occurrences
                else {
                    order.resize(n);
                    std::iota(order.begin(), order.end(), 0);
                    std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                        const auto a = lst(x), c2 = lst(y);
                        if (a.second != c2.second) return a.second < c2.second;
                        return std::lexicographical_compare(a.first, a.first + a.second, c2.first, c2.first + c2.second);
                    });
                    for (int32_t i = 0; i < n && !dropped;) {
                        const auto a = lst(order[i]);
                        int32_t j = i + 1;
                        while (j < n) { const auto q = lst(order[j]); if (q.second != a.second || !std::equal(q.first, q.first + q.second, a.first)) break; ++j; }
                        if ((float)(j - i) >= thr) dropped = true;
                        i = j;
                    }
                }
                if (dropped) continue;                                  // :624-632 id dropped entirely
            }
            arena.clear();
            koff.assign(1, 0);
            for (int32_t t = 0; t < n; ++t) {
                const Raw &o = raw[b + t];
                const auto &ix = kmer_index[o.h];
                for (uint32_t q = ix.voff[o.m]; q < ix.voff[o.m + 1]; ++q) {
                    char buf[12];
                    int len = 0;
                    uint32_t x = (uint32_t)ix.v[q];                     // vertex ids are non-negative
                    do { buf[len++] = (char)('0' + x % 10); x /= 10; } while (x);
                    while (len) arena += buf[--len];
                    arena += '_';
                }
                koff.push_back((uint32_t)arena.size());
            }
            order.resize(n);
            std::iota(order.begin(), order.end(), 0);
            // std::map<std::string,...> iteration = lexicographic on the key; inside a key, push order
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return key(x) < key(y); });
            // Anchor_hits_1[r][h] in map-iteration order, then std::sort per (r,h) (:641-663)
            byhap = order;                                             // haplotype ascending, map-iteration order inside one
            std::stable_sort(byhap.begin(), byhap.end(), [&](int32_t x, int32_t y) { return raw[b + x].h < raw[b + y].h; });
            for (int32_t g0 = 0; g0 < n;) {
                const int32_t h = raw[b + byhap[g0]].h;
                int32_t g1 = g0;
                while (g1 < n && raw[b + byhap[g1]].h == h) ++g1;
                grp.assign(byhap.begin() + g0, byhap.begin() + g1);
                g0 = g1;
                const auto &ix = kmer_index[h];
                std::sort(grp.begin(), grp.end(), [&](int32_t x, int32_t y) {
                    const uint32_t mx = raw[b + x].m, my = raw[b + y].m;
                    const int32_t ax0 = ix.v[ix.voff[mx]], ay0 = ix.v[ix.voff[my]];
                    if (ax0 != ay0) return ax0 < ay0;
                    return ix.v[ix.voff[mx + 1] - 1] < ix.v[ix.voff[my + 1] - 1];
                });
                for (int32_t x : grp) {
                    const uint32_t m = raw[b + x].m;
                    Occ o{r, (int32_t)h, (uint32_t)vpool_l.size(), ix.voff[m + 1] - ix.voff[m]};
                    vpool_l.insert(vpool_l.end(), ix.v.begin() + ix.voff[m], ix.v.begin() + ix.voff[m + 1]);
                    occs_l.push_back(o);
                }
            }
        }
    }
    lap_a("filter + sort");
    {
        size_t no = 0, nv = 0;
        for (int c = 0; c < n_chunks; ++c) { no += occs_c[c].size(); nv += vpool_c[c].size(); }
        occs.reserve(no); vpool.reserve(nv);
        for (int c = 0; c < n_chunks; ++c) {
            const uint32_t base = (uint32_t)vpool.size();
            vpool.insert(vpool.end(), vpool_c[c].begin(), vpool_c[c].end());
            for (Occ o : occs_c[c]) { o.off += base; occs.push_back(o); }
        }
    }
    lap_a("concatenate");
    };
    if (dev_anchors) {
        dg_anchor_result ar;
        if (be.anchor_finish(be.ctx, sp_hash.data(), (int64_t)sp_hash.size(), opt.threshold * num_walks, &ar) != 0) {
            err = std::string("anchor_finish failed: ") + (be.last_error ? be.last_error() : "?"); return -1;
        }
        lap_a("device join+filter");
        hint_lattice();
        if (ar.n_unstable_groups > 0) {
            // a group whose order would hinge on std::sort's unstable partitioning (dg_anchor.hip): the host algorithm decides
            if (!opt.quiet) fprintf(stderr, "[dg::anchors] %lld occurrence group(s) need the host sort; redoing the stage on the host\n", (long long)ar.n_unstable_groups);
            dev_anchors = false;
            host_index();
            if (rc_sketch != 0) { err = err_sketch; return -1; }
            host_join();
        } else {
            occs.resize((size_t)ar.n_occ);
            for (int64_t i = 0; i < ar.n_occ; ++i) occs[i] = Occ{ar.occ_id[i], ar.occ_hap[i], ar.occ_off[i], ar.occ_len[i]};
            vpool.assign(ar.vpool, ar.vpool + ar.n_vtx);
        }
        for (void *q : {(void *)ar.occ_id, (void *)ar.occ_hap, (void *)ar.occ_off, (void *)ar.occ_len, (void *)ar.vpool}) if (q) be.free_buf(q);
    } else {
        host_join();
    }
    sum.anchors_per_hap.assign(num_walks, 0);
    for (auto &o : occs) sum.anchors_per_hap[o.h]++;
    if (!opt.quiet) {
        std::cerr << "Number of Anchors" << std::endl;                 // :674-685
        for (uint32_t h = 0; h < num_walks; ++h) fprintf(stderr, "%s : %d\n", hap_id2name[h].c_str(), (int)sum.anchors_per_hap[h]);
    }
    stamp("compute_anchors+filter+sort", t0);

    if (!opt.quiet) std::cout << "Classifying kmers..." << std::endl;  // :784
    return 0;
}

void Pipeline::wait_fit() {                                             // homo_bv is valid after this
    if (!fit_pending) return;
    fit_t0 = now_s();                                                  // the stage's time is what the caller waits here
    if (fit_thread.joinable()) fit_thread.join();
    fit_pending = false;
    const KGParams &P = sum.fit.P;
    if (!opt.quiet)
        fprintf(stderr, "[M::%s] Fitted model: best NLL=%.2f, u_v=%.2f (hom mean), sd_v=%.2f (hom SD), "
                "var_w=%.2f, p_d=%.2f, zp_copy=%.2f, zp_copy_het=%.2f, err_shape=%.2f, max_copy=%d\n",
                "compute_and_classify_anchors", sum.fit.nll, P.u_v, P.sd_v, P.var_w, P.p_d, P.zp_copy, P.zp_copy_het, P.err_shape, P.max_copy);
    if (!opt.quiet) {
        int64_t tot = std::max<int64_t>(1, count_sp_r);
        fprintf(stderr, "[M::%s] Phasing done. Homozygous: %.2f%%, Heterozygous: %.2f%%, Total kmers: %lld\n", "compute_and_classify_anchors",
                100.f * float(fit_n_hom) / tot, 100.f * float(count_sp_r - fit_n_hom) / tot, (long long)count_sp_r);
    }
    std::vector<int32_t>().swap(fit_sp_count);
    stamp("fit+classify (joined)", fit_t0);
}



// ======================================================================================
// ExpandedGraph  (ExpandedGraph.hpp:29-102, 269-409), flat CSR restatement
// ======================================================================================
void ExpandedGraph::permute(const uvec<int32_t> &order) {
    // new vertex i = old vertex order[i]; adjacency keeps its per-vertex order (ExpandedGraph.hpp:93-101, 392-400)
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    double tl = now_s();
    auto lap = [&](const char *w) { if (dbg) { double t = now_s(); fprintf(stderr, "[dg::permute] %-18s %.3f s\n", w, t - tl); tl = t; } };
    const int32_t nn = (int32_t)order.size();
    uvec<int32_t> new_idx(nn);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < nn; ++i) new_idx[order[i]] = i;
    uvec<int64_t> noff((size_t)nn + 1, 0);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < nn; ++i) noff[i + 1] = deg(order[i]);      // (random gathers in parallel, the running sum alone is cheap)
    for (int32_t i = 0; i < nn; ++i) noff[i + 1] += noff[i];
    lap("new_idx+noff");
    uvec<int32_t> ndst(adj_dst.size());
    uvec<uint8_t> nw(adj_w.size());
    lap("alloc");
    // the remap is a random gather (cache-miss bound): spread it over the host threads
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < nn; ++i) {
        int64_t o = noff[i];
        for (int64_t e = adj_off[order[i]]; e < adj_off[order[i] + 1]; ++e, ++o) { ndst[o] = new_idx[adj_dst[e]]; nw[o] = adj_w[e]; }
    }
    lap("edges");
    adj_off.swap(noff); adj_dst.swap(ndst); adj_w.swap(nw);
    uvec<int32_t> nh(nn);
    uvec<uint32_t> noo(nn), nol(nn);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < nn; ++i) { nh[i] = haplotype[order[i]]; noo[i] = orig_off[order[i]]; nol[i] = orig_len[order[i]]; }
    haplotype.swap(nh); orig_off.swap(noo); orig_len.swap(nol);
    if ((int32_t)level.size() == nn) {
        uvec<int32_t> nl(nn);
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < nn; ++i) nl[i] = level[order[i]];
        level.swap(nl);
    }
    lap("vertex arrays");
    uvec<int64_t> nco((size_t)nn + 1, 0);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < nn; ++i) nco[i + 1] = ncol(order[i]);
    for (int32_t i = 0; i < nn; ++i) nco[i + 1] += nco[i];
    uvec<int32_t> ncp(col_pool.size());
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < nn; ++i)
        std::copy(col_pool.begin() + col_off[order[i]], col_pool.begin() + col_off[order[i] + 1], ncp.begin() + nco[i]);
    col_off.swap(nco); col_pool.swap(ncp);
    lap("colours");
}

void ExpandedGraph::topologically_reorder(int sink) {                  // ExpandedGraph.hpp:29-102
    std::vector<int32_t> indeg(n, 0);
    const int64_t n_edges = (int64_t)adj_dst.size();
    if (n_edges < ((int64_t)1 << 26)) {                                // (MHC-24: 0.105 s serial, 0.123 s with atomics; 5 Mbp x 100 walks: 1.07 -> 0.89 s)
        for (int32_t d : adj_dst) ++indeg[d];
    } else {
#pragma omp parallel for schedule(static)
        for (int64_t e = 0; e < n_edges; ++e) {
#pragma omp atomic
            ++indeg[adj_dst[e]];
        }
    }
    uvec<int32_t> order;                                               // doubles as the FIFO queue
    order.reserve(n);
    for (int32_t v = 0; v < n; ++v) if (indeg[v] == 0 && v != sink) order.push_back(v);   // never push the sink now
    bool sink_ready = (indeg[sink] == 0);
    size_t head = 0;
    while (head < order.size() || sink_ready) {
        int u;
        if (head < order.size()) u = order[head++];                    // process the queue first
        else { u = sink; sink_ready = false; order.push_back(sink); ++head; }   // queue empty -> only the sink is left
        for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) {
            const int v = adj_dst[e];
            if (--indeg[v] == 0) { if (v == sink) sink_ready = true; else order.push_back(v); }
        }
    }
    if ((int32_t)order.size() != n) throw std::runtime_error("Graph contains a cycle; topological order impossible");
    permute(order);
}

int ExpandedGraph::strict_bfs_levelize_and_reorder() {                 // ExpandedGraph.hpp:269-409
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    double tl = now_s();
    auto lap = [&](const char *w) { if (dbg) { double t = now_s(); fprintf(stderr, "[dg::levelize] %-18s %.3f s\n", w, t - tl); tl = t; } };
    const int32_t n0 = n;
    if (n0 == 0) return 0;
    int source = -1;
    auto take_source = [&](int32_t v) {                                // ExpandedGraph.hpp:283-296: exactly one vertex without in-edges may have out-edges
        if (source == -1) source = v;
        else { std::cout << "Uh oh, multiple potential sources found while leveling\n"; std::exit(-1); }
    };
    // 1)-3) levels.  The reference seeds lvl with the BFS distance from the source, takes a Kahn order and relaxes
    // lvl[v] = max(lvl[v], lvl[u] + 1) along it (ExpandedGraph.hpp:300-352).  The fixed point is the longest-path distance
    // from the source whatever the seed (a vertex's BFS parent already forces lvl >= dist) and whichever topological order is
    // used; vertices without in-edges stay at 0.  After topologically_reorder every edge goes from a smaller to a larger id,
    // so the ids themselves are such an order: one pass, no queue, no BFS.  (Any other input takes the literal route.)
    std::vector<int32_t> lvl(n0, 0);
    bool sorted = true;
#pragma omp parallel for schedule(static) reduction(&& : sorted)
    for (int32_t u = 0; u < n0; ++u)
        for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) sorted = sorted && adj_dst[e] > u;
    if (getenv("DG_LEVELIZE_LITERAL")) sorted = false;                // (tests: the literal BFS + Kahn + relaxation route must give the same levels)
    if (sorted) {
        for (int32_t u = 0; u < n0; ++u) {
            const int32_t lu = lvl[u] + 1;
            for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) { int32_t &lv = lvl[adj_dst[e]]; if (lv < lu) lv = lu; }
        }
        for (int32_t v = 0; v < n0; ++v) if (lvl[v] == 0 && deg(v) > 0) take_source(v);      // level 0 <=> no in-edge
        if (source < 0) throw std::runtime_error("bad source index");
        lap("levels (one pass)");
    } else {
        std::vector<int32_t> indeg(n0, 0);
        for (int32_t d : adj_dst) ++indeg[d];
        for (int32_t v = 0; v < n0; ++v) if (indeg[v] == 0 && deg(v) > 0) take_source(v);
        if (source < 0 || source >= n0) throw std::runtime_error("bad source index");
        std::vector<int32_t> dist(n0, -1), q;                          // 1) BFS from the source
        q.reserve(n0);
        dist[source] = 0; q.push_back(source);
        for (size_t h = 0; h < q.size(); ++h) {
            const int u = q[h];
            for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) { const int v = adj_dst[e]; if (dist[v] == -1) { dist[v] = dist[u] + 1; q.push_back(v); } }
        }
        lap("indeg+bfs");
        std::vector<int32_t> topo;                                     // 2) Kahn over ALL indeg-0 vertices
        topo.reserve(n0);
        for (int32_t v = 0; v < n0; ++v) if (indeg[v] == 0) topo.push_back(v);
        for (size_t h = 0; h < topo.size(); ++h) {
            const int u = topo[h];
            for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) if (--indeg[adj_dst[e]] == 0) topo.push_back(adj_dst[e]);
        }
        if ((int32_t)topo.size() != n0) throw std::runtime_error("Graph contains a cycle; strict leveling requires a DAG");
        for (int32_t v = 0; v < n0; ++v) if (dist[v] >= 0) lvl[v] = dist[v];         // 3) seed / relax
        for (int u : topo) for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) { const int v = adj_dst[e]; if (lvl[v] <= lvl[u]) lvl[v] = lvl[u] + 1; }
    }

    lap("kahn+relax");
    // 4) dummies for skipped levels: edge (u,v,w) with gap g becomes u -w-> d1 -0-> ... -0-> dg -0-> v; dummy ids are
    //    handed out in (u ascending, edge order) sequence (prefix sum, so vertices can be processed in parallel), each
    //    inherits haplotype[u] and u's original-vertex list.
    std::vector<int64_t> dbase((size_t)n0 + 1, 0);                     // dummies created before vertex u's edges
#pragma omp parallel for schedule(static)
    for (int32_t u = 0; u < n0; ++u) {
        int64_t c = 0;
        for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) { const int gap = lvl[adj_dst[e]] - lvl[u] - 1; if (gap > 0) c += gap; }
        dbase[u + 1] = c;
    }
    for (int32_t u = 0; u < n0; ++u) dbase[u + 1] += dbase[u];
    const int64_t n_dummy = dbase[n0];
    const int64_t n1l = (int64_t)n0 + n_dummy;
    if (n1l >= INT32_MAX) throw std::runtime_error("expanded graph too large");
    const int32_t n1 = (int32_t)n1l;
    uvec<int32_t> lv(n1), hp2(n1);
    uvec<uint32_t> oo(n1), ol(n1);
    uvec<int64_t> noff((size_t)n1 + 1, 0);
    uvec<int32_t> ndst((size_t)adj_dst.size() + (size_t)n_dummy);
    uvec<uint8_t> nw(ndst.size());
    // old vertices keep their out-degree and edge slots; dummy d (id n0 + d) owns the single slot E + d
    const int64_t E0 = (int64_t)adj_dst.size();
#pragma omp parallel for schedule(static)
    for (int32_t u = 0; u <= n0; ++u) noff[u] = adj_off[u];
#pragma omp parallel for schedule(static)
    for (int32_t d = n0 + 1; d <= n1; ++d) noff[d] = E0 + (d - n0);
#pragma omp parallel for schedule(static)
    for (int32_t u = 0; u < n0; ++u) {
        lv[u] = lvl[u]; hp2[u] = haplotype[u]; oo[u] = orig_off[u]; ol[u] = orig_len[u];
        int32_t next_dummy = n0 + (int32_t)dbase[u];
        for (int64_t e = adj_off[u]; e < adj_off[u + 1]; ++e) {
            const int v = adj_dst[e], w = adj_w[e];
            const int gap = lvl[v] - lvl[u] - 1;
            if (gap <= 0) { ndst[e] = v; nw[e] = (uint8_t)w; continue; }
            int64_t slot = e;                                           // where the next hop is written
            for (int step = 1; step <= gap; ++step) {
                const int32_t dmy = next_dummy++;
                lv[dmy] = lvl[u] + step; hp2[dmy] = haplotype[u]; oo[dmy] = orig_off[u]; ol[dmy] = orig_len[u];
                ndst[slot] = dmy; nw[slot] = (uint8_t)(step == 1 ? w : 0);
                slot = E0 + (dmy - n0);
            }
            ndst[slot] = v; nw[slot] = 0;
        }
    }
    adj_off.swap(noff); adj_dst.swap(ndst); adj_w.swap(nw);
    haplotype.swap(hp2); orig_off.swap(oo); orig_len.swap(ol); level.swap(lv);
    {
        uvec<int64_t> nco((size_t)n1 + 1);                             // dummies have no colour
        for (int32_t v = 0; v <= n0; ++v) nco[v] = col_off[v];
        for (int32_t v = n0 + 1; v <= n1; ++v) nco[v] = col_off[n0];
        col_off.swap(nco);
    }
    n = n1;

    lap("dummies");
    // 5) order by (level, id): stable, so a counting sort by level -- in parallel: every thread owns a contiguous range of ids,
    //    counts its vertices per level, and scatters them behind the counts of the threads before it
    int max_level = 0;
#pragma omp parallel for schedule(static) reduction(max : max_level)
    for (int32_t v = 0; v < n1; ++v) if (level[v] > max_level) max_level = level[v];
    const int NT = std::max(1, std::min(omp_get_max_threads(), 32));
    const size_t NL = (size_t)max_level + 1;
    std::vector<int32_t> hist((size_t)NT * NL, 0);
    auto v_lo = [&](int t) { return (int32_t)((int64_t)n1 * t / NT); };
#pragma omp parallel for num_threads(NT) schedule(static, 1)
    for (int t = 0; t < NT; ++t) {
        int32_t *h = hist.data() + (size_t)t * NL;
        for (int32_t v = v_lo(t); v < v_lo(t + 1); ++v) ++h[level[v]];
    }
    level_off.assign(max_level + 2, 0);
    int max_width = 0;
    for (size_t l = 0; l < NL; ++l) {                                   // per level: width, and each thread's first slot
        int32_t run = level_off[l];
        for (int t = 0; t < NT; ++t) { const int32_t c = hist[(size_t)t * NL + l]; hist[(size_t)t * NL + l] = run; run += c; }
        level_off[l + 1] = run;
        max_width = std::max(max_width, run - level_off[l]);
    }
    uvec<int32_t> order(n1);
#pragma omp parallel for num_threads(NT) schedule(static, 1)
    for (int t = 0; t < NT; ++t) {
        int32_t *fill = hist.data() + (size_t)t * NL;
        for (int32_t v = v_lo(t); v < v_lo(t + 1); ++v) order[fill[level[v]]++] = v;
    }
    lap("sort");
    permute(order);
    lap("permute");
    return max_width;
}

// ======================================================================================
// DpGraphStorage
// ======================================================================================
dg_dp_graph DpGraphStorage::view(int R) const {
    dg_dp_graph g;
    g.n_vertices = (int32_t)(out_off.size() - 1);
    g.n_levels = (int32_t)(level_off.size() - 1);
    g.R = R;
    g.level_off = level_off.data();
    g.out_off = out_off.data(); g.out_dst = out_dst.data(); g.out_w = out_w.data();
    g.hom_off = hom_off.data(); g.het_off = het_off.data();
    g.hom_col = hom_col.data(); g.het_col = het_col.data();
    return g;
}

namespace {
template <class V> void wr(std::ofstream &f, const V &v) {
    using T = typename V::value_type;
    uint64_t n = v.size();
    f.write((const char *)&n, 8);
    f.write((const char *)v.data(), (std::streamsize)(n * sizeof(T)));
}
template <class V> bool rd(std::ifstream &f, V &v) {
    using T = typename V::value_type;
    uint64_t n = 0;
    if (!f.read((char *)&n, 8)) return false;
    v.resize(n);
    return (bool)f.read((char *)v.data(), (std::streamsize)(n * sizeof(T)));
}
}  // namespace

// file = "DGDP0001" | int32 R | 8 length-prefixed arrays (level_off,out_off,out_dst,out_w,hom_off,hom_col,het_off,het_col)
bool DpGraphStorage::save(const std::string &path, int R) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f.write("DGDP0001", 8);
    int32_t r = R;
    f.write((const char *)&r, 4);
    wr(f, level_off); wr(f, out_off); wr(f, out_dst); wr(f, out_w);
    wr(f, hom_off); wr(f, hom_col); wr(f, het_off); wr(f, het_col);
    return (bool)f;
}
bool DpGraphStorage::load(const std::string &path, int &R) {
    std::ifstream f(path, std::ios::binary);
    char magic[8];
    if (!f || !f.read(magic, 8) || memcmp(magic, "DGDP0001", 8) != 0) return false;
    int32_t r;
    if (!f.read((char *)&r, 4)) return false;
    R = r;
    return rd(f, level_off) && rd(f, out_off) && rd(f, out_dst) && rd(f, out_w) &&
           rd(f, hom_off) && rd(f, hom_col) && rd(f, het_off) && rd(f, het_col);
}

// ======================================================================================
// haploid DP  (approximator.cpp:44-168) -- CPU by design
// ======================================================================================
// Haploid (vertex, r) tables of approximator.cpp:44-72 in gather form (the host twin of dg_dp_solve_haploid): vertices are
// in topological order, so dp[u][.] is final before any successor of u is visited and
//   dp[v][r2] = max(0, max over in-edges (u, w) of dp[u][r2 - w] + |color[v]|).
// The reference's scatter loop only replaces on a strictly larger value (:60), i.e. the first candidate in its visiting
// order (u ascending, source r ascending = weight-1 edge before weight-0 edge of the same u, adjacency order) keeps a
// tie; the in-edge lists below are built in exactly that order.  Every state starts at 0 with back pointers -1 (:50-52).
namespace {
struct HapTables {
    int RP = 0;
    std::vector<int32_t> dp, back_vtx, back_r;                         // [v * RP + r]
    size_t at(int v, int r) const { return (size_t)v * RP + r; }
};

void haploid_tables_host(const ExpandedGraph &g, int R, HapTables &T) {
    const int n = g.n, RP = R + 1;
    std::vector<int64_t> in_off((size_t)n + 1, 0);
    for (int64_t e = 0; e < g.adj_off[n]; ++e) in_off[g.adj_dst[e] + 1]++;
    for (int v = 0; v < n; ++v) in_off[v + 1] += in_off[v];
    std::vector<uint32_t> in_src((size_t)g.adj_off[n]);                // source | weight << 31
    {
        std::vector<int64_t> fill(in_off.begin(), in_off.end() - 1);
        for (int u = 0; u < n; ++u)
            for (int64_t e = g.adj_off[u]; e < g.adj_off[u + 1]; ++e) in_src[fill[g.adj_dst[e]]++] = (uint32_t)u | ((uint32_t)g.adj_w[e] << 31);
    }
    for (int v = 0; v < n; ++v) {                                      // a source's weight-1 edges first, stably
        uint32_t *a = in_src.data() + in_off[v];
        const int64_t d = in_off[v + 1] - in_off[v];
        for (int64_t i = 0; i < d;) {
            int64_t j = i;
            while (j < d && (a[j] & 0x7FFFFFFFu) == (a[i] & 0x7FFFFFFFu)) ++j;
            if (j - i > 1) std::stable_partition(a + i, a + j, [](uint32_t x) { return (x >> 31) != 0; });
            i = j;
        }
    }
    T.RP = RP;
    T.dp.assign((size_t)n * RP, 0); T.back_vtx.assign((size_t)n * RP, -1); T.back_r.assign((size_t)n * RP, -1);
    for (int v = 0; v < n; ++v) {
        const int gain = (int)g.ncol(v);
        for (int r2 = 0; r2 <= R; ++r2) {
            int best = 0, from = -1, from_r = -1;
            for (int64_t e = in_off[v]; e < in_off[v + 1]; ++e) {
                const int u = (int)(in_src[e] & 0x7FFFFFFFu), r = r2 - (int)(in_src[e] >> 31);
                if (r < 0) continue;
                const int cand = T.dp[T.at(u, r)] + gain;
                if (cand > best) { best = cand; from = u; from_r = r; }
            }
            T.dp[T.at(v, r2)] = best; T.back_vtx[T.at(v, r2)] = from; T.back_r[T.at(v, r2)] = from_r;
        }
    }
}

// visits the vertices of the path that ends in (sink, r), sink first (:83-101, :141-153)
template <class F> void walk_back(const HapTables &T, int sink, int r, F &&visit) {
    for (int v = sink; v != -1;) {
        visit(v);
        const size_t o = T.at(v, r);
        v = T.back_vtx[o];
        r = T.back_r[o];
    }
}
}  // namespace

std::vector<int> Pipeline::haploid_dp(const ExpandedGraph &g, int R, std::string &err) {
    const bool dbg_h = getenv("DG_DEBUG") != nullptr;
    double th0 = now_s();
    const int n = g.n;
    HapTables T;
    // Device or host?  The tables are a chain of dependent levels (longest-path depth); on the device one workgroup walks
    // it at ~0.7 us per level whatever its width, the host gather loop costs ~5 ns per (in-edge, r).  Graphs of this
    // pipeline are a few vertices wide (MHC_4: 499 k vertices on 250 k levels: device 0.21 s, host 0.045 s), so `auto`
    // goes to the device only when a level holds enough vertices to pay for its barrier.
    bool on_device = be.dp_solve_haploid && opt.haploid_mode != 1;
    if (on_device && opt.haploid_mode == 0) {
        std::vector<int32_t> depth(n, 0);
        int32_t deepest = 0;
        for (int u = 0; u < n; ++u) {
            for (int64_t e = g.adj_off[u]; e < g.adj_off[u + 1]; ++e) depth[g.adj_dst[e]] = std::max(depth[g.adj_dst[e]], depth[u] + 1);
            deepest = std::max(deepest, depth[u]);
        }
        on_device = (double)g.adj_off[n] * (R + 1) / (double)(deepest + 1) >= 140.0 * 16;   // (in-edge, r) items per level vs 0.7 us of 16 host threads
    }
    if (on_device) {                                                   // the device loop
        std::vector<int32_t> ncol(n);
        for (int v = 0; v < n; ++v) ncol[v] = (int32_t)g.ncol(v);
        dg_hap_graph hg{n, R, g.adj_off.data(), g.adj_dst.data(), g.adj_w.data(), ncol.data()};
        T.RP = R + 1;
        T.dp.resize((size_t)n * T.RP); T.back_vtx.resize(T.dp.size()); T.back_r.resize(T.dp.size());
        if (be.dp_solve_haploid(be.ctx, &hg, T.dp.data(), T.back_vtx.data(), T.back_r.data()) != 0) {
            err = std::string("dp_solve_haploid failed: ") + (be.last_error ? be.last_error() : "?");
            return {};
        }
    } else {
        haploid_tables_host(g, R, T);
    }
    if (dbg_h) fprintf(stderr, "[dg::haploid] (vertex, r) tables %.3f s\n", now_s() - th0);
    th0 = now_s();
    // :74-113.  Per recombination count: number of distinct colours on its path, and (certificate line) their mean
    // occurrence count.  The reference fills an unordered_set and a std::map per r; flat counters do, the R + 1 walks
    // run in parallel.
    int32_t max_col = -1;
    for (int32_t c : g.col_pool) max_col = std::max(max_col, c);
    std::vector<int> colors_by_r(R + 1, 0);
    std::vector<float> avg_by_r(R + 1, 0.f);
#pragma omp parallel for schedule(dynamic, 1) num_threads(opt.threads)
    for (int r = 0; r <= R; r++) {
        std::vector<int32_t> cnt((size_t)max_col + 1, 0);
        int distinct = 0;
        walk_back(T, n - 1, r, [&](int v) { for (int64_t q = g.col_off[v]; q < g.col_off[v + 1]; ++q) distinct += (cnt[g.col_pool[q]]++ == 0); });
        colors_by_r[r] = distinct;
        float total = 0;                                               // :106-111: float sum in ascending colour order
        for (int32_t c = 0; c <= max_col; ++c) if (cnt[c]) total += cnt[c];
        avg_by_r[r] = total / distinct;                                // 0/0 -> nan, as the reference prints it
    }
    if (dbg_h) fprintf(stderr, "[dg::haploid] per-r backtracks %.3f s\n", now_s() - th0);
    if (!opt.quiet)
        for (int i = 0; i < R; ++i) std::cout << "Approximation ratio certificate: " << avg_by_r[i] << std::endl;
    // :116-136  the first r whose gain in distinct colours, as an angle against the largest gain, falls below 5 degrees
    // (double arithmetic; a 0/0 slope is NaN, compares false and falls through to r = 0 like the reference)
    double steepest = 0;
    for (int r = 0; r < R; ++r) {
        if (!opt.quiet) std::cout << "r: " << r << " true score: " << colors_by_r[r] << std::endl;
        steepest = std::max(steepest, (double)std::abs(colors_by_r[r + 1] - colors_by_r[r]));
    }
    int best_r = 0;
    for (int r = 0; r < R; ++r) {
        const int gain = colors_by_r[r + 1] - colors_by_r[r];
        const double deg = std::atan(static_cast<double>(gain) / steepest) * 180.0 / M_PI;
        if (!opt.quiet)
            std::cout << "r: " << r << " -> " << r + 1 << ", \xCE\x94" "colors: " << gain << ", angle: " << deg << "\xC2\xB0" << std::endl;
        if (deg < 5 /* HAP_ANGLE_THRESHOLD */) { best_r = r; break; }
    }
    if (!opt.quiet) std::cerr << "Recombination count: " << best_r << std::endl;
    sum.best_r_haploid = best_r;
    std::vector<int> path;                                             // :141-153
    walk_back(T, n - 1, best_r, [&](int v) { path.push_back(v); });
    std::vector<int> out;                                              // original vertices, source to sink, first occurrence only (:30-40)
    std::unordered_set<int> seen;
    for (auto it = path.rbegin(); it != path.rend(); ++it)
        for (uint32_t q = 0; q < g.orig_len[*it]; ++q) {
            const int uo = g.orig_pool[g.orig_off[*it] + q];
            if (seen.insert(uo).second) out.push_back(uo);
        }
    return out;
}

// ======================================================================================
// Approximator::solve  (approximator.cpp:1014-1331)
// ======================================================================================
int Pipeline::solve(std::string &err) {
    double t0 = now_s();
    if (opt.ploidy != 2 || getenv("DG_GRAPH_LITERAL")) wait_fit();
    if (opt.ploidy == 2 && !getenv("DG_GRAPH_LITERAL")) {
        // the fused route (fast_graph.cpp) covers everything up to the levelized graph; it declines inputs it does not model
        // (empty walks, several sources, a vertex deeper than the sink ...), which then take the literal route below
        // (heap objects: a process that is about to exit -- the CLI -- skips their teardown, ~0.07 s of munmap and 2 x 10^6 small
        // destructors on MHC-24; Options::leak_at_exit)
        ExpandedGraph *gf = new ExpandedGraph();
        auto *anchorsByHapF = new std::vector<std::vector<AnchorRec>>();
        std::vector<uint8_t> color_homo_bv_f;
        auto drop = [&]() { if (!opt.leak_at_exit) { delete gf; delete anchorsByHapF; } };
        if (build_levelized_fast(*gf, *anchorsByHapF, color_homo_bv_f)) {
            stamp("levelized_graph_build", t0);
            int rc = diploid(*gf, color_homo_bv_f, *anchorsByHapF, err);
            drop();
            if (rc != 0) return rc;
            if (!opt.quiet) std::cout << "Diploid sequences written to: " << opt.hap_file << std::endl;   // :1330
            return 0;
        }
        delete gf; delete anchorsByHapF;
        wait_fit();
        sum.n_colours = 0;
        t0 = now_s();
    }
    int32_t number_of_vertices = 0;
    for (size_t h = 0; h < paths.size(); h++) number_of_vertices += (int32_t)paths[h].size();
    const int H = (int)paths.size();
    // Adjacency is recorded as one global push log; a stable counting sort by source gives the CSR with the
    // reference's per-vertex push order (chain edge, weight-1 edges, start->super edges, overlap edges).
    struct ELog { int32_t src, dst; uint8_t w; };
    std::vector<ELog> elog;
    elog.reserve((size_t)number_of_vertices * 2 + 1024);
    ExpandedGraph g;
    int32_t nvert = 2 + number_of_vertices;                            // :1022
    g.haplotype.assign(nvert, 0);                                      // :1025 (source and sink keep 0)
    g.orig_off.assign(nvert, 0);
    g.orig_len.assign(nvert, 0);
    g.orig_pool.reserve((size_t)number_of_vertices + vpool.size());
    std::vector<int32_t> v2e((size_t)n_vtx * H, -1);                   // vertex_to_expanded_map[v][h]  (:1023)

    const int sink = nvert - 1;
    int32_t current_vertex = 1;
    for (int h = 0; h < H; h++) {                                      // :1029-1049
        elog.push_back({0, current_vertex, 0});
        for (size_t i = 0; i < paths[h].size(); i++) {
            v2e[(size_t)paths[h][i] * H + h] = current_vertex;         // last occurrence wins (:1035)
            g.orig_off[current_vertex] = (uint32_t)g.orig_pool.size();
            g.orig_len[current_vertex] = 1;
            g.orig_pool.push_back((int32_t)paths[h][i]);
            g.haplotype[current_vertex] = h;
            if (i < paths[h].size() - 1) elog.push_back({current_vertex, current_vertex + 1, 0});
            else elog.push_back({current_vertex, sink, 0});
            current_vertex++;
        }
    }
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    double tl = now_s();
    auto lap = [&](const char *w) { if (dbg) { double t = now_s(); fprintf(stderr, "[dg::build] %-18s %.3f s\n", w, t - tl); tl = t; } };
    lap("chains");
    // recombination edges (:1051-1095)
    std::vector<int64_t> wslot_off((size_t)n_vtx + 1, 0);              // vertex_w_uv[u][j] flattened
    for (size_t u = 0; u < adj_list.size(); u++) wslot_off[u + 1] = wslot_off[u] + (int64_t)adj_list[u].size();
    std::vector<int32_t> vertex_w_uv((size_t)wslot_off[n_vtx], -1);
    std::vector<uint8_t> w_filled;                                     // "adjacency of w_uv is non-empty" (:1082)
    for (int h = 0; h < H; h++) {
        for (size_t i = 0; i < paths[h].size(); i++) {
            const int u = (int)paths[h][i];
            for (size_t j = 0; j < adj_list[u].size(); j++) {
                const int v = (int)adj_list[u][j];
                if (i == paths[h].size() - 1 || v != (int)paths[h][i + 1]) {
                    int32_t &wv = vertex_w_uv[wslot_off[u] + (int64_t)j];
                    if (wv == -1) {
                        wv = nvert++;
                        g.haplotype.push_back(-1);
                        g.orig_off.push_back(0);
                        g.orig_len.push_back(0);
                        w_filled.push_back(0);
                    }
                    elog.push_back({v2e[(size_t)u * H + h], wv, 1});
                    uint8_t &filled = w_filled[wv - (number_of_vertices + 2)];
                    if (!filled)
                        for (int hh = 0; hh < H; ++hh) {
                            const int32_t v_e = v2e[(size_t)v * H + hh];
                            if (v_e >= 0) { elog.push_back({wv, v_e, 0}); filled = 1; }
                        }
                }
            }
        }
    }
    { std::vector<int32_t>().swap(vertex_w_uv); }

    lap("recomb edges");
    // anchors -> AnchorRec per haplotype (:1114-1176)
    std::vector<std::vector<AnchorRec>> anchorsByHap(paths.size());
    std::vector<int32_t> color_to_anchor;
    int nextID = nvert;
    int colourID = 0;
    {
        std::vector<size_t> cnt(H, 0);
        for (const Occ &o : occs) cnt[o.h]++;
        for (int h = 0; h < H; ++h) anchorsByHap[h].reserve(cnt[h]);
        size_t p = 0;
        while (p < occs.size()) {                                      // ids without occurrences use no colour
            const int32_t a = occs[p].a;
            for (; p < occs.size() && occs[p].a == a; ++p) {           // occs sorted by (a, h, occurrence order)
                const Occ &o = occs[p];
                const int h = o.h;
                const int startOrig = vpool[o.off], endOrig = vpool[o.off + o.len - 1];
                const int startExp = v2e[(size_t)startOrig * H + h], endExp = v2e[(size_t)endOrig * H + h];
                int nodeID;
                if (startExp == endExp) {
                    nodeID = startExp;
                } else {
                    elog.push_back({startExp, nextID, 0});             // :1148
                    elog.push_back({nextID, endExp, 0});               // :1149
                    g.orig_off.push_back((uint32_t)g.orig_pool.size());
                    g.orig_len.push_back(o.len);
                    g.orig_pool.insert(g.orig_pool.end(), vpool.begin() + o.off, vpool.begin() + o.off + o.len);
                    g.haplotype.push_back(-1);
                    nodeID = nextID++;
                }
                anchorsByHap[h].push_back({startOrig, endOrig, startExp, endExp, {colourID}, nodeID});
            }
            color_to_anchor.push_back(a);
            colourID++;
        }
    }
    nvert = nextID;
    const int n_colours = colourID;
    sum.n_colours = n_colours;
    { std::vector<int32_t>().swap(v2e); }

    lap("anchor recs");
    // per-haplotype sweep: overlap edges + containment colour propagation (:1193-1246)
    // Haplotypes are independent here (anchor records, node ids and stacks are per haplotype); the overlap edges
    // each one produces are appended to the push log afterwards in haplotype order, as the serial loop would.
    std::vector<std::vector<ELog>> ov_edges(paths.size());
    std::vector<std::vector<std::pair<int32_t, int32_t>>> colpairs_h(paths.size());   // (nodeID, colour)
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t hh = 0; hh < (int64_t)paths.size(); ++hh) {
        const size_t h = (size_t)hh;
        auto &vec = anchorsByHap[h];
        if (vec.empty()) continue;
        std::sort(vec.begin(), vec.end(), [](const AnchorRec &a, const AnchorRec &b) {
            if (a.startExp != b.startExp) return a.startExp < b.startExp;
            else return a.endExp < b.endExp;
        });
        std::vector<AnchorRec *> stk;
        for (auto &anc : vec) {
            while (!stk.empty() && stk.back()->endExp < anc.startExp) stk.pop_back();
            if (!stk.empty() && anc.startExp <= stk.back()->endExp && stk.back()->nodeID != anc.nodeID)
                ov_edges[h].push_back({stk.back()->nodeID, anc.nodeID, 0});
            for (int i = (int)stk.size() - 1; i >= 0; --i) {
                if (anc.endExp <= stk[i]->endExp) {
                    for (int c : anc.colours)
                        if (std::find(stk[i]->colours.begin(), stk[i]->colours.end(), c) == stk[i]->colours.end())
                            stk[i]->colours.push_back(c);
                } else break;
            }
            stk.push_back(&anc);
        }
        auto &cp = colpairs_h[h];
        for (const auto &anc : vec)                                    // :1240-1245: per node, sorted-unique union
            for (int c : anc.colours) cp.emplace_back(anc.nodeID, c);
        std::sort(cp.begin(), cp.end());
        cp.erase(std::unique(cp.begin(), cp.end()), cp.end());
    }
    lap("sweep");
    for (auto &ve : ov_edges) elog.insert(elog.end(), ve.begin(), ve.end());
    { std::vector<std::vector<ELog>>().swap(ov_edges); }
    // node ids of different haplotypes are disjoint, so the per-haplotype sorted lists only need a count + scatter
    std::vector<std::pair<int32_t, int32_t>> colpairs;
    {
        size_t tot = 0;
        for (auto &cp : colpairs_h) tot += cp.size();
        colpairs.reserve(tot);
        for (auto &cp : colpairs_h) colpairs.insert(colpairs.end(), cp.begin(), cp.end());
    }
    { std::vector<std::vector<std::pair<int32_t, int32_t>>>().swap(colpairs_h); }

    // assemble the flat graph
    g.n = nvert;
    g.adj_off.assign((size_t)nvert + 1, 0);
    for (const ELog &e : elog) g.adj_off[e.src + 1]++;
    for (int32_t v = 0; v < nvert; ++v) g.adj_off[v + 1] += g.adj_off[v];
    g.adj_dst.resize(elog.size());
    g.adj_w.resize(elog.size());
    {
        // stable scatter by source, in parallel: every thread owns a contiguous range of sources (balanced by edge
        // count) and reads the whole push log in order, so a vertex keeps its push order as in the serial loop
        std::vector<int64_t> fill(g.adj_off.begin(), g.adj_off.end() - 1);
        const int T = std::max(1, std::min(opt.threads, 32));
        std::vector<int32_t> cut(T + 1, nvert);
        cut[0] = 0;
        for (int t = 1; t < T; ++t) {
            const int64_t want = (int64_t)elog.size() * t / T;
            cut[t] = (int32_t)(std::lower_bound(g.adj_off.begin(), g.adj_off.end(), want) - g.adj_off.begin());
            cut[t] = std::min(std::max(cut[t], cut[t - 1]), nvert);
        }
#pragma omp parallel for num_threads(T) schedule(static, 1)
        for (int t = 0; t < T; ++t) {
            const int32_t lo = cut[t], hi = cut[t + 1];
            if (lo >= hi) continue;
            for (const ELog &e : elog)
                if (e.src >= lo && e.src < hi) { const int64_t o = fill[e.src]++; g.adj_dst[o] = e.dst; g.adj_w[o] = e.w; }
        }
    }
    { std::vector<ELog>().swap(elog); }
    g.col_off.assign((size_t)nvert + 1, 0);
    for (auto &pc : colpairs) g.col_off[pc.first + 1]++;
    for (int32_t v = 0; v < nvert; ++v) g.col_off[v + 1] += g.col_off[v];
    g.col_pool.resize(colpairs.size());
    {
        std::vector<int64_t> fill(g.col_off.begin(), g.col_off.end() - 1);
        for (auto &pc : colpairs) g.col_pool[fill[pc.first]++] = pc.second;   // a node's colours arrive ascending
    }
    { std::vector<std::pair<int32_t, int32_t>>().swap(colpairs); }
    lap("assemble");
    stamp("expanded_graph_build", t0);
    t0 = now_s();
    g.topologically_reorder(sink);                                     // :1256
    stamp("topologically_reorder", t0);

    if (opt.ploidy == 1) {                                             // :1260-1278
        t0 = now_s();
        std::vector<int> dp_path = haploid_dp(g, opt.R, err);
        if (!err.empty()) return -1;
        std::string out;
        for (auto u : dp_path) out += node_seq[u];
        std::ofstream f(opt.hap_file, std::ios::out);
        if (!f.is_open()) { err = "cannot open output file " + opt.hap_file; return -1; }
        f << ">" << "dp_sol" << " LN:" << out.size() << std::endl;
        for (size_t i = 0; i < out.size(); i += 80) f << out.substr(i, 80) << std::endl;
        f.close();
        if (!f.good()) { err = "write to " + opt.hap_file + " failed"; return -1; }
        sum.len1 = (int64_t)out.size();
        stamp("haploid_dp+write", t0);
    } else {
        std::vector<uint8_t> color_homo_bv(n_colours, 0);              // :1283-1290
        for (int c = 0; c < n_colours; ++c) if (homo_bv[color_to_anchor[c]]) color_homo_bv[c] = 1;
        t0 = now_s();
        g.strict_bfs_levelize_and_reorder();                           // :1302
        stamp("strict_levelize", t0);
        int rc = diploid(g, color_homo_bv, anchorsByHap, err);
        if (rc != 0) return rc;
    }
    if (!opt.quiet) std::cout << "Diploid sequences written to: " << opt.hap_file << std::endl;   // :1330
    return 0;
}

// ======================================================================================
// diploid_dp_approximation_solver minus the level loop  (approximator.cpp:362-453, 720-1011)
// ======================================================================================
int Pipeline::diploid(ExpandedGraph &g, const std::vector<uint8_t> &color_homo_bv,
                      const std::vector<std::vector<AnchorRec>> &anchorsByHap, std::string &err) {
    double t0 = now_s();
    const int L = (int)g.level_off.size() - 1;
    const int nV = g.n;
    if (be.hint_dp_soon && !g.colours_split) {                         // level widths are final: the exact lattice size (the fused route has said so already)
        double cells = 0;
        for (int l = 1; l < L; ++l) { const double kw = (double)(g.level_off[l + 1] - g.level_off[l]); cells += kw * kw; }
        be.hint_dp_soon(be.ctx, (int64_t)std::min(9.0e18, cells * (opt.R + 1)));
    }
    if (!opt.quiet && g.level_off[1] - g.level_off[0] > 1) std::cout << "There is more than one source on level zero!" << std::endl;
    // the flat graph already is the dg_dp_graph layout (vertex ids are level-sorted: ExpandedGraph.hpp:360-407)
    // Topology arrays are handed to the device library in place (no copies); only the HOM / HET colour CSR (:431-453,
    // lists are sorted-unique already) is new: counts, prefix sums, fill -- in parallel over vertex blocks.
    dpg = DpGraphStorage();
    if (!opt.quiet) std::cout << "Creating hetro/hom-zygous colors per vertex lists" << std::endl;
    if (g.colours_split) {                                             // the fused route wrote the split lists directly
        dpg.hom_off.swap(g.hom_off); dpg.het_off.swap(g.het_off); dpg.hom_col.swap(g.hom_col); dpg.het_col.swap(g.het_col);
    } else {
    dpg.hom_off.assign((size_t)nV + 1, 0);
    dpg.het_off.assign((size_t)nV + 1, 0);
    for (int c : g.col_pool) (void)color_homo_bv.at(c);                // same out_of_range behaviour as the reference's .at()
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nV; ++v) {
        int64_t nh = 0;
        for (int64_t q = g.col_off[v]; q < g.col_off[v + 1]; ++q) nh += color_homo_bv[g.col_pool[q]] == 1;
        dpg.hom_off[v + 1] = nh;
        dpg.het_off[v + 1] = (g.col_off[v + 1] - g.col_off[v]) - nh;
    }
    for (int v = 0; v < nV; ++v) { dpg.hom_off[v + 1] += dpg.hom_off[v]; dpg.het_off[v + 1] += dpg.het_off[v]; }
    dpg.hom_col.resize((size_t)dpg.hom_off[nV]);
    dpg.het_col.resize((size_t)dpg.het_off[nV]);
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nV; ++v) {
        int64_t ph = dpg.hom_off[v], pt = dpg.het_off[v];
        for (int64_t q = g.col_off[v]; q < g.col_off[v + 1]; ++q) {
            const int c = g.col_pool[q];
            if (color_homo_bv[c] == 1) dpg.hom_col[ph++] = c; else dpg.het_col[pt++] = c;
        }
    }
    }
    sum.n_levels = L;
    sum.n_vertices = nV;
    stamp("dp_prologue_flatten", t0);
    if (!opt.dump_prefix.empty()) {                                    // the dump wants the topology too
        dpg.level_off = g.level_off; dpg.out_off = g.adj_off; dpg.out_dst = g.adj_dst; dpg.out_w = g.adj_w;
        dpg.save(opt.dump_prefix + ".dpg", opt.R);
        uvec<int32_t>().swap(dpg.level_off); uvec<int64_t>().swap(dpg.out_off);
        uvec<int32_t>().swap(dpg.out_dst); uvec<uint8_t>().swap(dpg.out_w);
    }
    if (opt.dump_only) { err = "dump_only"; return 1; }

    // ---- the level loop + sink read-out: DEVICE (approximator.cpp:532-716, 774-785) ----
    t0 = now_s();
    if (!opt.quiet) std::cout << "Running DP" << std::endl;
    const int R = opt.R;
    const int cap = R + 8;
    std::vector<int32_t> p1f(cap), p1t(cap), p2f(cap), p2t(cap);
    dg_dp_graph view = dpg.view(R);
    view.n_vertices = nV; view.n_levels = L;
    view.level_off = g.level_off.data(); view.out_off = g.adj_off.data(); view.out_dst = g.adj_dst.data(); view.out_w = g.adj_w.data();
    dg_dp_result res;
    memset(&res, 0, sizeof(res));
    res.p1_from = p1f.data(); res.p1_to = p1t.data(); res.p2_from = p2f.data(); res.p2_to = p2t.data();
    res.cap = cap;
    int rc = be.dp_solve_diploid(be.ctx, &view, &res);
    if (rc != 0) { err = std::string("dp_solve_diploid failed: ") + (be.last_error ? be.last_error() : "?"); return -1; }
    stamp("dp_level_loop", t0);
    t0 = now_s();
    sum.dp_value = res.value; sum.s_het = res.s_het; sum.cells = res.cells; sum.relaxations = res.relaxations;
    if (!opt.quiet) std::cout << "DP value: " << res.value << std::endl;       // :776
    std::vector<std::pair<int, int>> wp1, wp2;
    for (int i = 0; i < res.n_p1 && i < cap; ++i) wp1.emplace_back(p1f[i], p1t[i]);
    for (int i = 0; i < res.n_p2 && i < cap; ++i) wp2.emplace_back(p2f[i], p2t[i]);
    const int r1 = (int)wp1.size() - 1, r2 = (int)wp2.size() - 1;              // :784-785

    auto find_next_zero_hap = [&](int src, int target_hap) -> int {          // :732-755
        if (g.haplotype.at(src) == target_hap && g.orig_len.at(src) > 0) return src;
        std::queue<int> q;
        std::unordered_set<int> visited;
        q.push(src); visited.insert(src);
        while (!q.empty()) {
            int u = q.front(); q.pop();
            for (int64_t e = g.adj_off[u]; e < g.adj_off[u + 1]; ++e) {
                const int v = g.adj_dst[e];
                if (g.adj_w[e] != 0) continue;
                if (!visited.insert(v).second) continue;
                if (g.haplotype.at(v) == target_hap && g.orig_len.at(v) > 0) return v;
                q.push(v);
            }
        }
        return -1;
    };

    std::unordered_map<int, int> p_color_freq[2];
    std::vector<int> p_colors[2];
    std::string hap_seq[2];
    for (int which = 0; which < 2; ++which) {                                 // :790-923
        const auto &wedges = which == 0 ? wp1 : wp2;
        const char *tag = which == 0 ? "P1" : "P2";
        std::string &hs = hap_seq[which];
        const int first_vertex = g.level_off.at(0);                   // vertices_in_level[0][0]
        int start_exp = first_vertex;
        for (int i = 0; i < (int)wedges.size(); i++) {
            const auto &edge = wedges.at(i);
            if (g.orig_len[edge.first] != 1) {
                std::cout << tag << ": Vertex " << edge.first << " in map back has " << g.orig_len[edge.first]
                          << " original vertices" << std::endl;
                exit(1);
            }
            int end_exp = edge.first;
            int h = g.haplotype.at(end_exp);
            if (start_exp == first_vertex)
                for (int v = g.level_off.at(1); v < g.level_off.at(2); ++v) if (g.haplotype.at(v) == h) start_exp = v;
            if (g.orig_len.at(start_exp) < 1 || g.orig_len.at(end_exp) < 1) throw std::out_of_range("original_vertex.at(0)");
            int start_org = g.orig_pool[g.orig_off[start_exp]];
            int end_org = g.orig_pool[g.orig_off[end_exp]];
            bool activated = false;
            for (int t = 0; t < (int)paths[h].size(); t++) {
                if ((int)paths[h][t] == start_org) activated = true;
                if (activated) hs += node_seq[paths[h][t]];
                if ((int)paths[h][t] == end_org) { activated = false; break; }
            }
            for (const auto &a : anchorsByHap[h])
                if (a.startOrg > start_org && a.endOrg < end_org)
                    for (auto c : a.colours) {
                        if (p_color_freq[which].find(c) == p_color_freq[which].end()) { p_color_freq[which][c] = 1; p_colors[which].push_back(c); }
                        else p_color_freq[which][c] += 1;
                    }
            if (g.haplotype.at(edge.second), edge.second >= g.level_off[L - 1]) break;   // level[edge.second] == L - 1 (ids are level-sorted)
            const auto &next_edge = wedges.at(i + 1);
            int next_hap = g.haplotype.at(next_edge.first);
            int next_start = find_next_zero_hap(edge.second, next_hap);
            if (next_start != -1) start_exp = next_start;
            else std::cout << tag << " (path recovery) Could not find next_hap=" << next_hap << " from " << edge.second << " via 0-weight edges\n";
        }
    }
    sum.r1 = r1; sum.r2 = r2;
    sum.len1 = (int64_t)hap_seq[0].size(); sum.len2 = (int64_t)hap_seq[1].size();

    {   // score + approximation certificate (:933-1004) -- stdout only
        auto split = [&](const std::vector<int> &cs, std::vector<int> &hom, std::vector<int> &het) {
            for (auto c : cs) { if (color_homo_bv[c]) hom.push_back(c); else het.push_back(c); }
            std::sort(hom.begin(), hom.end()); hom.erase(std::unique(hom.begin(), hom.end()), hom.end());
            std::sort(het.begin(), het.end()); het.erase(std::unique(het.begin(), het.end()), het.end());
        };
        std::vector<int> h1, t1, h2, t2, inter, symd;
        split(p_colors[0], h1, t1); split(p_colors[1], h2, t2);
        std::set_intersection(h1.begin(), h1.end(), h2.begin(), h2.end(), std::back_inserter(inter));
        std::set_symmetric_difference(t1.begin(), t1.end(), t2.begin(), t2.end(), std::back_inserter(symd));
        int intersection_count = (int)inter.size(), symdiff_count = (int)symd.size();
        int m_G_hom = 0, m_G_het = 0;
        auto freq = [&](int which, int c) { auto it = p_color_freq[which].find(c); return it == p_color_freq[which].end() ? 0 : it->second; };
        for (auto c : inter) { int k1 = freq(0, c), k2 = freq(1, c); m_G_hom += (k1 >= k2 ? k1 : k2); }
        for (auto c : symd) m_G_het += freq(0, c) + freq(1, c);
        float m_G_hom_avg = m_G_hom / (float)intersection_count;
        float m_G_het_avg = m_G_het / (float)symdiff_count;
        float m_bar = std::max(m_G_hom_avg, m_G_het_avg);
        int loss_het = res.s_het - m_G_het;
        float additive_term = loss_het / (float)m_G_het_avg;
        int obj = intersection_count + symdiff_count;
        sum.obj = obj;
        if (!opt.quiet) {
            std::cout << "r: " << R << " obj: " << obj << std::endl;
            float ub = m_bar * (obj + additive_term);
            std::cout << "Approximation certificate: multiplicative factor: " << ub / (float)obj << std::endl;
        }
    }
    if (!opt.quiet)
        std::cout << "recombinations in P1: " << r1 << ", recombinations in P2: " << r2 << ", bp of P1: " << hap_seq[0].length()
                  << ", bp of P2: " << hap_seq[1].length() << std::endl;                 // :1307-1308
    {
        // :1314-1325 -- same bytes (80 columns, '\n' line ends), assembled in memory and written once instead of one
        // flushed line at a time
        std::string text;
        text.reserve(hap_seq[0].size() + hap_seq[1].size() + (hap_seq[0].size() + hap_seq[1].size()) / 80 + 128);
        for (int q = 0; q < 2; ++q) {
            text += q == 0 ? ">sol_1 bp:" : ">sol_2 bp:";
            text += std::to_string(hap_seq[q].size());
            text += '\n';
            for (size_t i = 0; i < hap_seq[q].size(); i += 80) { text.append(hap_seq[q], i, 80); text += '\n'; }
        }
        std::ofstream f(opt.hap_file, std::ios::out | std::ios::binary);
        if (!f.is_open()) { err = "cannot open output file " + opt.hap_file; return -1; }
        f.write(text.data(), (std::streamsize)text.size());
        f.close();
        if (!f.good()) { err = "write to " + opt.hap_file + " failed"; return -1; }
    }
    stamp("traceback+write", t0);
    return 0;
}

// "id hap v0,v1,..." per occurrence in Anchor_hits order (id asc, hap asc, occurrence order), then "homo id" per set bit of
// homo_bv: the format oracle/ref_harness.cpp dumps from the reference's own Solver object (tests/golden/anchors.json)
bool Pipeline::dump_anchors(const std::string &path) const {
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return false;
    for (const Occ &o : occs) {
        fprintf(f, "%d %d ", o.a, o.h);
        for (uint32_t q = 0; q < o.len; ++q) fprintf(f, "%s%d", q ? "," : "", vpool[o.off + q]);
        fputc('\n', f);
    }
    for (size_t id = 0; id < homo_bv.size(); ++id) if (homo_bv[id]) fprintf(f, "homo %zu\n", id);
    return fclose(f) == 0;
}

int Pipeline::run(std::string &err) {                                  // main.cpp:117-165
    sum = Summary();
    opt.threads = std::max(1, opt.threads);                            // -t0 / negative: every num_threads clause below sees a valid count
#ifdef _OPENMP
    omp_set_num_threads(opt.threads);
#endif
    t_run0 = now_s();
    // the reads file is parsed on a thread of its own beside the GFA (neither needs the other)
    std::thread reads_thread;
    std::string reads_err;
    int reads_rc = 0;
    double reads_dt = 0;
    const bool want_reads = !spectrum_injected && (opt.ploidy == 1 || opt.ploidy == 2);
    if (want_reads) reads_thread = std::thread([&] { const double t = now_s(); reads.clear(); reads_rc = read_sequences(opt.reads_file, reads, reads_err) ? 0 : -1; reads_dt = now_s() - t; });
    const int grc = load_graph(err);
    if (reads_thread.joinable()) reads_thread.join();
    if (grc) return -1;
    if (want_reads) {
        if (reads_rc) { err = reads_err; return -1; }
        sum.stage_s.emplace_back("read_ip_reads (beside the GFA)", reads_dt);
        if (!opt.quiet) fprintf(stderr, "[dg::stage] %-28s %.3f s\n", "read_ip_reads (beside GFA)", reads_dt);
        reads_loaded = true;
    }
    return run_loaded(err);
}

std::string Pipeline::haplotype_sequence(uint32_t h) const {           // solver.cpp:283-288
    std::string hap;
    size_t tot = 0;
    for (uint32_t v : paths.at(h)) tot += node_seq[v].size();
    hap.reserve(tot);
    for (uint32_t v : paths[h]) hap += node_seq[v];
    return hap;
}

int Pipeline::run_loaded(std::string &err) {                           // main.cpp:163-165
    opt.threads = std::max(1, opt.threads);
#ifdef _OPENMP
    omp_set_num_threads(opt.threads);
#endif
    const double t0 = t_run0 > 0 ? t_run0 : now_s();
    if (opt.ploidy != 1 && opt.ploidy != 2) {
        std::cout << "Current approximator support is only for ploidy = 1 or ploidy = 2" << std::endl;
        return 0;
    }
    if (!spectrum_injected && !reads_loaded && load_reads(err)) return -1;
    if (compute_and_classify_anchors(err)) return -1;
    if (!opt.anchor_dump.empty()) {
        wait_fit();
        if (!dump_anchors(opt.anchor_dump)) { err = "cannot write " + opt.anchor_dump; return -1; }
        if (opt.dump_only && opt.dump_prefix.empty()) { err = "dump_only"; return -1; }
    }
    if (solve(err)) return -1;
    sum.stage_s.emplace_back("total", now_s() - t0);
    return 0;
}

}  // namespace dg
