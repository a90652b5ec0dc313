// dg_dp_run and the DP entry points of the C ABI: issues the level chain (plain launches or replayed hipGraph batches),
// drives lattice chunks / segments (checkpoint + recompute when the back-pointer lattice outgrows HBM), the score-delta
// windows and the traceback, and turns the edge records into the two weighted-edge lists (approximator.cpp:757-785).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "dg_dp.hpp"

namespace dgi {

constexpr int DELTA_NO_PIECES = 1 << 30;

double wall_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

void graphs_clear(DpState &S) {                 // captured level batches: stale as soon as the graph, the lattice or an option changes
    for (auto &kv : S.graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    S.graphs.clear();
    S.graph_hist.clear();
}

void dp_state_free(DpState *s) {
    if (!s) return;
    graphs_clear(*s);
    sweep_prefetch_free(*s);
    delta_overlap_free(*s);
    { std::unique_lock<std::mutex> lk(s->pool.mu); s->pool.target = 0; }
    s->pool.cv.notify_all();
    if (s->pool.th.joinable()) s->pool.th.join();
    const double tf0 = wall_s();
    for (void *q : s->pool.chunks) (void)hipFree(q);
    if (getenv("DG_DEBUG")) fprintf(stderr, "[dipgenie_hip] destroy: %zu lattice chunks freed in %.3f s\n", s->pool.chunks.size(), wall_s() - tf0);
    for (auto &e : s->ev) if (e) (void)hipEventDestroy(e);
    delete s;
}

// ---------------------------------------------------------------------------------------------
// lattice chunk pool
// ---------------------------------------------------------------------------------------------
// Ask for `target` chunks (the latest request wins): starts the allocation thread if chunks are missing.  Chunks
// already mapped beyond the target are kept (hipFree of 8 GB costs ~0.1 s) unless pool_trim is called.  Returns at once.
void pool_request(DpState &S, int device, size_t target) {
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
                // (a lowered target must get through a pause: pool_trim joins this thread while dg_dp_load_graph holds the pause)
                Sp->pool.cv.wait(lk2, [&] { return !Sp->pool.paused || Sp->pool.chunks.size() >= Sp->pool.target; });
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
PoolPause::PoolPause(DpState &s) : S(s) { std::unique_lock<std::mutex> lk(S.pool.mu); S.pool.paused = true; }
PoolPause::~PoolPause() { { std::unique_lock<std::mutex> lk(S.pool.mu); S.pool.paused = false; } S.pool.cv.notify_all(); }
// free chunks beyond `keep` (and stop asking for more than that)
void pool_trim(DpState &S, size_t keep) {
    { std::unique_lock<std::mutex> lk(S.pool.mu); S.pool.target = std::min(std::min(S.pool.target, keep), S.pool.chunks.size()); }
    S.pool.cv.notify_all();
    if (S.pool.th.joinable()) S.pool.th.join();               // it stops at the next chunk boundary (a paused one: at once)
    std::unique_lock<std::mutex> lk(S.pool.mu);
    while (S.pool.chunks.size() > keep) { (void)hipFree(S.pool.chunks.back()); S.pool.chunks.pop_back(); }
    S.pool.running = false;
}
// wait until chunk c exists; nullptr if the allocation failed
void *pool_wait(DpState &S, size_t c) {
    std::unique_lock<std::mutex> lk(S.pool.mu);
    S.pool.cv.wait(lk, [&] { return S.pool.chunks.size() > c || S.pool.failed || !S.pool.running; });
    return S.pool.chunks.size() > c ? S.pool.chunks[c] : nullptr;
}

// ---------------------------------------------------------------------------------------------
// one DP pass over the resident graph
// ---------------------------------------------------------------------------------------------
namespace {

struct Run {
    dg_ctx *c;
    DpState &S;
    hipStream_t s;
    SweepLaunch X;
    int64_t n_launch = 0;
    int next_warm = 0;                                  // next level at which the look-ahead launch is due
    double host_enqueue_s = 0;                          // host time spent issuing the sweep's launches (DG_DEBUG)
    std::vector<uint16_t *> pool_base;

    Run(dg_ctx *c_, DpState &S_) : c(c_), S(S_), s(c_->stream) { sweep_prepare(S, X); }
    int n_win() const { return (int)S.dwin_t.size() - 1; }
    int32_t *state_ptr(int level) const { return (int32_t *)(S.d_ring.as<char>() + (size_t)(level & 1) * S.state_alloc_bytes) + S.pad_front; }
    size_t level_cells(int level) const {               // state size of a level (level 0: the source, k = 1)
        const int64_t k = level == 0 ? 1 : S.descs[level].k2;
        return (size_t)(k * k * S.RP);
    }
    void load_window(int w) { X.A.delta = X.F.delta = delta_launch_window(S, w, s); }

    // issues the launches of destination levels [l0, l1) of the range that began at lb
    int issue_levels(int l0, int l1, int lb, int le) {
        for (int l = l0; l < l1; ++l) {
            if (S.level_win[l] >= 0 && S.level_win[l] != S.cur_win) load_window(S.level_win[l]);
            if (S.warm_ahead > 0 && l >= next_warm && (l == lb || !(S.pf_active && S.pf_far > 0))) {
                // tables of the batch after this one (and, at the start of a range, of this one too)
                const int q0 = l == lb ? l : (int)std::min<int64_t>(l + S.warm_ahead, le), q1 = (int)std::min<int64_t>(l + 2 * S.warm_ahead, le);
                if (q1 > q0) sweep_warm_tables(S, X, q0, q1, s);
            }
            if (l >= next_warm) next_warm = l + (int)std::max<int64_t>(S.warm_ahead, 1);
            sweep_launch_level(S, X, l, s);
            ++n_launch;
            // profiling aid: rocprofv3 --pmc crashes when ~10^5 dispatches are queued without a drain
            if (S.sync_every > 0 && n_launch % S.sync_every == 0) DG_HIP(hipStreamSynchronize(s));
        }
        return DG_OK;
    }

    // Sweeps destination levels [lb, le).  bp_biased = lattice pointer minus the offset of level lb's first cell
    // (so the kernels keep using the global LevelDesc::bp_off), or nullptr for a value-only pass.
    // Issuing a level costs the host 2.3-5.4 us (hipLaunchKernelGGL; it varies from run to run on a shared host), the GPU
    // 2.7-4.2 us: since the sweep's own period came down to ~4.1 us the chain is host-bound as often as not (MHC-24 sweeps of
    // 596-742 ms with plain launches, pass after pass of one process).  So the levels are captured into hipGraphs of 1,000
    // launches and replayed: 590.5-593 ms on MHC-24, the capturing pass included, 320 ms on MHC_4 (option graph_batch: 0 =
    // plain launches, which at their best are 2 % faster).
    // A stream that cannot be captured (e.g. a caller-provided legacy stream) or a failed instantiation switches the
    // context back to plain launches for good; the batch at hand is then issued again, plainly.
    int sweep_range(int lb, int le, uint16_t *bp_biased) {
        X.A.bp = bp_biased; X.F.bp = bp_biased;
        next_warm = lb;
        if (int rc = sweep_prefetch_begin(S, X, lb, le, n_win() == 1, s)) return rc;
        const int64_t gb = S.graph_batch >= 0 ? S.graph_batch : 1000;
        // what issue_levels bakes into a captured batch besides the levels: whether the periodic look-ahead launches are left to the
        // prefetcher's far blocks -- part of the cache key (a second DP state on the device switches the prefetcher off)
        const int pf_key = ((S.pf_active && S.pf_far > 0) ? 1 : 0) | (S.rp_active << 1);
        for (int l0 = lb; l0 < le;) {
            const bool use_graph = gb > 0 && n_win() == 1 && S.sync_every == 0 && !S.graph_failed;
            const int l1 = use_graph ? (int)std::min<int64_t>((int64_t)l0 + gb, le) : le;
            // score deltas computed beside the sweep (delta_launch_overlapped): the stream waits for the pieces that hold a level below l1
            while (S.delta_piece_next < (int)S.delta_piece_level.size() && S.delta_piece_level[S.delta_piece_next] < l1) {
                DG_HIP(hipStreamWaitEvent(s, S.delta_piece_ev[S.delta_piece_next], 0));
                ++S.delta_piece_next;
            }
            hipGraphExec_t *slot = nullptr;
            bool capturing = false;
            if (use_graph) {
                const auto key = std::make_tuple(l0, l1, (const void *)bp_biased, pf_key);
                slot = &S.graphs[key];
                if (*slot) {
                    DG_HIP(hipGraphLaunch(*slot, s));
                    const std::vector<int64_t> &h = S.graph_hist[key];
                    for (size_t q = 0; q < h.size(); ++q) S.launch_hist[q] += h[q];
                    n_launch += l1 - l0; l0 = l1;
                    continue;
                }
                if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
                else { (void)hipGetLastError(); S.graph_failed = true; continue; }
            }
            const int64_t n_launch_before = n_launch;
            std::vector<int64_t> hist_before(S.launch_hist, S.launch_hist + 64 * 4);
            if (int rc = issue_levels(l0, l1, lb, le)) {
                if (capturing) {                                    // leave the stream usable: end the capture, drop the half-built graph
                    hipGraph_t cg = nullptr;
                    (void)hipStreamEndCapture(s, &cg);
                    if (cg) (void)hipGraphDestroy(cg);
                    (void)hipGetLastError();
                    S.graphs.erase(std::make_tuple(l0, l1, (const void *)bp_biased, pf_key));
                }
                return rc;
            }
            if (capturing) {
                hipGraph_t cg = nullptr;
                const bool ok = hipStreamEndCapture(s, &cg) == hipSuccess && cg && hipGraphInstantiate(slot, cg, nullptr, nullptr, 0) == hipSuccess;
                if (cg) (void)hipGraphDestroy(cg);
                if (!ok) {                                          // nothing of this batch has run: issue it again without a graph
                    (void)hipGetLastError();
                    *slot = nullptr; S.graph_failed = true; n_launch = n_launch_before;
                    std::copy(hist_before.begin(), hist_before.end(), S.launch_hist);
                    continue;
                }
                std::vector<int64_t> &h = S.graph_hist[std::make_tuple(l0, l1, (const void *)bp_biased, pf_key)];
                h.resize(64 * 4);
                for (size_t q = 0; q < h.size(); ++q) h[q] = S.launch_hist[q] - hist_before[q];
                DG_HIP(hipGraphLaunch(*slot, s));
            }
            l0 = l1;
        }
        sweep_prefetch_end(S, le, s);
        return DG_OK;
    }

    // Launches issued while the pool thread is still mapping chunks would each queue behind a multi-GB hipMalloc,
    // so there is nothing to overlap: wait for every chunk this run uses first.
    int wait_for_chunks() {
        if (S.d_bp.p) return DG_OK;
        const double tw0 = wall_s();
        const size_t need = (size_t)std::min(S.seg_chunks, (int)S.chunk_begin.size() - 1);
        if (!pool_wait(S, need - 1)) { set_error("back-pointer lattice: hipMalloc of a %.1f GB chunk failed", S.pool.chunk_units * 2 / 1e9); return DG_ERR_OOM; }
        std::unique_lock<std::mutex> lk(S.pool.mu);
        for (size_t q = 0; q < need; ++q) pool_base.push_back((uint16_t *)S.pool.chunks[q]);
        if (getenv("DG_DEBUG")) fprintf(stderr, "[dipgenie_hip] run: waited %.3f s for lattice chunks\n", wall_s() - tw0);
        return DG_OK;
    }

    // Sweeps the chunks [c0, c1) with back-pointers into pool chunks 0 .. c1-c0-1 (or the one exact buffer), then walks
    // them last first; from_sink = this is the walk that starts at the sink.
    int sweep_and_walk(int c0, int c1, bool from_sink, bool mark_forward_end) {
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
        if (S.test_poison_level > 0 && S.test_poison_level < S.L)        // tests: a level nobody swept / a damaged lattice
            for (int ch = c0; ch < c1; ++ch) {
                const int lb = S.d_bp.p ? 1 : S.chunk_begin[ch], le = S.d_bp.p ? S.L : S.chunk_begin[ch + 1], lp = (int)S.test_poison_level;
                if (lp >= lb && lp < le) DG_HIP(hipMemsetAsync(biased[ch - c0] + S.descs[lp].bp_off, (int)S.test_poison_byte, 2 * (size_t)S.level_units[lp], s));
            }
        for (int ch = c1 - 1; ch >= c0; --ch) {
            const int lb = S.d_bp.p ? 1 : S.chunk_begin[ch], le = S.d_bp.p ? S.L : S.chunk_begin[ch + 1];
            trace_launch_warm_rows(S, lb, le, s);
            trace_launch_chain(S, le - 1, lb, biased[ch - c0], from_sink && ch == c1 - 1 ? state_ptr(S.L - 1) : (const int32_t *)nullptr, s);
        }
        return DG_OK;
    }

    int forward_and_trace() {
        const int n_seg = (int)S.seg_begin.size() - 1, n_chunks_all = (int)S.chunk_begin.size() - 1;
        DG_HIP(hipEventRecord(S.ev[0], s));
        S.cur_win = -1;
        S.delta_piece_next = DELTA_NO_PIECES;
        if (n_win() == 1 && S.n_delta_blocks > 0) X.A.delta = X.F.delta = delta_launch_overlapped(S, s);   // everything fits: the head up front (delta_ms), the rest beside the sweep
        DG_HIP(hipEventRecord(S.ev[1], s));
        if (S.want_digest) DG_HIP(hipMemsetAsync(S.d_digest.p, 0, 8 * (size_t)S.L, s));
#ifdef DG_SWEEP_PROBE
        {   // slot 0 of every level takes an atomicMin: start from all ones
            std::vector<unsigned long long> init((size_t)S.L * 8, 0ULL);
            for (int l = 0; l < S.L; ++l) init[(size_t)l * 8] = ~0ULL;
            DG_HIP(hipMemcpy(S.d_probe.p, init.data(), 8 * init.size(), hipMemcpyHostToDevice));
        }
#endif
        sweep_init_state(S, s);
        if (n_seg == 1) {
            // whole lattice resident: one sweep with back-pointers, then the chain walk chunk by chunk, last first
            if (int rc = sweep_and_walk(0, S.d_bp.p ? 1 : n_chunks_all, true, true)) return rc;
        } else {
            // pass 1: values only, keeping the state in front of every segment
            const int64_t dig = S.want_digest;
            int64_t planes_swept = 0;
            S.rp_active = S.RP;
            for (int sg = 0; sg < n_seg; ++sg) {
                if (sg > 0)
                    DG_HIP(hipMemcpyAsync(S.d_ckpt.as<int32_t>() + S.ckpt_off[sg], state_ptr(S.seg_begin[sg] - 1),
                                          4 * level_cells(S.seg_begin[sg] - 1), hipMemcpyDeviceToDevice, s));
                if (int rc = sweep_range(S.seg_begin[sg], S.seg_begin[sg + 1], nullptr)) return rc;
            }
            DG_HIP(hipEventRecord(S.ev[2], s));                 // (the re-sweeps below are booked under traceback_ms)
            // pass 2: last segment first -- restore its input state, re-sweep its chunks with back-pointers, walk them
            S.want_digest = 0;                                  // digests were accumulated in pass 1
            // Along a path the recombination count only grows, and a cell of plane r gathers from planes r, r - 1, r - 2 of the level
            // before: once the walk has left segment sg on plane r*, the cells it can meet in the segments before lie on planes
            // <= r*, and those depend on planes <= r* only.  Every earlier segment is therefore re-swept up to the plane its
            // successor's walk ended on (one 16-byte read per segment) -- the path uses its recombinations along the whole panel, so
            // about half of the second pass goes away.  (The finish kernel re-scores the walked path against the DP value as ever.)
            for (int sg = n_seg - 1; sg >= 0; --sg) {
                const int lb = S.seg_begin[sg];
                if (sg > 0)
                    DG_HIP(hipMemcpyAsync(state_ptr(lb - 1), S.d_ckpt.as<int32_t>() + S.ckpt_off[sg], 4 * level_cells(lb - 1), hipMemcpyDeviceToDevice, s));
                else
                    sweep_init_state(S, s);
                const int c0 = sg * S.seg_chunks, c1 = std::min(n_chunks_all, c0 + S.seg_chunks);
                if (int rc = sweep_and_walk(c0, c1, sg == n_seg - 1, false)) { S.want_digest = dig; S.rp_active = S.RP; return rc; }
                if (S.plane_limit && sg > 0) {
                    ChainState cs{0, 0, S.RP - 1, 0};
                    if (hipMemcpyAsync(&cs, S.d_chain.p, sizeof cs, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { S.want_digest = dig; S.rp_active = S.RP; DG_HIP(hipGetLastError()); return DG_ERR_HIP; }
                    S.rp_active = (cs.value != CHAIN_CORRUPT && cs.r >= 0 && cs.r < S.RP) ? cs.r + 1 : S.RP;
                    planes_swept += (int64_t)S.rp_active * (S.seg_begin[sg] - S.seg_begin[sg - 1]);
                }
            }
            S.rp_active = S.RP;
            S.want_digest = dig;
            if (getenv("DG_DEBUG") && S.plane_limit) fprintf(stderr, "[dipgenie_hip] run: second pass swept %.1f %% of the (level, plane) pairs before the last segment\n",
                                                              100.0 * (double)planes_swept / std::max(1.0, (double)S.RP * (S.seg_begin[n_seg - 1] - 1)));
        }
        trace_launch_finish(S, s);
        DG_HIP(hipEventRecord(S.ev[3], s));
        DG_HIP(hipGetLastError());
        return DG_OK;
    }
};

}  // namespace

static int dp_run(dg_ctx *c, dg_dp_result *res) {
    DpState *Sp = c->dp;
    if (!Sp || !Sp->loaded) { set_error("dg_dp_run: no graph loaded"); return DG_ERR_STATE; }
    if (!res) { set_error("dg_dp_run: null result"); return DG_ERR_ARG; }
    DpState &S = *Sp;
    Run run(c, S);
    hipStream_t s = c->stream;
    if (int rc = run.wait_for_chunks()) return rc;
    TraceOut to;
    std::vector<int32_t> edges(4 * (size_t)S.cap);
    memset(S.launch_hist, 0, sizeof S.launch_hist);
    run.n_launch = 0;
    if (int rc = run.forward_and_trace()) return rc;
    DG_HIP(hipMemcpyAsync(&to, S.d_trace.p, sizeof to, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(edges.data(), S.d_edges.p, 4 * edges.size(), hipMemcpyDeviceToHost, s));
    if (S.want_digest) {
        S.digest_host.assign(S.L, 0);
        DG_HIP(hipMemcpyAsync(S.digest_host.data(), S.d_digest.p, 8 * (size_t)S.L, hipMemcpyDeviceToHost, s));
    }
    DG_HIP(hipStreamSynchronize(s));
#ifdef DG_SWEEP_PROBE
    if (const char *po = getenv("DG_PROBE_OUT")) {
        std::vector<unsigned long long> pr((size_t)S.L * 8);
        DG_HIP(hipMemcpy(pr.data(), S.d_probe.p, 8 * pr.size(), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(po, "wb")) { fwrite(pr.data(), 8, pr.size(), f); fclose(f); }
    }
#endif
    if (getenv("DG_DEBUG") && S.lean_chain) trace_debug_report(S);
    if (getenv("DG_DEBUG") && S.pf_stream) {
        int w[4] = {0, 0, 0, 0};
        if (hipStreamSynchronize(S.pf_stream) == hipSuccess && hipMemcpy(w, S.d_pfctl.p, sizeof w, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[dipgenie_hip] run: L2 table prefetcher covered %d levels (since load)\n", w[2]);
    }
    if (getenv("DG_DEBUG"))
        fprintf(stderr, "[dipgenie_hip] run: host issued %lld sweep launches in %.1f ms (%.2f us each)\n", (long long)run.n_launch, 1e3 * run.host_enqueue_s,
                1e6 * run.host_enqueue_s / (double)std::max<int64_t>(run.n_launch, 1));
    DG_HIP(hipEventElapsedTime(&S.timing.delta_ms, S.ev[0], S.ev[1]));
    DG_HIP(hipEventElapsedTime(&S.timing.forward_ms, S.ev[1], S.ev[2]));
    DG_HIP(hipEventElapsedTime(&S.timing.traceback_ms, S.ev[2], S.ev[3]));
    DG_HIP(hipEventElapsedTime(&S.timing.total_ms, S.ev[0], S.ev[3]));
    S.timing.n_forward_launches = run.n_launch;
    S.timing.n_segments = (int32_t)S.seg_begin.size() - 1;
    S.timing.n_chunks = (int32_t)S.chunk_begin.size() - 1;
    if (to.value == CHAIN_CORRUPT || to.corrupt) { set_error("back-pointer lattice is corrupt: the chain walk left its level (a level was not swept?)"); return DG_ERR_STATE; }
    if (to.value != NEG_INF && to.path_score != to.value) {
        set_error("traceback path scores %d but the DP value is %d: sweep, lattice and walk disagree", to.path_score, to.value); return DG_ERR_STATE;
    }
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
    if (!c->dp) c->dp = new dgi::DpState(c->device);
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
extern "C" int dg_dp_get_table_digest(dg_ctx *c, uint64_t *out, int n) {
    if (int rc = dgi::bind(c)) return rc;
    return dgi::dp_table_digest(c, out, n);
}
extern "C" int dg_dp_get_launch_profile(dg_ctx *c, char *buf, int cap) {
    if (!c || !c->dp || !buf || cap < 2) { dgi::set_error("dg_dp_get_launch_profile: no state"); return DG_ERR_STATE; }
    std::string out;
    for (int q = 0; q < 64 * 4; ++q) {
        const int64_t n = c->dp->launch_hist[q];
        if (!n) continue;
        char item[96];
        if (q == 0) snprintf(item, sizeof item, "dp_sweep_kernel:%lld", (long long)n);
        else if (q >= 40 * 4) snprintf(item, sizeof item, "dp_sweep_sym_kernel<%d,%s>:%lld", q / 4 - 40, (q & 2) ? "general" : "lean", (long long)n);
        else snprintf(item, sizeof item, "dp_sweep_%s_kernel<%d,%s>:%lld", (q & 1) ? "coop" : "fast", q / 4, (q & 2) ? "general" : "lean", (long long)n);
        if (!out.empty()) out += ' ';
        out += item;
    }
    if ((int)out.size() + 1 > cap) { dgi::set_error("dg_dp_get_launch_profile: buffer too small (%zu needed)", out.size() + 1); return DG_ERR_ARG; }
    memcpy(buf, out.c_str(), out.size() + 1);
    return DG_OK;
}
// Options: parity / test knobs (digest, fast, adaptive_rc, coop, rowx, lean_chain, segment_cells, delta_cap_entries, lattice_chunk_cells,
// graph_batch, warm_ahead), profiler aid (sync_every), tuning (rc_*, bp_nt_min_cells, max_blocks, host_threads).
extern "C" int dg_dp_set_option(dg_ctx *c, const char *key, int64_t v) {
    if (!c || !key) { dgi::set_error("dg_dp_set_option: null"); return DG_ERR_ARG; }
    if (!c->dp) c->dp = new dgi::DpState(c->device);
    dgi::DpState &S = *c->dp;
    dgi::graphs_clear(S);
    struct { const char *name; int64_t *field; int64_t lo; } plain[] = {
        {"digest", &S.want_digest, 0}, {"fast", &S.use_fast, 0}, {"adaptive_rc", &S.adaptive_rc, 0}, {"coop", &S.use_coop, 0},
        {"rowx", &S.use_rowx, 0}, {"lean_chain", &S.use_lean_chain, 0},   // take effect at the next load
        {"segment_cells", &S.segment_cells, 0}, {"sync_every", &S.sync_every, 0}, {"rc_t0_ns", &S.rc_t0_ns, 0}, {"rc_tg_ps", &S.rc_tg_ps, 0},
        {"rc_tw_ps", &S.rc_tw_ps, 0}, {"bp_nt_min_cells", &S.bp_nt_min_cells, 0}, {"warm_ahead", &S.warm_ahead, 0}, {"graph_batch", &S.graph_batch, -1}, {"l2_prefetch", &S.l2_prefetch, 0}, {"delta_overlap", &S.delta_overlap, 0}, {"pf_far", &S.pf_far, 0},
        {"host_threads", &S.host_threads, 1}, {"host_tables", &S.host_tables, 0}, {"side_stream", &S.side_stream, -1}, {"plane_limit", &S.plane_limit, 0},
#ifdef DG_SYM
        {"sym", &S.use_sym, 0}, {"sym_min_k2", &S.sym_min_k2, 1}, {"sym_rc", &S.sym_rc, 1}, {"sym_dbg", &S.sym_dbg, 0}, {"sym_fold", &S.sym_fold, 0},
#endif
        {"test_poison_level", &S.test_poison_level, 0}, {"test_poison_byte", &S.test_poison_byte, 0},
    };
    for (auto &o : plain)
        if (!strcmp(key, o.name)) { *o.field = v < o.lo ? o.lo : v; return DG_OK; }
    if (!strcmp(key, "delta_cap_entries")) S.delta_cap_entries = v > 0 ? v : (int64_t)4 << 30;   // takes effect at the next load
    else if (!strcmp(key, "rc_cap")) S.rc_cap = v > 0 ? v : 65536;
    else if (!strcmp(key, "max_blocks")) S.max_blocks = v > 0 ? v : 1024;
    else if (!strcmp(key, "lattice_chunk_cells")) {          // size of one lattice chunk (in 16-bit back-pointer units = cells on ordinary levels; default 2^32 = 8 GB)
        if (v < 1) { dgi::set_error("lattice_chunk_cells must be positive"); return DG_ERR_ARG; }
        dgi::pool_clear(S);
        S.chunk_units_cfg = S.pool.chunk_units = ((size_t)v + 1) & ~(size_t)1;
    }
    else { dgi::set_error("unknown option %s", key); return DG_ERR_ARG; }
    return DG_OK;
}
