// Read-sharded minimizer scoring inside ONE process (bin/DipGenie --gpus N, BASELINE configs[3]): one host thread and one dg_ctx per
// device, RCCL reached through librccl directly (dlopen: the single-GPU CLI never loads it).  Semantics of solver.cpp:526-555,
// 711-755 over all reads; the exchange is SURVEY.md s8e:
//   1. every rank sketches its contiguous block of reads on its device (dg_sketch_reads_dev: sorted distinct (hash, #reads));
//   2. dictionary path: counts of the haplotype-minimizer dictionary D in the local run, RCCL all-reduce(sum) of the hit vector;
//   3. spectrum path: the uint64 hash space is cut into `world` equal ranges (dg_sketch_partition_dev), the runs go to their owners
//      in one grouped send / receive (the send counts are shared host memory: the ranks are threads of one process), every owner
//      merges what it received (dg_sketch_merge_runs_dev): rank r holds range r of the exact global spectrum;
//   4. the ranges, concatenated in rank order, are Sp_R's keys with kmer_count; the per-range histograms add up to Hist_kmer.
// Transport 1 (tests on a one-GPU box, where RCCL refuses two ranks on one device): the same steps with the collectives staged
// through host memory between the threads -- every device operation is the product's.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <thread>

#include "dg_internal.hpp"

namespace dgi {
namespace {

struct Rccl {                                            // librccl entry points, bound on first use
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((lib = dlopen(n, RTLD_LAZY | RTLD_LOCAL))) break;
        if (!lib) { err = std::string("librccl not found: ") + dlerror(); return false; }
#define DG_SYM(F) do { F = (decltype(F))dlsym(lib, "nccl" #F); if (!F) { err = "librccl lacks nccl" #F; return false; } } while (0)
        DG_SYM(CommInitAll); DG_SYM(CommDestroy); DG_SYM(GetErrorString); DG_SYM(AllReduce); DG_SYM(Send); DG_SYM(Recv); DG_SYM(GroupStart); DG_SYM(GroupEnd);
#undef DG_SYM
        return true;
    }
};

struct Barrier {                                         // reusable barrier of the rank threads (C++17)
    std::mutex mu;
    std::condition_variable cv;
    int n = 0, waiting = 0;
    unsigned long gen = 0;
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

inline void shard_bounds(int64_t n, int world, int rank, int64_t &lo, int64_t &hi) {   // dist_sketch.py: shard_bounds
    const int64_t base = n / world, rem = n % world;
    lo = rank * base + std::min<int64_t>(rank, rem);
    hi = lo + base + (rank < rem ? 1 : 0);
}

}  // namespace
}  // namespace dgi

struct dg_shard {
    int W = 0, transport = 0;
    std::vector<int> dev;
    std::vector<dg_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    dgi::Rccl rccl;
    dgi::Barrier bar;
};

extern "C" dg_shard *dg_shard_create(int n_ranks, const int *devices, int transport) {
    if (n_ranks < 1 || n_ranks > 64 || (transport != 0 && transport != 1)) { dgi::set_error("dg_shard_create: 1..64 ranks, transport 0 (RCCL) or 1 (host-staged)"); return nullptr; }
    dg_shard *S = new dg_shard();
    S->W = n_ranks; S->transport = transport; S->bar.n = n_ranks;
    for (int r = 0; r < n_ranks; ++r) S->dev.push_back(devices ? devices[r] : r);
    if (transport == 0)
        for (int r = 0; r < n_ranks; ++r)
            for (int q = 0; q < r; ++q)
                if (S->dev[q] == S->dev[r]) { dgi::set_error("dg_shard_create: RCCL needs one device per rank (device %d given twice)", S->dev[r]); delete S; return nullptr; }
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    const double tc0 = now();
    for (int r = 0; r < n_ranks; ++r) {
        dg_ctx *c = dg_create(S->dev[r]);                // (no CPU fallback: fails without a gfx950 device)
        if (!c) { for (dg_ctx *q : S->ctx) dg_destroy(q); delete S; return nullptr; }
        S->ctx.push_back(c);
    }
    const double tc_ctx = now();
    if (transport == 0) {
        std::string err;
        if (!S->rccl.load(err)) { dgi::set_error("dg_shard_create: %s", err.c_str()); for (dg_ctx *q : S->ctx) dg_destroy(q); delete S; return nullptr; }
        S->comm.assign(n_ranks, nullptr);
        const double tc1 = now();
        const ncclResult_t e = S->rccl.CommInitAll(S->comm.data(), n_ranks, S->dev.data());
        if (dbg) fprintf(stderr, "[dipgenie_hip] shard: %d contexts %.3f s, librccl bound %.3f s, ncclCommInitAll %.3f s\n", n_ranks, tc_ctx - tc0, tc1 - tc_ctx, now() - tc1);
        if (e != ncclSuccess) { dgi::set_error("ncclCommInitAll(%d devices): %s", n_ranks, S->rccl.GetErrorString(e)); for (dg_ctx *q : S->ctx) dg_destroy(q); delete S; return nullptr; }
    }
    return S;
}

extern "C" void dg_shard_destroy(dg_shard *S) {
    if (!S) return;
    for (ncclComm_t c : S->comm) if (c) (void)S->rccl.CommDestroy(c);
    for (dg_ctx *c : S->ctx) dg_destroy(c);
    delete S;
}

extern "C" int dg_shard_n_ranks(dg_shard *S) { return S ? S->W : 0; }
extern "C" dg_ctx *dg_shard_ctx(dg_shard *S, int rank) { return (S && rank >= 0 && rank < S->W) ? S->ctx[rank] : nullptr; }

namespace dgi {
namespace {

struct ScoreJob {                                        // shared by the rank threads of one dg_shard_score_reads call
    dg_shard *S;
    const char *bases; const int64_t *read_off; int64_t n_reads; int k, w;
    const uint64_t *hap_hash; int64_t n_hap_hash;        // concatenated haplotype minimizer hashes (any order, duplicates): D = their sorted distinct set
    int n_bins;
    std::vector<std::vector<int64_t>> split;             // [rank][W + 1] send offsets of the rank's run
    std::vector<std::vector<uint64_t>> stage_h;          // host-staged transport: the rank's whole run
    std::vector<std::vector<int32_t>> stage_c;
    std::vector<std::vector<int32_t>> stage_hits;        // host-staged transport: the rank's hit vector
    std::vector<std::vector<uint64_t>> out_h;            // the rank's range of the global spectrum
    std::vector<std::vector<int32_t>> out_c;
    std::vector<std::vector<uint64_t>> out_hist;
    std::vector<int32_t> hits;                           // all-reduced hit vector (rank 0 copies it out)
    int64_t n_dict = 0;
    std::vector<int> rc;
    std::vector<std::string> err;
    std::vector<double> ms_sketch, ms_exchange;
};

#define DG_RANK(call) do { if (int rc_ = (call)) { J.rc[r] = rc_; J.err[r] = dg_last_error(); return; } } while (0)
#define DG_RHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { J.rc[r] = DG_ERR_HIP; J.err[r] = std::string(#call " failed: ") + hipGetErrorString(e_); return; } } while (0)
#define DG_RNCCL(call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) { J.rc[r] = DG_ERR_HIP; J.err[r] = std::string(#call " failed: ") + S.rccl.GetErrorString(e_); return; } } while (0)

// One rank.  A rank that fails keeps walking through the barriers (the others must not wait for it forever), and nobody enters the
// exchange unless every rank came through its local part (all_ok(), evaluated behind a barrier).  A failure INSIDE the grouped RCCL
// exchange can still leave the peers waiting in it: that is RCCL's contract.
void score_rank(ScoreJob &J, int r) {
    dg_shard &S = *J.S;
    const int W = S.W;
    dg_ctx *c = S.ctx[r];
    hipStream_t s = c->stream;
    auto all_ok = [&]() { for (int q = 0; q < W; ++q) if (J.rc[q]) return false; return true; };
    DevBuf d_bases, d_off, d_hash, d_cnt, d_dict, d_ones, d_hap, d_hits, d_split, d_rh, d_rc, d_mh, d_mc, d_hist;
    int64_t lo = 0, hi = 0, n_local = 0, n_range = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    auto body1 = [&]() {                                 // local sketch, dictionary counts
        DG_RANK(bind(c));
        DG_RHIP(hipEventCreate(&ev0)); DG_RHIP(hipEventCreate(&ev1)); DG_RHIP(hipEventCreate(&ev2));
        shard_bounds(J.n_reads, W, r, lo, hi);
        const int64_t b0 = J.read_off[lo], b1 = J.read_off[hi], nb = b1 - b0, nr = hi - lo;
        std::vector<int64_t> off((size_t)nr + 1);
        for (int64_t q = 0; q <= nr; ++q) off[q] = J.read_off[lo + q] - b0;
        DG_RANK(d_bases.ensure((size_t)std::max<int64_t>(nb, 1))); DG_RANK(d_off.ensure(8 * ((size_t)nr + 1)));
        const int64_t cap = std::max<int64_t>(nb, 1);                      // distinct (hash, read) pairs <= windows <= bases
        DG_RANK(d_hash.ensure(8 * (size_t)cap)); DG_RANK(d_cnt.ensure(4 * (size_t)cap));
        DG_RHIP(hipMemcpyAsync(d_bases.p, J.bases + b0, (size_t)nb, hipMemcpyHostToDevice, s));
        DG_RHIP(hipMemcpyAsync(d_off.p, off.data(), 8 * off.size(), hipMemcpyHostToDevice, s));
        DG_RHIP(hipEventRecord(ev0, s));
        DG_RANK(dg_sketch_reads_dev(c, d_bases.as<char>(), d_off.as<int64_t>(), nr, nb, J.k, J.w, d_hash.as<uint64_t>(), d_cnt.as<int32_t>(), cap, &n_local));
        DG_RHIP(hipEventRecord(ev1, s));
        if (J.n_hap_hash > 0) {                          // D on this device: sorted distinct haplotype minimizer hashes (every rank builds its own copy)
            DG_RANK(d_hap.ensure(8 * (size_t)J.n_hap_hash)); DG_RANK(d_ones.ensure(4 * (size_t)J.n_hap_hash));
            DG_RANK(d_dict.ensure(8 * (size_t)J.n_hap_hash)); DG_RANK(d_mc.ensure(4 * (size_t)J.n_hap_hash));
            DG_RHIP(hipMemcpyAsync(d_hap.p, J.hap_hash, 8 * (size_t)J.n_hap_hash, hipMemcpyHostToDevice, s));
            DG_RHIP(hipMemsetD32Async((hipDeviceptr_t)d_ones.p, 1, (size_t)J.n_hap_hash, s));   // one occurrence each (a count of 0 would read as exchange padding)
            int64_t nd = 0;
            DG_RANK(dg_sketch_merge_runs_dev(c, d_hap.as<uint64_t>(), d_ones.as<int32_t>(), J.n_hap_hash, d_dict.as<uint64_t>(), d_mc.as<int32_t>(), J.n_hap_hash, &nd));
            if (r == 0) J.n_dict = nd;
            DG_RANK(d_hits.ensure(4 * (size_t)std::max<int64_t>(nd, 1)));
            DG_RHIP(hipMemsetAsync(d_hits.p, 0, 4 * (size_t)std::max<int64_t>(nd, 1), s));
            DG_RANK(dg_sketch_count_dictionary_dev(c, d_dict.as<uint64_t>(), nd, d_hash.as<uint64_t>(), d_cnt.as<int32_t>(), n_local, d_hits.as<int32_t>()));
        }
        DG_RANK(d_split.ensure(8 * ((size_t)W + 1)));
        DG_RANK(dg_sketch_partition_dev(c, d_hash.as<uint64_t>(), n_local, W, d_split.as<int64_t>()));
        J.split[r].assign((size_t)W + 1, 0);
        DG_RHIP(hipMemcpyAsync(J.split[r].data(), d_split.p, 8 * ((size_t)W + 1), hipMemcpyDeviceToHost, s));
        if (S.transport == 1) {                          // host-staged: the whole run and the hit vector go through host memory
            J.stage_h[r].resize((size_t)n_local); J.stage_c[r].resize((size_t)n_local);
            DG_RHIP(hipMemcpyAsync(J.stage_h[r].data(), d_hash.p, 8 * (size_t)n_local, hipMemcpyDeviceToHost, s));
            DG_RHIP(hipMemcpyAsync(J.stage_c[r].data(), d_cnt.p, 4 * (size_t)n_local, hipMemcpyDeviceToHost, s));
        }
        DG_RHIP(hipStreamSynchronize(s));
    };
    body1();
    S.bar.wait();                                        // every rank's send offsets (and staged runs) are visible
    if (S.transport == 1 && J.n_hap_hash > 0 && all_ok()) {               // (n_dict was published by rank 0 in front of the barrier)
        J.stage_hits[r].assign((size_t)J.n_dict, 0);
        if (hipMemcpy(J.stage_hits[r].data(), d_hits.p, 4 * (size_t)J.n_dict, hipMemcpyDeviceToHost) != hipSuccess) { J.rc[r] = DG_ERR_HIP; J.err[r] = "hit vector download failed"; }
    }
    if (S.transport == 1) S.bar.wait();
    auto body2 = [&]() {                                 // hit-vector all-reduce, hash-range exchange, merge of the owned range
        if (J.n_hap_hash > 0) {
            if (S.transport == 0) DG_RNCCL(S.rccl.AllReduce(d_hits.p, d_hits.p, (size_t)J.n_dict, ncclInt32, ncclSum, S.comm[r], s));
            else {
                std::vector<int32_t> sum((size_t)J.n_dict, 0);
                for (int q = 0; q < W; ++q) for (size_t t = 0; t < sum.size(); ++t) sum[t] += J.stage_hits[q][t];
                DG_RHIP(hipMemcpyAsync(d_hits.p, sum.data(), 4 * sum.size(), hipMemcpyHostToDevice, s));
                DG_RHIP(hipStreamSynchronize(s));
            }
        }
        std::vector<int64_t> in_off((size_t)W + 1, 0);
        for (int p = 0; p < W; ++p) in_off[p + 1] = in_off[p] + (J.split[p][r + 1] - J.split[p][r]);
        const int64_t total = in_off[W];
        DG_RANK(d_rh.ensure(8 * (size_t)std::max<int64_t>(total, 1))); DG_RANK(d_rc.ensure(4 * (size_t)std::max<int64_t>(total, 1)));
        if (S.transport == 0) {
            DG_RNCCL(S.rccl.GroupStart());
            for (int p = 0; p < W; ++p) {
                const int64_t out_n = J.split[r][p + 1] - J.split[r][p], in_n = in_off[p + 1] - in_off[p];
                if (p == r) {                            // own range: a device copy
                    if (out_n) {
                        DG_RHIP(hipMemcpyAsync(d_rh.as<uint64_t>() + in_off[p], d_hash.as<uint64_t>() + J.split[r][p], 8 * (size_t)out_n, hipMemcpyDeviceToDevice, s));
                        DG_RHIP(hipMemcpyAsync(d_rc.as<int32_t>() + in_off[p], d_cnt.as<int32_t>() + J.split[r][p], 4 * (size_t)out_n, hipMemcpyDeviceToDevice, s));
                    }
                    continue;
                }
                if (out_n) {
                    DG_RNCCL(S.rccl.Send(d_hash.as<uint64_t>() + J.split[r][p], (size_t)out_n, ncclUint64, p, S.comm[r], s));
                    DG_RNCCL(S.rccl.Send(d_cnt.as<int32_t>() + J.split[r][p], (size_t)out_n, ncclInt32, p, S.comm[r], s));
                }
                if (in_n) {
                    DG_RNCCL(S.rccl.Recv(d_rh.as<uint64_t>() + in_off[p], (size_t)in_n, ncclUint64, p, S.comm[r], s));
                    DG_RNCCL(S.rccl.Recv(d_rc.as<int32_t>() + in_off[p], (size_t)in_n, ncclInt32, p, S.comm[r], s));
                }
            }
            DG_RNCCL(S.rccl.GroupEnd());
        } else {
            for (int p = 0; p < W; ++p) {
                const int64_t in_n = in_off[p + 1] - in_off[p], src = J.split[p][r];
                if (!in_n) continue;
                DG_RHIP(hipMemcpyAsync(d_rh.as<uint64_t>() + in_off[p], J.stage_h[p].data() + src, 8 * (size_t)in_n, hipMemcpyHostToDevice, s));
                DG_RHIP(hipMemcpyAsync(d_rc.as<int32_t>() + in_off[p], J.stage_c[p].data() + src, 4 * (size_t)in_n, hipMemcpyHostToDevice, s));
            }
        }
        DG_RANK(d_mh.ensure(8 * (size_t)std::max<int64_t>(total, 1))); DG_RANK(d_mc.ensure(4 * (size_t)std::max<int64_t>(total, 1)));
        DG_RANK(dg_sketch_merge_runs_dev(c, d_rh.as<uint64_t>(), d_rc.as<int32_t>(), total, d_mh.as<uint64_t>(), d_mc.as<int32_t>(), std::max<int64_t>(total, 1), &n_range));
        DG_RANK(d_hist.ensure(8 * (size_t)J.n_bins));
        DG_RHIP(hipMemsetAsync(d_hist.p, 0, 8 * (size_t)J.n_bins, s));
        DG_RANK(dg_sketch_histogram_dev(c, d_mc.as<int32_t>(), n_range, J.n_bins, d_hist.as<uint64_t>()));
        J.out_h[r].resize((size_t)n_range); J.out_c[r].resize((size_t)n_range); J.out_hist[r].assign((size_t)J.n_bins, 0);
        DG_RHIP(hipMemcpyAsync(J.out_h[r].data(), d_mh.p, 8 * (size_t)n_range, hipMemcpyDeviceToHost, s));
        DG_RHIP(hipMemcpyAsync(J.out_c[r].data(), d_mc.p, 4 * (size_t)n_range, hipMemcpyDeviceToHost, s));
        DG_RHIP(hipMemcpyAsync(J.out_hist[r].data(), d_hist.p, 8 * (size_t)J.n_bins, hipMemcpyDeviceToHost, s));
        if (r == 0 && J.n_hap_hash > 0) { J.hits.resize((size_t)J.n_dict); DG_RHIP(hipMemcpyAsync(J.hits.data(), d_hits.p, 4 * (size_t)J.n_dict, hipMemcpyDeviceToHost, s)); }
        DG_RHIP(hipEventRecord(ev2, s));
        DG_RHIP(hipStreamSynchronize(s));
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, ev0, ev1); (void)hipEventElapsedTime(&b, ev1, ev2);
        J.ms_sketch[r] = a; J.ms_exchange[r] = b;
    };
    if (all_ok()) body2();                               // (a failed rank anywhere: nobody enters a collective)
    S.bar.wait();                                        // staging buffers stay alive until every rank has copied from them
    for (hipEvent_t e : {ev0, ev1, ev2}) if (e) (void)hipEventDestroy(e);
}

}  // namespace
}  // namespace dgi

extern "C" int dg_shard_score_reads(dg_shard *S, const char *bases, const int64_t *read_off, int64_t n_reads, int k, int w, const uint64_t *hap_hash, int64_t n_hap_hash,
                                    uint64_t **sp_hash, int32_t **sp_count, int64_t *n_sp, int64_t *hist, int n_bins, int64_t *n_dict, int64_t *dict_hits,
                                    double *ms_sketch_max, double *ms_exchange_max) {
    using namespace dgi;
    if (!S || !bases || !read_off || n_reads < 0 || !sp_hash || !sp_count || !n_sp || n_bins < 2 || (n_hap_hash > 0 && !hap_hash)) { set_error("dg_shard_score_reads: bad arguments"); return DG_ERR_ARG; }
    const int W = S->W;
    ScoreJob J;
    J.S = S; J.bases = bases; J.read_off = read_off; J.n_reads = n_reads; J.k = k; J.w = w; J.hap_hash = hap_hash; J.n_hap_hash = n_hap_hash; J.n_bins = n_bins;
    J.split.resize(W); J.stage_h.resize(W); J.stage_c.resize(W); J.stage_hits.resize(W); J.out_h.resize(W); J.out_c.resize(W); J.out_hist.resize(W);
    J.rc.assign(W, 0); J.err.assign(W, ""); J.ms_sketch.assign(W, 0); J.ms_exchange.assign(W, 0);
    std::vector<std::thread> th;
    for (int r = 1; r < W; ++r) th.emplace_back(score_rank, std::ref(J), r);
    score_rank(J, 0);
    for (auto &t : th) t.join();
    for (int r = 0; r < W; ++r)
        if (J.rc[r]) { set_error("dg_shard_score_reads: rank %d: %s", r, J.err[r].c_str()); return J.rc[r]; }
    int64_t n = 0;
    for (int r = 0; r < W; ++r) n += (int64_t)J.out_h[r].size();
    uint64_t *oh = (uint64_t *)malloc(8 * (size_t)std::max<int64_t>(n, 1));
    int32_t *oc = (int32_t *)malloc(4 * (size_t)std::max<int64_t>(n, 1));
    if (!oh || !oc) { free(oh); free(oc); set_error("dg_shard_score_reads: out of host memory"); return DG_ERR_OOM; }
    int64_t at = 0;
    for (int r = 0; r < W; ++r) {                        // ranges are disjoint and ascend with the rank: concatenation = the sorted global spectrum
        if (!J.out_h[r].empty()) { memcpy(oh + at, J.out_h[r].data(), 8 * J.out_h[r].size()); memcpy(oc + at, J.out_c[r].data(), 4 * J.out_c[r].size()); }
        at += (int64_t)J.out_h[r].size();
    }
    *sp_hash = oh; *sp_count = oc; *n_sp = n;
    if (hist) for (int b = 0; b < n_bins; ++b) { int64_t v = 0; for (int r = 0; r < W; ++r) v += (int64_t)J.out_hist[r][b]; hist[b] = v; }
    if (n_dict) *n_dict = J.n_dict;
    if (dict_hits) { int64_t v = 0; for (int32_t x : J.hits) v += x > 0; *dict_hits = v; }
    if (ms_sketch_max) *ms_sketch_max = *std::max_element(J.ms_sketch.begin(), J.ms_sketch.end());
    if (ms_exchange_max) *ms_exchange_max = *std::max_element(J.ms_exchange.begin(), J.ms_exchange.end());
    return DG_OK;
}
