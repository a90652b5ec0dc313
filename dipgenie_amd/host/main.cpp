// DipGenie drop-in CLI for the diploid hot path on MI355X.
//
// Same command line as the reference (/root/reference/src/main.cpp:39-111): -t -p -R -g -r -o -k -w
// -T -d are honoured; -a -q -N -m -P -H -l -c are accepted and ignored exactly as the reference's DP
// path ignores them. The two device loops go through libdipgenie_hip.so; there is no
// CPU fallback: without a usable gfx950 device the program exits with an error.
#include <sys/resource.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "pipeline.hpp"

// The device context is created on a side thread while the main thread reads the GFA and the reads (HIP start-up is
// ~0.3 s, the two overlap); the first backend call joins it.
struct LazyCtx {
    std::thread th;
    dg_ctx *ctx = nullptr;
    std::string err;
    bool joined = false;
    void start(int device, int k, int w) {
        th = std::thread([this, device, k, w] {
            const double tc0 = dg::now_s();
            ctx = dg_create(device);
            if (!ctx) { err = dg_last_error(); return; }
            const double tc1 = dg::now_s();
            // first launch from the sketch module loads its code object and makes the first device allocations (~10 ms):
            // paid here, beside the GFA parse, instead of in front of the first haplotype
            const std::string warm(256, 'A');
            uint64_t *h = nullptr; int64_t *p = nullptr; int64_t n = 0;
            if (k >= 1 && k <= 255 && w >= 1 && dg_sketch_haplotype(ctx, warm.data(), (int64_t)warm.size(), k, w, &h, &p, &n) == DG_OK) { dg_free(h); dg_free(p); }
            if (getenv("DG_DEBUG")) fprintf(stderr, "[dg::main] side thread: dg_create %.3f s, first sketch call %.3f s\n", tc1 - tc0, dg::now_s() - tc1);
        });
    }
    dg_ctx *get() {
        if (!joined) { th.join(); joined = true; }
        return ctx;
    }
    // the reference's error exits (exit(1) in read_gfa on a reverse-strand walk step, solver.cpp:116-119) run the static destructors while
    // the side thread may still be creating the context: a joinable std::thread there would end the process with SIGABRT instead of code 1
    ~LazyCtx() { if (th.joinable()) th.join(); }
};
static LazyCtx g_lazy;

// DG_DEBUG timeline: seconds since the kernel started this process (exec, dynamic linking and the HIP library's static
// initialisers all run before main)
static double since_exec_s() {
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return -1;
    char buf[1024];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *q = strrchr(buf, ')');                      // field 22 (starttime, clock ticks since boot) = 20th after the ')'
    if (!q) return -1;
    unsigned long long start = 0;
    int field = 3;                                          // q + 2 = field 3 (state)
    for (q += 2; *q && field < 22; ++q) if (*q == ' ') ++field;
    if (sscanf(q, "%llu", &start) != 1) return -1;
    struct timespec ts;
    clock_gettime(CLOCK_BOOTTIME, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec - (double)start / (double)sysconf(_SC_CLK_TCK);
}
static const char *b_last_error() { return g_lazy.joined && !g_lazy.ctx ? g_lazy.err.c_str() : dg_last_error(); }

static int b_sketch_reads(void *c, const char *b, const int64_t *off, int64_t n, int k, int w, uint64_t **h, int32_t **cnt, int64_t *nd) {
    dg_ctx *x = ((LazyCtx *)c)->get();
    return x ? dg_sketch_reads(x, b, off, n, k, w, h, cnt, nd) : DG_ERR_NO_DEVICE;
}
static int b_sketch_hap(void *c, const char *s, int64_t len, int k, int w, uint64_t **h, int64_t **p, int64_t *n) {
    dg_ctx *x = ((LazyCtx *)c)->get();
    return x ? dg_sketch_haplotype(x, s, len, k, w, h, p, n) : DG_ERR_NO_DEVICE;
}
static int b_dp(void *c, const dg_dp_graph *g, dg_dp_result *r) {
    dg_ctx *x = ((LazyCtx *)c)->get();
    if (!x) return DG_ERR_NO_DEVICE;
    // DG_DP_OPTIONS="key=value,key=value": dg_dp_set_option tuning knobs for profiling and for the segmented-lattice tests (none needed
    // in normal use).  Honoured with a notice on stderr; the fault-injection keys (test_*) are refused here -- they exist for the
    // library's own tests, which set them through the C ABI.
    if (const char *o = getenv("DG_DP_OPTIONS")) {
        std::string all(o);
        if (!all.empty()) fprintf(stderr, "[dg::main] DG_DP_OPTIONS honoured: %s\n", all.c_str());
        for (size_t a = 0; a < all.size();) {
            size_t b = all.find(',', a);
            if (b == std::string::npos) b = all.size();
            const std::string kv = all.substr(a, b - a);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) {
                const std::string key = kv.substr(0, eq);
                if (key.rfind("test_", 0) == 0) { fprintf(stderr, "[E::main] DG_DP_OPTIONS: '%s' is a fault-injection key, refused\n", key.c_str()); return DG_ERR_ARG; }
                if (dg_dp_set_option(x, key.c_str(), atoll(kv.c_str() + eq + 1)) != DG_OK) return DG_ERR_ARG;
            }
            a = b + 1;
        }
    }
    return dg_dp_solve_diploid(x, g, r);
}
static int b_hap(void *c, const dg_hap_graph *g, int32_t *dp, int32_t *bv, int32_t *br) {
    dg_ctx *x = ((LazyCtx *)c)->get();
    return x ? dg_dp_solve_haploid(x, g, dp, bv, br) : DG_ERR_NO_DEVICE;
}
static int b_anchor_begin(void *c, int32_t nh, int32_t nv, const int32_t *top, int k, int w) {
    dg_ctx *x = ((LazyCtx *)c)->get();
    return x ? dg_anchor_begin(x, nh, nv, top, k, w) : DG_ERR_NO_DEVICE;
}
static int b_anchor_add(void *c, int32_t h, const char *s, int64_t len, const int32_t *sv, const int64_t *ss, int64_t ns, int64_t *n) {
    dg_ctx *x = ((LazyCtx *)c)->get();
    return x ? dg_anchor_add_haplotype(x, h, s, len, sv, ss, ns, n) : DG_ERR_NO_DEVICE;
}
static int b_anchor_finish(void *c, const uint64_t *sp, int64_t n, float thr, dg_anchor_result *out) {
    dg_ctx *x = ((LazyCtx *)c)->get();
    return x ? dg_anchor_finish(x, sp, n, thr, out) : DG_ERR_NO_DEVICE;
}
static void b_hint(void *c, int64_t est_cells) {   // overlap the lattice reservation (2 B/cell) with the host stages
    dg_ctx *x = ((LazyCtx *)c)->get();
    const double bytes = 2.0 * (double)est_cells;     // a low estimate is harmless: the rest is mapped before the sweep starts
    if (x && bytes >= 4e9) dg_dp_prealloc(x, bytes > 8e18 ? 0 : (int64_t)bytes);   // small lattices allocate instantly anyway
}

// ---- --gpus N: minimizer scoring sharded over N devices inside this one process (BASELINE configs[3]) ----
// One host thread per device.  Haplotype h is sketched on device h mod N (index_kmers is independent per haplotype, solver.cpp:470-473);
// the reads are scored by dg_shard_score_reads (contiguous blocks, RCCL all-reduce of the dictionary hit vector, hash-range exchange,
// per-owner merge: dipgenie_amd/csrc/dg_shard.hip); spectrum and sketches are handed to the host pipeline exactly as the Python
// driver hands them in (Pipeline::inj_*), and everything from the anchor join on -- fit, graph, DP, FASTA -- runs on the first device.
struct ShardInfo { int gpus = 0, transport = 0; int64_t n_dict = 0, dict_hits = 0; double ms_sketch = 0, ms_exchange = 0, wall_s = 0; };
static int run_sharded(dg::Pipeline &p, int n_gpus, int transport, const std::vector<int> &devices, ShardInfo &info, std::string &err) {
    const double t0 = dg::now_s();
    p.sum = dg::Summary();
    // contexts + RCCL communicator (HIP start-up, librccl, ncclCommInitAll: ~1.3 s) on a thread of their own beside the GFA and the reads
    dg_shard *sh = nullptr;
    std::string sh_err, reads_err;
    int reads_rc = 0;
    std::thread sh_thread([&] { sh = dg_shard_create(n_gpus, devices.empty() ? nullptr : devices.data(), transport); if (!sh) sh_err = dg_last_error(); });
    std::thread reads_thread([&] { reads_rc = p.load_reads(reads_err); });
    const int grc = p.load_graph(err);
    reads_thread.join();
    sh_thread.join();
    if (!sh) { err = sh_err; return -2; }                                // no gfx950 device / no librccl: no fallback
    if (grc || reads_rc) { if (!grc) err = reads_err; dg_shard_destroy(sh); return -1; }
    const double t1 = dg::now_s();
    p.inj_hap.assign(p.num_walks, {});
    std::vector<std::string> terr((size_t)n_gpus);
    std::vector<std::thread> th;
    for (int r = 0; r < n_gpus; ++r)
        th.emplace_back([&, r] {
            dg_ctx *c = dg_shard_ctx(sh, r);
            for (uint32_t h = (uint32_t)r; h < p.num_walks; h += (uint32_t)n_gpus) {
                const std::string seq = p.haplotype_sequence(h);
                uint64_t *hh = nullptr; int64_t *pp = nullptr; int64_t n = 0;
                if (dg_sketch_haplotype(c, seq.data(), (int64_t)seq.size(), p.opt.k, p.opt.w, &hh, &pp, &n) != DG_OK) { terr[r] = dg_last_error(); return; }
                p.inj_hap[h].hash.assign(hh, hh + n); p.inj_hap[h].pos.assign(pp, pp + n); p.inj_hap[h].set = true;
                dg_free(hh); dg_free(pp);
            }
        });
    for (auto &t : th) t.join();
    for (auto &e : terr) if (!e.empty()) { err = "haplotype sketch: " + e; dg_shard_destroy(sh); return -1; }
    std::vector<uint64_t> hap_hash;
    for (auto &hs : p.inj_hap) hap_hash.insert(hap_hash.end(), hs.hash.begin(), hs.hash.end());
    std::vector<int64_t> off(p.reads.size() + 1, 0);
    for (size_t r = 0; r < p.reads.size(); ++r) off[r + 1] = off[r] + (int64_t)p.reads[r].second.size();
    std::string bases;
    bases.reserve((size_t)off.back());
    for (auto &rd : p.reads) bases += rd.second;
    const int64_t n_reads = (int64_t)p.reads.size();
    p.reads.clear(); p.reads.shrink_to_fit();
    const double t2 = dg::now_s();
    uint64_t *sph = nullptr; int32_t *spc = nullptr; int64_t nsp = 0;
    std::vector<int64_t> hist(4096, 0);
    if (dg_shard_score_reads(sh, bases.data(), off.data(), n_reads, p.opt.k, p.opt.w, hap_hash.data(), (int64_t)hap_hash.size(), &sph, &spc, &nsp, hist.data(), (int)hist.size(),
                             &info.n_dict, &info.dict_hits, &info.ms_sketch, &info.ms_exchange) != DG_OK) { err = dg_last_error(); dg_shard_destroy(sh); return -1; }
    p.inj_sp_hash.assign(sph, sph + nsp); p.inj_sp_count.assign(spc, spc + nsp); p.inj_hist = hist; p.spectrum_injected = true;
    dg_free(sph); dg_free(spc);
    const double t3 = dg::now_s();
    if (!p.opt.quiet) fprintf(stderr, "[dg::shard] %d ranks (%s): haplotype sketches %.3f s, read scoring %.3f s (slowest rank: sketch %.2f ms, exchange + merge %.2f ms); dictionary %lld, hit by reads %lld\n",
                              n_gpus, transport == 0 ? "RCCL" : "host-staged", t2 - t1, t3 - t2, info.ms_sketch, info.ms_exchange, (long long)info.n_dict, (long long)info.dict_hits);
    p.sum.stage_s.emplace_back("sharded: contexts + communicator (beside GFA + reads)", t1 - t0);
    p.sum.stage_s.emplace_back("sharded: haplotype sketches", t2 - t1);
    p.sum.stage_s.emplace_back("sharded: read scoring", t3 - t2);
    g_lazy.ctx = dg_shard_ctx(sh, 0); g_lazy.joined = true;              // the rest of the pipeline: the first device's context
    info.gpus = n_gpus; info.transport = transport;
    const int rc = p.run_loaded(err);
    info.wall_s = dg::now_s() - t0;
    return rc;                                                           // (the shard object lives until the process exits: its first context is in use)
}

static void usage(FILE *fp, const dg::Options &o) {   // main.cpp:90-110
    fprintf(fp, "Usage: PHI -g <target.gfa> -r <reads.fa> -o <haplotype.fasta> \n");
    fprintf(fp, "Options:\n");
    fprintf(fp, "    -a bool      DP approximation mode\n");
    fprintf(fp, "    -k INT       K-mer size [%d]\n", o.k);
    fprintf(fp, "    -w INT       Minimizer window size [%d]\n", o.w);
    fprintf(fp, "    -R INT       Recombination limit [%d]\n", o.R);
    fprintf(fp, "    -p INT       Ploidy (default diploid i.e -p2, use -p1 for haploid) [%d]\n", o.ploidy);
    fprintf(fp, "    -T FLOAT     Threshold for minimizer filtering [%.3f]\n", o.threshold);
    fprintf(fp, "    -t INT       Threads [%d]\n", o.threads);
    fprintf(fp, "    -g INT       GFA file [%s]\n", o.gfa_file.c_str());
    fprintf(fp, "    -r INT       Read [%s]\n", o.reads_file.c_str());
    fprintf(fp, "    -o INT       Output haplotype [%s]\n", o.hap_file.c_str());
    fprintf(fp, "    -d bool      Debug mode [%d]\n", (int)o.debug);
    fprintf(fp, "    -G INT       (MI355X build) HIP device ordinal [0]\n");
    fprintf(fp, "    --gpus INT   (MI355X build) shard the minimizer scoring over INT devices (RCCL); the DP runs on the first [1]\n");
}

int main(int argc, char **argv) {
    // kernel arguments in device memory (the default of this ROCm stack; in host memory every level launch of the sweep
    // costs 1.9 us more: 929 vs 666 ms on MHC-24) -- pinned before the HIP runtime starts
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
    const bool dbg_tl = getenv("DG_DEBUG") != nullptr;
    if (dbg_tl) fprintf(stderr, "[dg::main] main() entered %.3f s after exec\n", since_exec_s());
    dg::Pipeline p;
    int device = 0, help = 0;
    std::string json;
    for (int i = 1; i < argc; ++i)
        if (!strcmp(argv[i], "--version")) { fprintf(stderr, "PHI version: 1.0 (dipgenie-mi355x)\n"); return 0; }
    // long options of this build, taken out of argv before the reference's getopt string sees it:
    //   --gpus N, --shard-transport rccl|host (host: tests on a one-GPU box), --shard-devices a,b,c (default 0 .. N-1; host transport: -G for every rank)
    int n_gpus = 1, shard_transport = -1;
    std::vector<int> shard_devices;
    {
        int w = 1;
        for (int i = 1; i < argc; ++i) {
            auto val = [&](const char *name) -> const char * {
                const size_t n = strlen(name);
                if (!strncmp(argv[i], name, n) && argv[i][n] == '=') return argv[i] + n + 1;
                if (!strcmp(argv[i], name) && i + 1 < argc) return argv[++i];
                return nullptr;
            };
            if (!strncmp(argv[i], "--gpus", 6)) { const char *v = val("--gpus"); if (v) { n_gpus = atoi(v); continue; } }
            if (!strncmp(argv[i], "--shard-transport", 17)) { const char *v = val("--shard-transport"); if (v) { shard_transport = !strcmp(v, "host") ? 1 : 0; continue; } }
            if (!strncmp(argv[i], "--shard-devices", 15)) { const char *v = val("--shard-devices"); if (v) { for (const char *q = v; *q;) { shard_devices.push_back(atoi(q)); q = strchr(q, ','); if (!q) break; ++q; } continue; } }
            argv[w++] = argv[i];
        }
        argc = w;
    }
    int c;
    // reference option string: "x:p:d:c:l:s:m:R:P:a:q:T:H:N:m:h:k:w:t:g:r:o:DSc" (main.cpp:39); -G -J -D -A -X are ours
    while ((c = getopt(argc, argv, "x:p:d:c:l:s:m:R:P:a:q:T:H:N:h:k:w:t:g:r:o:G:J:D:A:X")) >= 0) {
        switch (c) {
        case 'w': p.opt.w = atoi(optarg); break;
        case 'k': p.opt.k = atoi(optarg); break;
        case 'p': p.opt.ploidy = atoi(optarg); break;
        case 't': p.opt.threads = atoi(optarg); break;
        case 'g': p.opt.gfa_file = optarg; break;
        case 'R': p.opt.R = atoi(optarg); break;
        case 'T': p.opt.threshold = (float)atof(optarg); break;
        case 'r': p.opt.reads_file = optarg; break;
        case 'o': p.opt.hap_file = optarg; break;
        case 'd': p.opt.debug = atoi(optarg); break;
        case 'h': help = 1; break;
        case 'G': device = atoi(optarg); break;
        case 'J': json = optarg; break;
        case 'D': p.opt.dump_prefix = optarg; break;
        case 'A': p.opt.anchor_dump = optarg; break;
        case 'X': p.opt.dump_only = true; break;
        default: break;   // parsed-but-unused on this path
        }
    }
    if (argc < 2 || p.opt.gfa_file.empty() || p.opt.reads_file.empty() || p.opt.hap_file.empty() || help) {
        usage(stderr, p.opt);
        return 1;
    }
    const double t0 = dg::now_s();
    const bool sharded = n_gpus > 1 || shard_transport >= 0;
    if (n_gpus < 1 || n_gpus > 64) { fprintf(stderr, "[E::main] --gpus must be 1 .. 64\n"); return 1; }
    if (sharded && p.opt.ploidy != 2) { fprintf(stderr, "[E::main] --gpus shards the diploid path (-p2)\n"); return 1; }
    if (!sharded) g_lazy.start(device, p.opt.k, p.opt.w);
    p.be.ctx = &g_lazy;
    p.be.sketch_reads = b_sketch_reads;
    p.be.sketch_haplotype = b_sketch_hap;
    p.be.dp_solve_diploid = b_dp;
    p.be.free_buf = dg_free;
    p.be.hint_dp_soon = b_hint;
    p.be.dp_solve_haploid = b_hap;
    if (const char *hm = getenv("DG_HAPLOID")) p.opt.haploid_mode = !strcmp(hm, "host") ? 1 : (!strcmp(hm, "device") ? 2 : 0);   // default: by graph shape
    p.be.anchor_begin = b_anchor_begin;
    p.be.anchor_add_haplotype = b_anchor_add;
    p.be.anchor_finish = b_anchor_finish;
    if (getenv("DG_HOST_ANCHORS")) p.opt.host_anchors = true;
    p.opt.leak_at_exit = !getenv("DG_CLEAN_EXIT");   // A/B and parity runs: the host join / filter / sort
    p.be.last_error = b_last_error;
    std::string err;
    if (dbg_tl) fprintf(stderr, "[dg::main] run() starts %.3f s after main\n", dg::now_s() - t0);
    ShardInfo shard;
    int rc;
    if (sharded) {
        if (shard_transport < 0) shard_transport = 0;
        if (shard_devices.empty() && shard_transport == 1) shard_devices.assign((size_t)n_gpus, device);   // host-staged: every rank on -G
        if (!shard_devices.empty() && (int)shard_devices.size() != n_gpus) { fprintf(stderr, "[E::main] --shard-devices needs %d entries\n", n_gpus); return 1; }
        rc = run_sharded(p, n_gpus, shard_transport, shard_devices, shard, err);
        if (rc == -2) { fprintf(stderr, "[E::main] %s\n", err.c_str()); return 2; }
    } else rc = p.run(err);
    if (dbg_tl) {
        fprintf(stderr, "[dg::main] run() returned %.3f s after main", dg::now_s() - t0);
        for (auto &st : p.sum.stage_s) if (st.first == "total") fprintf(stderr, " (its own total: %.3f s)", st.second);
        fputc('\n', stderr);
    }
    dg_ctx *ctx = g_lazy.get();
    if (!ctx) { fprintf(stderr, "[E::main] %s\n", g_lazy.err.c_str()); return 2; }   // no gfx950 device: no CPU fallback
    if (rc != 0 && err == "dump_only") { std::cout.flush(); fflush(nullptr); _exit(0); }   // -X: stop after the dump(s)
    if (rc != 0) { fprintf(stderr, "[E::main] %s\n", err.c_str()); dg_destroy(ctx); return 1; }
    dg_dp_timing tm;
    const bool have_tm = p.opt.ploidy == 2 && dg_dp_get_timing(ctx, &tm) == DG_OK;
    if (have_tm)
        fprintf(stderr, "[dg::dp] delta %.3f ms, forward %.3f ms (%lld launches), traceback %.3f ms; %.3f G cells, %.3f G relaxations\n",
                tm.delta_ms, tm.forward_ms, (long long)tm.n_forward_launches, tm.traceback_ms, p.sum.cells / 1e9, p.sum.relaxations / 1e9);
    if (!json.empty()) {
        FILE *f = fopen(json.c_str(), "w");
        if (f) {
            fprintf(f, "{\"dp_value\": %d, \"s_het\": %d, \"r1\": %d, \"r2\": %d, \"obj\": %d, \"len1\": %lld, \"len2\": %lld, "
                       "\"spectrum\": %lld, \"n_levels\": %lld, \"n_vertices\": %lld, \"cells\": %llu, \"relaxations\": %llu, "
                       "\"gpus\": %d, \"shard_transport\": \"%s\", \"dictionary\": %lld, \"dictionary_hits\": %lld, \"shard_sketch_ms\": %.3f, \"shard_exchange_ms\": %.3f, "
                       "\"best_r_haploid\": %d, \"fit_nll\": %.17g, \"dp_segments\": %d, \"dp_chunks\": %d, \"dp_forward_ms\": %.3f, \"dp_traceback_ms\": %.3f, \"dp_forward_launches\": %lld, \"dp_edge_pairs\": %llu, \"dp_colour_entries\": %llu, \"stages\": {",
                    p.sum.dp_value, p.sum.s_het, p.sum.r1, p.sum.r2, p.sum.obj, (long long)p.sum.len1, (long long)p.sum.len2,
                    (long long)p.sum.spectrum, (long long)p.sum.n_levels, (long long)p.sum.n_vertices,
                    (unsigned long long)p.sum.cells, (unsigned long long)p.sum.relaxations, sharded ? shard.gpus : 1, !sharded ? "none" : (shard.transport ? "host" : "rccl"),
                    (long long)shard.n_dict, (long long)shard.dict_hits, shard.ms_sketch, shard.ms_exchange, p.sum.best_r_haploid, p.sum.fit.nll,
                    have_tm ? tm.n_segments : 0, have_tm ? tm.n_chunks : 0, have_tm ? tm.forward_ms : 0.f, have_tm ? tm.traceback_ms : 0.f,
                    have_tm ? (long long)tm.n_forward_launches : 0LL, have_tm ? (unsigned long long)tm.edge_pairs : 0ULL, have_tm ? (unsigned long long)tm.colour_entries : 0ULL);
            for (size_t i = 0; i < p.sum.stage_s.size(); ++i)
                fprintf(f, "%s\"%s\": %.6f", i ? ", " : "", p.sum.stage_s[i].first.c_str(), p.sum.stage_s[i].second);
            fprintf(f, "}}\n");
            fclose(f);
        }
    }
    fprintf(stderr, "[M::main] Real time: %.3f sec\n", dg::now_s() - t0);
    if (dbg_tl) {
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[dg::main] leaving %.3f s after exec; peak RSS %.2f GB, %ld minor page faults, user %.2f s, system %.2f s\n", since_exec_s(),
                ru.ru_maxrss / 1048576.0, ru.ru_minflt, ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec, ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec);
        if (FILE *f = fopen("/proc/self/smaps_rollup", "r")) {      // what the resident set is made of right now
            char line[256];
            while (fgets(line, sizeof line, f))
                if (!strncmp(line, "Rss:", 4) || !strncmp(line, "AnonHugePages:", 14) || !strncmp(line, "Anonymous:", 10) || !strncmp(line, "Shared_Clean:", 13) ||
                    !strncmp(line, "Private_Clean:", 14) || !strncmp(line, "Pss_File:", 9) || !strncmp(line, "Pss_Shmem:", 10))
                    fprintf(stderr, "[dg::main] smaps_rollup %s", line);
            fclose(f);
        }
    }
    // Everything is written and closed.  Tearing down tens of GB of device chunks and host vectors one by one costs
    // ~0.4 s that the operating system does for free at exit; DG_CLEAN_EXIT=1 keeps the orderly path (leak checkers).
    if (!getenv("DG_CLEAN_EXIT")) {
        std::cout.flush();
        fflush(nullptr);
        _exit(0);
    }
    const double td0 = dg::now_s();
    dg_destroy(ctx);
    if (dbg_tl) fprintf(stderr, "[dg::main] dg_destroy %.3f s\n", dg::now_s() - td0);
    if (getenv("DG_CLEAN_EXIT_FAST")) { std::cout.flush(); fflush(nullptr); _exit(0); }   // (measurement: device side released, host side left to the OS)
    return 0;
}
