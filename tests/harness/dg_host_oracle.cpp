// tests/harness/dg_host_oracle.cpp -- TEST HARNESS (not product).
//
// Runs the product's host pipeline (dipgenie_amd/host) with the two device loops supplied by the CPU
// oracle (oracle/liboracle.so) instead of libdipgenie_hip.so.  Purpose: prove, on a machine without
// a GPU, that the HOST stages (GFA reader, read_gfa order, anchors/filter, fit/classify, expanded
// graph, levelize, traceback, FASTA writer) reproduce the reference byte-for-byte, and dump the
// levelized DP graph (.dpg) that the GPU parity tests feed to both the oracle and the HIP path.
// Same flags as the product CLI, plus -D <prefix> (dump) and -J <file> (JSON summary), -A <file> (Anchor_hits dump).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>

#include "../../dipgenie_amd/host/pipeline.hpp"
#include "../../oracle/oracle.h"

// Optional on-disk cache of the oracle's sketches (-C dir): the literal string-based restatement needs ~95 s for
// 24 x 5 Mbp; host-stage work on big graphs re-uses its outputs. Keyed by a checksum of the input bytes.
static std::string g_cache;
static uint64_t fnv(const char *p, int64_t n, uint64_t h = 1469598103934665603ULL) {
    for (int64_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ULL; }
    return h;
}
template <class A, class B> static bool cache_load(const std::string &f, A **a, B **b, int64_t *n) {
    FILE *fp = fopen(f.c_str(), "rb");
    if (!fp) return false;
    bool ok = fread(n, 8, 1, fp) == 1;
    if (ok) {
        *a = (A *)malloc(sizeof(A) * (*n + 1)); *b = (B *)malloc(sizeof(B) * (*n + 1));
        ok = fread(*a, sizeof(A), *n, fp) == (size_t)*n && fread(*b, sizeof(B), *n, fp) == (size_t)*n;
    }
    fclose(fp);
    return ok;
}
template <class A, class B> static void cache_save(const std::string &f, const A *a, const B *b, int64_t n) {
    FILE *fp = fopen(f.c_str(), "wb");
    if (!fp) return;
    fwrite(&n, 8, 1, fp); fwrite(a, sizeof(A), n, fp); fwrite(b, sizeof(B), n, fp);
    fclose(fp);
}
static int o_sketch_reads(void *, const char *b, const int64_t *off, int64_t n, int k, int w, uint64_t **h, int32_t **c, int64_t *nd) {
    std::string f;
    if (!g_cache.empty()) {
        f = g_cache + "/reads_" + std::to_string(fnv(b, off[n], fnv((const char *)off, 8 * (n + 1)))) + "_" + std::to_string(k) + "_" + std::to_string(w);
        if (cache_load(f, h, c, nd)) return 0;
    }
    int rc = orc_sketch_reads(b, off, n, k, w, h, c, nd);
    if (rc == 0 && !f.empty()) cache_save(f, *h, *c, *nd);
    return rc;
}
static int o_sketch_hap(void *, const char *s, int64_t len, int k, int w, uint64_t **h, int64_t **p, int64_t *n) {
    std::string f;
    if (!g_cache.empty()) {
        f = g_cache + "/hap_" + std::to_string(fnv(s, len)) + "_" + std::to_string(k) + "_" + std::to_string(w);
        if (cache_load(f, h, p, n)) return 0;
    }
    int64_t cnt = orc_minimizers(s, len, k, w, nullptr, nullptr, 0);
    *h = (uint64_t *)malloc(sizeof(uint64_t) * (cnt + 1));
    *p = (int64_t *)malloc(sizeof(int64_t) * (cnt + 1));
    *n = orc_minimizers(s, len, k, w, *h, *p, cnt);
    if (!f.empty()) cache_save(f, *h, *p, *n);
    return 0;
}
static int o_dp(void *, const dg_dp_graph *g, dg_dp_result *r) {
    static_assert(sizeof(orc_dp_graph) == sizeof(dg_dp_graph), "layout");
    static_assert(sizeof(orc_dp_result) == sizeof(dg_dp_result), "layout");
    return orc_dp_solve_diploid((const orc_dp_graph *)g, (orc_dp_result *)r, nullptr);
}
static const char *o_err() { return "oracle"; }

// --fit: read "multiplicity freq" lines, print the host fitter's result in ref_harness's format
static int fit_mode() {
    std::vector<dg::HistBin> H;
    int m, maxm = 0; double f;
    while (scanf("%d %lf", &m, &f) == 2) { H.push_back({m, f}); if (m > maxm) maxm = m; }
    auto res = dg::kg_fit(H, 10, maxm, 8);
    printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", res.nll, res.P.u_v, res.P.sd_v, res.P.var_w,
           res.P.zp_copy, res.P.zp_copy_het, res.P.p_d, res.P.p_e, res.P.err_shape);
    for (int x = 1; x <= maxm; ++x) putchar(dg::kg_is_hom(res.P, x) ? 'O' : 'E');
    putchar('\n');
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "--fit")) return fit_mode();
    dg::Pipeline p;
    std::string json;
    int c;
    while ((c = getopt(argc, argv, "t:p:R:g:r:o:k:w:T:d:D:J:qXC:A:")) >= 0) {
        switch (c) {
        case 't': p.opt.threads = atoi(optarg); break;
        case 'p': p.opt.ploidy = atoi(optarg); break;
        case 'R': p.opt.R = atoi(optarg); break;
        case 'g': p.opt.gfa_file = optarg; break;
        case 'r': p.opt.reads_file = optarg; break;
        case 'o': p.opt.hap_file = optarg; break;
        case 'k': p.opt.k = atoi(optarg); break;
        case 'w': p.opt.w = atoi(optarg); break;
        case 'T': p.opt.threshold = (float)atof(optarg); break;
        case 'd': p.opt.debug = atoi(optarg); break;
        case 'D': p.opt.dump_prefix = optarg; break;
        case 'J': json = optarg; break;
        case 'q': p.opt.quiet = true; break;
        case 'X': p.opt.dump_only = true; break;
        case 'C': g_cache = optarg; break;
        case 'A': p.opt.anchor_dump = optarg; break;
        }
    }
    p.be.sketch_reads = o_sketch_reads;
    p.be.sketch_haplotype = o_sketch_hap;
    p.be.dp_solve_diploid = o_dp;
    p.be.free_buf = orc_free;
    p.be.last_error = o_err;
    std::string err;
    if (p.run(err) != 0) { if (err == "dump_only") return 0; fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
    if (!json.empty()) {
        FILE *f = fopen(json.c_str(), "w");
        fprintf(f, "{\"dp_value\": %d, \"s_het\": %d, \"r1\": %d, \"r2\": %d, \"obj\": %d, \"len1\": %lld, \"len2\": %lld, "
                   "\"spectrum\": %lld, \"n_levels\": %lld, \"n_vertices\": %lld, \"cells\": %llu, \"relaxations\": %llu, "
                   "\"best_r_haploid\": %d, \"fit_nll\": %.17g}\n",
                p.sum.dp_value, p.sum.s_het, p.sum.r1, p.sum.r2, p.sum.obj, (long long)p.sum.len1, (long long)p.sum.len2,
                (long long)p.sum.spectrum, (long long)p.sum.n_levels, (long long)p.sum.n_vertices,
                (unsigned long long)p.sum.cells, (unsigned long long)p.sum.relaxations, p.sum.best_r_haploid, p.sum.fit.nll);
        fclose(f);
    }
    return 0;
}
