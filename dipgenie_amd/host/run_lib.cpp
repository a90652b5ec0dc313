// libdipgenie_run.so -- the host pipeline behind include/dipgenie_run.h (sharded runs, BASELINE configs[3]).
// Backend-agnostic: DGR_BACKEND_HIP wires the Backend table to libdipgenie_hip.so (product); the test harness compiles the
// same file with DGR_BACKEND_ORACLE against oracle/liboracle.so (tests/harness/Makefile) -- never shipped.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/dipgenie_run.h"
#include "pipeline.hpp"

#if defined(DGR_BACKEND_ORACLE)
#include "../../oracle/oracle.h"
#elif !defined(DGR_BACKEND_HIP)
#error "define DGR_BACKEND_HIP (product) or DGR_BACKEND_ORACLE (test harness)"
#endif

static_assert(sizeof(dgr_options) == 56 && sizeof(dgr_summary) == 88, "layouts mirrored by dipgenie_amd/run_sharded.py");

namespace {
thread_local std::string g_err;
int fail(const std::string &m) { g_err = m; return -1; }
}  // namespace

struct dgr_handle {
    dg::Pipeline p;
    std::string hap_buf;
    int32_t hap_buf_h = -1;
    std::string read_bases;
    std::vector<int64_t> read_off;
    void *ctx = nullptr;              // dg_ctx (HIP backend)
    int device = 0;
};

#if defined(DGR_BACKEND_HIP)
static int b_sketch_reads(void *c, const char *b, const int64_t *off, int64_t n, int k, int w, uint64_t **h, int32_t **cnt, int64_t *nd) { return dg_sketch_reads((dg_ctx *)c, b, off, n, k, w, h, cnt, nd); }
static int b_sketch_hap(void *c, const char *s, int64_t len, int k, int w, uint64_t **h, int64_t **p, int64_t *n) { return dg_sketch_haplotype((dg_ctx *)c, s, len, k, w, h, p, n); }
static int b_dp(void *c, const dg_dp_graph *g, dg_dp_result *r) { return dg_dp_solve_diploid((dg_ctx *)c, g, r); }
static int b_hap(void *c, const dg_hap_graph *g, int32_t *dp, int32_t *bv, int32_t *br) { return dg_dp_solve_haploid((dg_ctx *)c, g, dp, bv, br); }
static int b_anchor_begin(void *c, int32_t nh, int32_t nv, const int32_t *top, int k, int w) { return dg_anchor_begin((dg_ctx *)c, nh, nv, top, k, w); }
static int b_anchor_add(void *c, int32_t h, const char *s, int64_t len, const int32_t *sv, const int64_t *ss, int64_t ns, int64_t *n) { return dg_anchor_add_haplotype((dg_ctx *)c, h, s, len, sv, ss, ns, n); }
static int b_anchor_add_sk(void *c, int32_t h, int64_t len, const uint64_t *hash, const int64_t *pos, int64_t n, const int32_t *sv, const int64_t *ss, int64_t ns) {
    return dg_anchor_add_haplotype_sketched((dg_ctx *)c, h, len, hash, pos, n, sv, ss, ns);
}
static int b_anchor_finish(void *c, const uint64_t *sp, int64_t n, float thr, dg_anchor_result *out) { return dg_anchor_finish((dg_ctx *)c, sp, n, thr, out); }
static void b_hint(void *c, int64_t est_cells) {
    const double bytes = 2.0 * (double)est_cells;
    if (bytes >= 4e9) dg_dp_prealloc((dg_ctx *)c, bytes > 8e18 ? 0 : (int64_t)bytes);
}
static const char *b_err() { return dg_last_error(); }
#else
static int o_sketch_reads(void *, const char *b, const int64_t *off, int64_t n, int k, int w, uint64_t **h, int32_t **c, int64_t *nd) { return orc_sketch_reads(b, off, n, k, w, h, c, nd); }
static int o_sketch_hap(void *, const char *s, int64_t len, int k, int w, uint64_t **h, int64_t **p, int64_t *n) {
    const int64_t cnt = orc_minimizers(s, len, k, w, nullptr, nullptr, 0);
    *h = (uint64_t *)malloc(sizeof(uint64_t) * (cnt + 1));
    *p = (int64_t *)malloc(sizeof(int64_t) * (cnt + 1));
    *n = orc_minimizers(s, len, k, w, *h, *p, cnt);
    return 0;
}
static int o_dp(void *, const dg_dp_graph *g, dg_dp_result *r) { return orc_dp_solve_diploid((const orc_dp_graph *)g, (orc_dp_result *)r, nullptr); }
static const char *o_err() { return "oracle"; }
#endif

extern "C" const char *dgr_last_error(void) { return g_err.c_str(); }

extern "C" dgr_handle *dgr_open(const dgr_options *o) {
    if (!o || !o->gfa_file || !o->out_file) { fail("dgr_open: gfa_file and out_file are required"); return nullptr; }
    dgr_handle *H = new dgr_handle();
    dg::Options &opt = H->p.opt;
    opt.gfa_file = o->gfa_file; opt.reads_file = o->reads_file ? o->reads_file : ""; opt.hap_file = o->out_file;
    opt.threads = o->threads > 0 ? o->threads : 4; opt.ploidy = o->ploidy ? o->ploidy : 2; opt.R = o->R;
    opt.k = o->k ? o->k : 31; opt.w = o->w ? o->w : 25; opt.threshold = o->threshold > 0 ? o->threshold : 1.0f;
    opt.quiet = o->quiet != 0;
    H->device = o->device;
    std::string err;
    H->p.sum = dg::Summary();
    if (H->p.load_graph(err) != 0) { fail("dgr_open: " + err); delete H; return nullptr; }
    H->p.inj_hap.assign(H->p.num_walks, {});
    return H;
}

extern "C" void dgr_close(dgr_handle *H) {
    if (!H) return;
#if defined(DGR_BACKEND_HIP)
    if (H->ctx) dg_destroy((dg_ctx *)H->ctx);
#endif
    delete H;
}

extern "C" int32_t dgr_n_haplotypes(dgr_handle *H) { return H ? (int32_t)H->p.num_walks : -1; }

extern "C" int dgr_haplotype_sequence(dgr_handle *H, int32_t h, const char **seq, int64_t *len) {
    if (!H || !seq || !len || h < 0 || h >= (int32_t)H->p.num_walks) return fail("dgr_haplotype_sequence: bad arguments");
    if (H->hap_buf_h != h) { H->hap_buf = H->p.haplotype_sequence((uint32_t)h); H->hap_buf_h = h; }
    *seq = H->hap_buf.data(); *len = (int64_t)H->hap_buf.size();
    return 0;
}

extern "C" int dgr_load_reads(dgr_handle *H, int64_t *n_reads, const char **bases, const int64_t **read_off) {
    if (!H || !n_reads || !bases || !read_off) return fail("dgr_load_reads: bad arguments");
    if (H->p.opt.reads_file.empty()) return fail("dgr_load_reads: no reads file");
    if (H->read_off.empty()) {
        std::string err;
        if (H->p.load_reads(err) != 0) return fail("dgr_load_reads: " + err);
        H->read_off.assign(H->p.reads.size() + 1, 0);
        for (size_t r = 0; r < H->p.reads.size(); ++r) H->read_off[r + 1] = H->read_off[r] + (int64_t)H->p.reads[r].second.size();
        H->read_bases.reserve((size_t)H->read_off.back());
        for (auto &rd : H->p.reads) H->read_bases += rd.second;
        H->p.reads.clear(); H->p.reads.shrink_to_fit();
    }
    *n_reads = (int64_t)H->read_off.size() - 1; *bases = H->read_bases.data(); *read_off = H->read_off.data();
    return 0;
}

extern "C" int dgr_inject_haplotype_sketch(dgr_handle *H, int32_t h, const uint64_t *hash, const int64_t *pos, int64_t n) {
    if (!H || h < 0 || h >= (int32_t)H->p.num_walks || n < 0 || (n > 0 && (!hash || !pos))) return fail("dgr_inject_haplotype_sketch: bad arguments");
    auto &s = H->p.inj_hap[h];
    s.hash.assign(hash, hash + n); s.pos.assign(pos, pos + n); s.set = true;
    return 0;
}

extern "C" int dgr_inject_spectrum(dgr_handle *H, const uint64_t *sp_hash, const int32_t *sp_count, int64_t n, const int64_t *hist, int32_t n_bins) {
    if (!H || n < 0 || (n > 0 && (!sp_hash || !sp_count)) || (hist && n_bins < 2)) return fail("dgr_inject_spectrum: bad arguments");
    H->p.inj_sp_hash.assign(sp_hash, sp_hash + n); H->p.inj_sp_count.assign(sp_count, sp_count + n);
    H->p.inj_hist.clear();
    if (hist) H->p.inj_hist.assign(hist, hist + n_bins);
    H->p.spectrum_injected = true;
    return 0;
}

extern "C" int dgr_solve(dgr_handle *H, dgr_summary *out) {
    if (!H) return fail("dgr_solve: null handle");
    dg::Pipeline &p = H->p;
    if (!p.spectrum_injected && p.opt.reads_file.empty()) return fail("dgr_solve: no reads file and no injected spectrum");
#if defined(DGR_BACKEND_HIP)
    if (!H->ctx) H->ctx = dg_create(H->device);
    if (!H->ctx) return fail(std::string("dgr_solve: ") + dg_last_error());        // no gfx950 device: no CPU fallback
    p.be.ctx = H->ctx;
    p.be.sketch_reads = b_sketch_reads; p.be.sketch_haplotype = b_sketch_hap; p.be.dp_solve_diploid = b_dp; p.be.dp_solve_haploid = b_hap;
    p.be.free_buf = dg_free; p.be.anchor_begin = b_anchor_begin; p.be.anchor_add_haplotype = b_anchor_add; p.be.anchor_finish = b_anchor_finish;
    p.be.anchor_add_haplotype_sketched = b_anchor_add_sk; p.be.hint_dp_soon = b_hint; p.be.last_error = b_err;
#else
    p.be.sketch_reads = o_sketch_reads; p.be.sketch_haplotype = o_sketch_hap; p.be.dp_solve_diploid = o_dp; p.be.free_buf = orc_free; p.be.last_error = o_err;
#endif
    const double t0 = dg::now_s();
    std::string err;
    if (p.run_loaded(err) != 0) return fail("dgr_solve: " + err);
    if (out) {
        const dg::Summary &s = p.sum;
        *out = dgr_summary{s.dp_value, s.s_het, s.r1, s.r2, s.obj, s.len1, s.len2, s.spectrum, s.n_levels, s.n_vertices, s.cells, s.relaxations, dg::now_s() - t0};
    }
    return 0;
}
