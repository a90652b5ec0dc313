// The product's backend for libdipgenie_run.so: every device loop goes through libdipgenie_hip.so (include/dipgenie_hip.h).
// No CPU fallback: without a gfx950 device dgr_solve fails with the HIP library's message.
#include "run_core.hpp"

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

int dgr_wire_backend(dgr_handle *H, std::string &err) {
    if (!H->ctx) H->ctx = dg_create(H->device);
    if (!H->ctx) { err = dg_last_error(); return -1; }
    dg::Pipeline &p = H->p;
    p.be.ctx = H->ctx;
    p.be.sketch_reads = b_sketch_reads; p.be.sketch_haplotype = b_sketch_hap; p.be.dp_solve_diploid = b_dp; p.be.dp_solve_haploid = b_hap;
    p.be.free_buf = dg_free; p.be.anchor_begin = b_anchor_begin; p.be.anchor_add_haplotype = b_anchor_add; p.be.anchor_finish = b_anchor_finish;
    p.be.anchor_add_haplotype_sketched = b_anchor_add_sk; p.be.hint_dp_soon = b_hint; p.be.last_error = b_err;
    return 0;
}

void dgr_unwire_backend(dgr_handle *H) {
    if (H->ctx) dg_destroy((dg_ctx *)H->ctx);
    H->ctx = nullptr;
}
