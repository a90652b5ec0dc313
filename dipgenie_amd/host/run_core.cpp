// libdipgenie_run.so -- the host pipeline behind include/dipgenie_run.h (sharded runs, BASELINE configs[3]).
// This file knows no backend: the device loops are reached through the Backend table that dgr_wire_backend() fills
// (run_backend_hip.cpp: libdipgenie_hip.so, the only backend this package holds; the CPU tests link a checker-side
// backend of their own from tests/harness/).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/dipgenie_run.h"
#include "run_core.hpp"

static_assert(sizeof(dgr_options) == 56 && sizeof(dgr_summary) == 88, "layouts mirrored by dipgenie_amd/run_sharded.py");

namespace {
thread_local std::string g_err;
int fail(const std::string &m) { g_err = m; return -1; }
}  // namespace

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
    dgr_unwire_backend(H);
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
    // a spectrum entry is a hash held by >= 1 read, keys strictly ascending (solver.cpp:526-546): the counts index the label table later
    for (int64_t i = 0; i < n; ++i) {
        if (sp_count[i] < 1) return fail("dgr_inject_spectrum: count < 1 at entry " + std::to_string(i));
        if (i > 0 && sp_hash[i] <= sp_hash[i - 1]) return fail("dgr_inject_spectrum: hashes not strictly ascending at entry " + std::to_string(i));
    }
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
    std::string werr;
    if (dgr_wire_backend(H, werr) != 0) return fail("dgr_solve: " + werr);
    const double t0 = dg::now_s();
    std::string err;
    if (p.run_loaded(err) != 0) return fail("dgr_solve: " + err);
    if (out) {
        const dg::Summary &s = p.sum;
        *out = dgr_summary{s.dp_value, s.s_het, s.r1, s.r2, s.obj, s.len1, s.len2, s.spectrum, s.n_levels, s.n_vertices, s.cells, s.relaxations, dg::now_s() - t0};
    }
    return 0;
}
