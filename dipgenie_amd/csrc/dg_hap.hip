// libdipgenie_hip.so -- haploid (vertex, r) DP on the device.
//
// Replaces the scatter loop of Approximator::dp_approximation_solver (/root/reference/src/approximator.cpp:44-72):
//   for u ascending, r ascending, (v, w) in adjacency order:
//       if r + w <= R and dp[u][r] + |color[v]| > dp[v][r + w]:  dp[v][r + w] = ..., back = (u, r)
// with every state starting at 0 and back pointers -1 (:50-52: paths may start mid-graph -- quirk kept).
//
// Gather form: the vertices are in topological order (every edge u -> v has u < v), so dp[u][.] is final before any
// destination of u is touched, and dp[v][r2] = max(0, max over in-edges (u, w) of dp[u][r2 - w] + |color[v]|).  The
// strict '>' makes the FIRST candidate of the scatter order win: in-edges are stored sorted by (u asc, w desc -- the
// smaller source r comes first --, adjacency order asc) and a candidate replaces the running best only if strictly larger.
//
// Level-synchronous: vertices are listed by longest-path depth (ties by id); one persistent workgroup walks the
// levels with a barrier in between (a level holds a handful of vertices: a launch per level would cost ~3 us x 10^5-10^6
// levels; the barrier costs ~1 us).  One wave per vertex, lanes over r.  Low value by SURVEY.md s8f-4 (the host scatter
// loop takes 0.05-0.12 s on MHC_4); it exists so that -p1 runs its DP behind the same boundary as -p2.
#include <algorithm>
#include <cstring>

#include "dg_internal.hpp"

namespace dgi {

struct HapVertex { uint32_t e0, deg; int32_t ncol, id; };         // a vertex in level order: its in-edge slice, |color|, id

__global__ __launch_bounds__(1024) void hap_dp_kernel(const uint32_t *__restrict__ level_off, int n_levels, const HapVertex *__restrict__ vtx,
                                                      const uint32_t *__restrict__ in_src /* source | w << 31 */, int RP,
                                                      int32_t *__restrict__ dp, int32_t *__restrict__ back_vtx, int32_t *__restrict__ back_r) {
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), n_waves = (int)(blockDim.x >> 6);
    for (int l = 0; l < n_levels; ++l) {
        const uint32_t a = level_off[l], b = level_off[l + 1];
        for (uint32_t v = a + (uint32_t)wave; v < b; v += (uint32_t)n_waves) {
            const HapVertex hv = vtx[v];
            for (int r2 = lane; r2 < RP; r2 += 64) {
                int best = 0, bu = -1, br = -1;                            // :50-52
                for (uint32_t e = hv.e0; e < hv.e0 + hv.deg; ++e) {
                    const uint32_t s = in_src[e];
                    const int u = (int)(s & 0x7FFFFFFFu), r = r2 - (int)(s >> 31);
                    if (r >= 0) {                                           // :60  r + w <= R
                        const int cand = dp[(size_t)u * RP + r] + hv.ncol;
                        if (cand > best) { best = cand; bu = u; br = r; }   // strict: first arrival keeps ties
                    }
                }
                const size_t o = (size_t)hv.id * RP + r2;
                dp[o] = best; back_vtx[o] = bu; back_r[o] = br;
            }
        }
        __syncthreads();                                                    // the next level reads what this one wrote
    }
}

}  // namespace dgi

using namespace dgi;

extern "C" int dg_dp_solve_haploid(dg_ctx *c, const dg_hap_graph *g, int32_t *dp, int32_t *back_vtx, int32_t *back_r) {
    if (int rc = bind(c)) return rc;
    if (!g || !g->out_off || !g->n_colours || !dp || !back_vtx || !back_r) { set_error("dg_dp_solve_haploid: null argument"); return DG_ERR_ARG; }
    const int n = g->n_vertices, R = g->R, RP = R + 1;
    if (n < 1 || R < 0 || R > 4096) { set_error("dg_dp_solve_haploid: bad sizes (V=%d R=%d)", n, R); return DG_ERR_ARG; }
    const int64_t E = g->out_off[n];
    if (g->out_off[0] != 0 || E < 0 || E >= ((int64_t)1 << 31) || (E > 0 && (!g->out_dst || !g->out_w))) { set_error("dg_dp_solve_haploid: bad edge arrays"); return DG_ERR_ARG; }
    // longest-path depth in one pass (topological order), then renumber by (depth, id)
    std::vector<int32_t> depth(n, 0);
    int32_t max_depth = 0;
    for (int u = 0; u < n; ++u) {
        if (g->out_off[u + 1] < g->out_off[u]) { set_error("out_off not monotone at %d", u); return DG_ERR_ARG; }
        for (int64_t e = g->out_off[u]; e < g->out_off[u + 1]; ++e) {
            const int v = g->out_dst[e];
            if (v <= u || v >= n) { set_error("edge %d->%d: vertices must be in topological order", u, v); return DG_ERR_ARG; }
            if (g->out_w[e] > 1) { set_error("edge weight %d > 1", (int)g->out_w[e]); return DG_ERR_ARG; }
            depth[v] = std::max(depth[v], depth[u] + 1);
        }
        max_depth = std::max(max_depth, depth[u]);
    }
    const int L = max_depth + 1;
    std::vector<uint32_t> level_off((size_t)L + 1, 0);
    for (int v = 0; v < n; ++v) level_off[depth[v] + 1]++;
    for (int l = 0; l < L; ++l) level_off[l + 1] += level_off[l];
    // vertices in level order with their in-edges in scatter arrival order: (u asc, w desc, adjacency order asc)
    std::vector<int32_t> slot_of(n);
    std::vector<HapVertex> vtx(n);
    {
        std::vector<uint32_t> fill(level_off.begin(), level_off.end() - 1);
        for (int v = 0; v < n; ++v) { const uint32_t q = fill[depth[v]]++; slot_of[v] = (int32_t)q; vtx[q] = HapVertex{0, 0, g->n_colours[v], v}; }
    }
    for (int64_t e = 0; e < E; ++e) vtx[slot_of[g->out_dst[e]]].deg++;
    uint32_t run = 0;
    for (int q = 0; q < n; ++q) { vtx[q].e0 = run; run += vtx[q].deg; }
    std::vector<uint32_t> in_src((size_t)std::max<int64_t>(E, 1)), fill(n);
    for (int q = 0; q < n; ++q) fill[q] = vtx[q].e0;
    for (int u = 0; u < n; ++u)                                             // filled in (u asc, adjacency order asc)
        for (int64_t e = g->out_off[u]; e < g->out_off[u + 1]; ++e) in_src[fill[slot_of[g->out_dst[e]]]++] = (uint32_t)u | ((uint32_t)g->out_w[e] << 31);
    for (int q = 0; q < n; ++q) {       // parallel edges of one source: the weight-1 ones arrive first (smaller source r), stably
        uint32_t *a = in_src.data() + vtx[q].e0;
        for (uint32_t i = 0; i < vtx[q].deg;) {
            uint32_t j = i;
            while (j < vtx[q].deg && (a[j] & 0x7FFFFFFFu) == (a[i] & 0x7FFFFFFFu)) ++j;
            if (j - i > 1) std::stable_partition(a + i, a + j, [](uint32_t x) { return (x >> 31) != 0; });
            i = j;
        }
    }
    hipStream_t s = c->stream;
    const size_t N = (size_t)n * RP;
    DevBuf d_lvl, d_vtx, d_in, d_dp, d_bv, d_br;
    if (int rc = d_lvl.ensure(4 * level_off.size())) return rc;
    if (int rc = d_vtx.ensure(sizeof(HapVertex) * (size_t)n)) return rc;
    if (int rc = d_in.ensure(4 * in_src.size())) return rc;
    if (int rc = d_dp.ensure(4 * N)) return rc;
    if (int rc = d_bv.ensure(4 * N)) return rc;
    if (int rc = d_br.ensure(4 * N)) return rc;
    DG_HIP(hipMemcpyAsync(d_lvl.p, level_off.data(), 4 * level_off.size(), hipMemcpyHostToDevice, s));
    DG_HIP(hipMemcpyAsync(d_vtx.p, vtx.data(), sizeof(HapVertex) * (size_t)n, hipMemcpyHostToDevice, s));
    DG_HIP(hipMemcpyAsync(d_in.p, in_src.data(), 4 * in_src.size(), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(hap_dp_kernel, dim3(1), dim3(1024), 0, s, d_lvl.as<uint32_t>(), L, d_vtx.as<HapVertex>(), d_in.as<uint32_t>(), RP, d_dp.as<int32_t>(),
                       d_bv.as<int32_t>(), d_br.as<int32_t>());
    DG_HIP(hipGetLastError());
    DG_HIP(hipMemcpyAsync(dp, d_dp.p, 4 * N, hipMemcpyDeviceToHost, s));          // tables are indexed by the caller's vertex ids
    DG_HIP(hipMemcpyAsync(back_vtx, d_bv.p, 4 * N, hipMemcpyDeviceToHost, s));
    DG_HIP(hipMemcpyAsync(back_r, d_br.p, 4 * N, hipMemcpyDeviceToHost, s));
    DG_HIP(hipStreamSynchronize(s));
    return DG_OK;
}
