// dg_dp_bench -- the DP pass of the bench workload timed in a plain C++ process over the C ABI (include/dipgenie_hip.h), i.e. on the HIP
// runtime libdipgenie_hip.so was BUILT against.  bench.py times its steps in a process that imported torch first and therefore runs
// the library on torch's bundled runtime; this driver repeats the DP part of those steps (same .dpg, resident graph, W warm-up
// passes, K timed passes bracketed by host clocks around dg_dp_run, which synchronises) so that the two can be compared.
//   dg_dp_bench graph.dpg [warmup [passes [device]]]   ->  one JSON line
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/dipgenie_hip.h"

template <class T>
static bool read_arr(FILE *f, std::vector<T> &v) {
    unsigned long long n = 0;
    if (fread(&n, 8, 1, f) != 1) return false;
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: dg_dp_bench graph.dpg [warmup [passes [device]]]\n"); return 1; }
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
    const int warm = argc > 2 ? atoi(argv[2]) : 1, passes = argc > 3 ? atoi(argv[3]) : 5, device = argc > 4 ? atoi(argv[4]) : 0;
    FILE *f = fopen(argv[1], "rb");
    char magic[8];
    int32_t R = 0;
    std::vector<int32_t> level_off, out_dst, hom_col, het_col;
    std::vector<int64_t> out_off, hom_off, het_off;
    std::vector<uint8_t> out_w;
    if (!f || fread(magic, 1, 8, f) != 8 || memcmp(magic, "DGDP0001", 8) || fread(&R, 4, 1, f) != 1 || !read_arr(f, level_off) || !read_arr(f, out_off) ||
        !read_arr(f, out_dst) || !read_arr(f, out_w) || !read_arr(f, hom_off) || !read_arr(f, hom_col) || !read_arr(f, het_off) || !read_arr(f, het_col)) {
        fprintf(stderr, "dg_dp_bench: cannot read %s\n", argv[1]);
        return 1;
    }
    fclose(f);
    static int32_t dummy = 0;
    dg_dp_graph g{};
    g.n_vertices = (int32_t)out_off.size() - 1; g.n_levels = (int32_t)level_off.size() - 1; g.R = R;
    g.level_off = level_off.data(); g.out_off = out_off.data(); g.out_dst = out_dst.data(); g.out_w = out_w.data();
    g.hom_off = hom_off.data(); g.het_off = het_off.data();
    g.hom_col = hom_col.empty() ? &dummy : hom_col.data(); g.het_col = het_col.empty() ? &dummy : het_col.data();
    dg_ctx *ctx = dg_create(device);
    if (!ctx) { fprintf(stderr, "dg_dp_bench: %s\n", dg_last_error()); return 2; }   // no gfx950 device: no CPU fallback
    if (dg_dp_load_graph(ctx, &g) != DG_OK) { fprintf(stderr, "dg_dp_bench: %s\n", dg_last_error()); return 1; }
    std::vector<int32_t> buf(4 * (size_t)(2 * R + 32));
    dg_dp_result res{};
    res.cap = 2 * R + 32;
    res.p1_from = buf.data(); res.p1_to = buf.data() + res.cap; res.p2_from = buf.data() + 2 * res.cap; res.p2_to = buf.data() + 3 * res.cap;
    for (int q = 0; q < warm; ++q)
        if (dg_dp_run(ctx, &res) != DG_OK) { fprintf(stderr, "dg_dp_bench: %s\n", dg_last_error()); return 1; }
    double fwd = 0, tb = 0, dl = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < passes; ++q) {
        if (dg_dp_run(ctx, &res) != DG_OK) { fprintf(stderr, "dg_dp_bench: %s\n", dg_last_error()); return 1; }
        dg_dp_timing tm;
        dg_dp_get_timing(ctx, &tm);
        fwd += tm.forward_ms; tb += tm.traceback_ms; dl += tm.delta_ms;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int built = 0, bound = 0;
    dg_hip_versions(&built, &bound);
    printf("{\"dp_value\": %d, \"cells\": %llu, \"passes\": %d, \"warmup\": %d, \"cells_per_s\": %.6g, \"ms_per_pass\": %.3f, \"forward_ms\": %.3f, \"traceback_ms\": %.3f, "
           "\"delta_ms\": %.3f, \"hip_built_against\": %d, \"hip_bound\": %d}\n",
           res.value, (unsigned long long)res.cells, passes, warm, (double)res.cells * passes / wall, 1e3 * wall / passes, fwd / passes, tb / passes, dl / passes, built, bound);
    fflush(stdout);
    _exit(0);
}
