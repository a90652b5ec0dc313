// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE: drives the REAL reference functions (linked from
// the reference's own objects, see oracle/Makefile `ref`) to produce known-answer vectors for
// tests/golden/.  Our code only; it includes the reference's headers from /root/reference/src at
// build time and copies nothing.
//
//   ref_harness hash KMER...                 -> hash128_to_64_ (solver.cpp:16-24), hex per line
//   ref_harness hashes K W  < seqs.txt       -> Solver::compute_hashes per line (sorted set, hex)
//   ref_harness minimizers K W < seqs.txt    -> Solver::index_kmers per line (emitted hashes in order)
//   ref_harness fit < hist.txt               -> KGFitterBO::fit + classify ("mult freq" per line)
//   ref_harness anchors K W T THREADS GFA READS -> Solver::read_gfa + read_ip_reads + compute_and_classify_anchors
//                                               (solver.cpp:27-245, 449-887), then Anchor_hits (solver.h:84) one line per
//                                               occurrence "id hap v0,v1,...", followed by "homo <id>" lines of homo_bv
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "solver.h"
#include "gfa.h"
#include "Classifier.hpp"
#include "Fitter.hpp"

uint64_t hash128_to_64_(const std::string &str);   // solver.cpp:16

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    std::string mode = argv[1];
    if (mode == "hash") {
        for (int i = 2; i < argc; ++i) printf("%016lx\n", (unsigned long)hash128_to_64_(argv[i]));
        return 0;
    }
    if (mode == "hashes" || mode == "minimizers") {
        Solver s(nullptr);
        s.k_mer = atoi(argv[2]);
        s.window = atoi(argv[3]);
        std::string line;
        while (std::getline(std::cin, line)) {
            if (mode == "hashes") {
                std::string copy = line;
                auto S = s.compute_hashes(copy);
                bool first = true;
                for (auto h : S) { printf("%s%016lx", first ? "" : " ", (unsigned long)h); first = false; }
            } else {
                s.node_seq = {line};
                s.paths = {{0u}};
                s.top_order_map = {0};
                auto idx = s.index_kmers(0);
                bool first = true;
                for (auto &m : idx) { printf("%s%016lx", first ? "" : " ", (unsigned long)m.first); first = false; }
            }
            printf("\n");
        }
        return 0;
    }
    if (mode == "anchors" && argc >= 8) {
        gfa_t *g = gfa_read(argv[6]);
        if (!g) return 2;
        Solver s(g);
        s.read_gfa();
        s.k_mer = atoi(argv[2]); s.window = atoi(argv[3]); s.threshold = (float)atof(argv[4]); s.num_threads = atoi(argv[5]);
        s.bucket_bits = 14; s.max_occ = 5000; s.debug = false;
        std::vector<std::pair<std::string, std::string>> reads;
        s.read_ip_reads(reads, argv[7]);
        s.compute_and_classify_anchors(reads);
        FILE *out = argc > 8 ? fopen(argv[8], "w") : stdout;
        for (size_t id = 0; id < s.Anchor_hits.size(); ++id)
            for (size_t h = 0; h < s.Anchor_hits[id].size(); ++h)
                for (auto &occ : s.Anchor_hits[id][h]) {
                    if (occ.empty()) continue;
                    fprintf(out, "%zu %zu ", id, h);
                    for (size_t q = 0; q < occ.size(); ++q) fprintf(out, "%s%d", q ? "," : "", occ[q]);
                    fputc('\n', out);
                }
        for (size_t id = 0; id < s.homo_bv.size(); ++id) if (s.homo_bv[id]) fprintf(out, "homo %zu\n", id);
        if (out != stdout) fclose(out);
        return 0;
    }
    if (mode == "fit") {
        std::vector<HistBin> H;
        int m; double f; int maxm = 0;
        while (scanf("%d %lf", &m, &f) == 2) { H.push_back({m, f}); if (m > maxm) maxm = m; }
        KGFitOptions opt;                      // solver.cpp:777-782
        opt.max_copy = 10; opt.max_x_use = maxm; opt.u_hi = maxm; opt.fit_error = true; opt.fit_varw = true;
        auto res = KGFitterBO::fit(H, opt);
        KmerGenieDiploidLike M(res.P);
        printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", res.nll, res.P.u_v, res.P.sd_v, res.P.var_w,
               res.P.zp_copy, res.P.zp_copy_het, res.P.p_d, res.P.p_e, res.P.err_shape);
        for (int x = 1; x <= maxm; ++x) putchar(M.classify(x).label == KGPosterior::HOM ? 'O' : 'E');
        putchar('\n');
        return 0;
    }
    return 1;
}
