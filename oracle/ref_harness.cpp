// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE: drives the REAL reference functions (linked from
// the reference's own objects, see oracle/Makefile `ref`) to produce known-answer vectors for
// tests/golden/.  Our code only; it includes the reference's headers from /root/reference/src at
// build time and copies nothing.
//
//   ref_harness hash KMER...                 -> hash128_to_64_ (solver.cpp:16-24), hex per line
//   ref_harness hashes K W  < seqs.txt       -> Solver::compute_hashes per line (sorted set, hex)
//   ref_harness minimizers K W < seqs.txt    -> Solver::index_kmers per line (emitted hashes in order)
//   ref_harness fit < hist.txt               -> KGFitterBO::fit + classify ("mult freq" per line)
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "solver.h"
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
