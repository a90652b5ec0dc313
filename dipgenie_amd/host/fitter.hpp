// HOM/HET k-mer multiplicity model: grid fit + posterior classification.
//
// Host-side (CPU, double precision) restatement of
//   KGFitterBO::fit                 /root/reference/src/Fitter.hpp:205-408 (grid branch :362-406)
//   nll_hist / f_hom_x / f_het_x    /root/reference/src/Fitter.hpp:74-144
//   KmerGenieDiploidLike::classify  /root/reference/src/Classifier.hpp:59-80
// The reference evaluates 2,100,875 grid points x N bins x 20 exp(); here the three mixture
// components are tabulated once per distinct (u,sd,zp) / (u,var_w,zph) / (s) triple -- each table
// entry is computed with exactly the reference's expression, so every grid point's NLL is the
// bit-identical double -- and the grid is scanned with the same loop order and strict '<' rule
// (first minimum wins).  This TU must be compiled with -ffp-contract=off.
#pragma once
#include <cstdint>
#include <vector>

namespace dg {

struct KGParams {                      // Classifier.hpp:16-31 (fields used on this path)
    double zp_copy = 1.3, zp_copy_het = 1.3, u_v = 4.0, sd_v = 1.2, var_w = 2.0, p_d = 0.5;
    int max_copy = 5;
    double p_e = 0.01, err_shape = 2.0;
};

struct HistBin { int multiplicity; double freq; };

struct KGFitResult { KGParams P; double nll; };

// opt: max_copy=10, max_x_use=u_hi=max_multiplicity, fit_error=fit_varw=true (solver.cpp:777-782)
KGFitResult kg_fit(const std::vector<HistBin> &hist, int max_copy, int max_multiplicity, int n_threads);

// true = HOM, false = HET  (Classifier.hpp:59-80)
bool kg_is_hom(const KGParams &P, int multiplicity);

}  // namespace dg
