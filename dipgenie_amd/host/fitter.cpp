#include "fitter.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace dg {

namespace {

inline double derr_old_val(int c, double s) {            // Fitter.hpp:74-80
    if (c <= 0) return 0.0;
    double a = std::pow((double)c, -s);
    double b = std::pow((double)(c + 1), -s);
    double v = a - b;
    return (v > 0.0 ? v : 1e-300);
}
inline std::vector<double> zeta_weights(double zp, int C) {   // Fitter.hpp:81-86
    std::vector<double> w(C + 1, 0.0);
    double S = 0.0;
    for (int k = 1; k <= C; k++) { w[k] = 1.0 / std::pow((double)k, zp); S += w[k]; }
    for (int k = 1; k <= C; k++) w[k] /= S;
    return w;
}
inline double normal_pdf(double x, double mu, double sd) {    // Fitter.hpp:87-91
    double s = std::max(sd, 1e-12), z = (x - mu) / s;
    static const double INV = 0.3989422804014327;
    return INV / s * std::exp(-0.5 * z * z);
}
inline double f_hom_x(int x, double u_v, double sd_v, const std::vector<double> &zeta, int C) {   // :102-110
    double sum = 0.0;
    for (int copy = 1; copy <= C; ++copy) {
        double mu = copy * u_v;
        double sd = std::sqrt((double)copy) * sd_v;
        sum += zeta[copy] * normal_pdf(x, mu, sd);
    }
    return std::max(sum, 1e-300);
}
inline double f_het_x(int x, double u_v, double var_w, const std::vector<double> &zeta, int C) {  // :111-121
    double u_base = 0.5 * u_v;
    double sd_base = 0.5 * std::sqrt(std::max(var_w, 1e-12));
    double sum = 0.0;
    for (int copy = 1; copy <= C; ++copy) {
        double mu = copy * u_base;
        double sd = std::sqrt((double)copy) * sd_base;
        sum += zeta[copy] * normal_pdf(x, mu, sd);
    }
    return std::max(sum, 1e-300);
}

std::vector<double> grid_or_freeze(double lo, double hi, int k) {   // Fitter.hpp:364-378
    if (std::fabs(hi - lo) < 1e-12) return {lo};
    std::vector<double> v;
    if (k <= 1) { v.push_back((lo + hi) / 2.0); return v; }
    for (int i = 0; i < k; i++) {
        double t = (double)i / (double)(k - 1);
        v.push_back(lo + t * (hi - lo));
    }
    return v;
}

}  // namespace

KGFitResult kg_fit(const std::vector<HistBin> &rawH, int max_copy, int max_multiplicity, int n_threads) {
    // dense histogram 0..N (Fitter.hpp:209-212); opt.max_x_use = max_multiplicity
    int Nmax = 0;
    for (auto &b : rawH) Nmax = std::max(Nmax, b.multiplicity);
    const int N = std::min(Nmax, max_multiplicity);
    std::vector<double> H(N + 1, 0.0);
    for (auto &b : rawH) if (b.multiplicity <= N) H[b.multiplicity] += b.freq;
    std::vector<int> xs;                       // bins with y>0 (nll_hist skips y<=0, :135)
    for (int x = 1; x <= N; ++x) if (H[x] > 0) xs.push_back(x);
    const int nb = (int)xs.size();

    // KGFitOptions defaults (Fitter.hpp:25-46) with u_hi = max_multiplicity
    auto U = grid_or_freeze(1.0, (double)max_multiplicity, 7);
    auto SD = grid_or_freeze(0.5, 2.0, 7);
    auto VW = grid_or_freeze(0.71, 4.0, 5);
    auto ZP = grid_or_freeze(1.01, 4.0, 7);
    auto ZPH = grid_or_freeze(1.01, 4.0, 7);
    auto PD = grid_or_freeze(0.1, 1.0, 7);
    auto PE = grid_or_freeze(0.0, 0.1, 5);
    auto SS = grid_or_freeze(1.01, 4.0, 5);
    const int nU = U.size(), nSD = SD.size(), nVW = VW.size(), nZP = ZP.size(), nZPH = ZPH.size(),
              nPD = PD.size(), nPE = PE.size(), nS = SS.size();

    // component tables, one row of nb doubles per distinct parameter triple
    std::vector<std::vector<double>> zeta(nZP);
    for (int a = 0; a < nZP; ++a) zeta[a] = zeta_weights(ZP[a], max_copy);   // ZP == ZPH grid
    std::vector<double> Fhom((size_t)nU * nSD * nZP * nb), Fhet((size_t)nU * nVW * nZPH * nb), Fe((size_t)nS * nb);
    for (int iu = 0; iu < nU; ++iu)
        for (int isd = 0; isd < nSD; ++isd)
            for (int iz = 0; iz < nZP; ++iz)
                for (int b = 0; b < nb; ++b)
                    Fhom[(((size_t)iu * nSD + isd) * nZP + iz) * nb + b] = f_hom_x(xs[b], U[iu], SD[isd], zeta[iz], max_copy);
    for (int iu = 0; iu < nU; ++iu)
        for (int ivw = 0; ivw < nVW; ++ivw)
            for (int iz = 0; iz < nZPH; ++iz)
                for (int b = 0; b < nb; ++b)
                    Fhet[(((size_t)iu * nVW + ivw) * nZPH + iz) * nb + b] = f_het_x(xs[b], U[iu], VW[ivw], zeta[iz], max_copy);
    for (int is = 0; is < nS; ++is)
        for (int b = 0; b < nb; ++b) Fe[(size_t)is * nb + b] = derr_old_val(xs[b], SS[is]);
    std::vector<double> Y(nb);
    for (int b = 0; b < nb; ++b) Y[b] = H[xs[b]];

    // scan: loop order u,sd,vw,zp,zph,pd,pe,s (Fitter.hpp:391-404); parallel over (u,sd) chunks,
    // each keeping its first strict minimum; chunks are then reduced in order with strict '<'.
    const int nchunk = nU * nSD;
    struct Best { double nll; int iu, isd, ivw, izp, izph, ipd, ipe, is; };
    std::vector<Best> best(nchunk, Best{std::numeric_limits<double>::infinity(), 0, 0, 0, 0, 0, 0, 0, 0});
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (int c = 0; c < nchunk; ++c) {
        const int iu = c / nSD, isd = c % nSD;
        Best bb = best[c];
        std::vector<double> inner(nb);
        for (int ivw = 0; ivw < nVW; ++ivw)
        for (int izp = 0; izp < nZP; ++izp) {
            const double *fhom = &Fhom[(((size_t)iu * nSD + isd) * nZP + izp) * nb];
            for (int izph = 0; izph < nZPH; ++izph) {
                const double *fhet = &Fhet[(((size_t)iu * nVW + ivw) * nZPH + izph) * nb];
                for (int ipd = 0; ipd < nPD; ++ipd) {
                    const double pd = PD[ipd];
                    for (int b = 0; b < nb; ++b) inner[b] = pd * fhet[b] + (1.0 - pd) * fhom[b];   // :140
                    for (int ipe = 0; ipe < nPE; ++ipe) {
                        const double pe = PE[ipe];
                        for (int is = 0; is < nS; ++is) {
                            const double *fe = &Fe[(size_t)is * nb];
                            double nll = 0.0;
                            for (int b = 0; b < nb; ++b) {
                                double mix = pe * fe[b] + (1.0 - pe) * inner[b];       // :139-140
                                nll += -Y[b] * std::log(mix + 1e-300);                 // :141
                            }
                            if (nll < bb.nll) bb = Best{nll, iu, isd, ivw, izp, izph, ipd, ipe, is};   // :404
                        }
                    }
                }
            }
        }
        best[c] = bb;
    }
    Best g = best[0];
    for (int c = 1; c < nchunk; ++c) if (best[c].nll < g.nll) g = best[c];
    KGFitResult r;
    r.P.max_copy = max_copy;
    r.P.u_v = U[g.iu]; r.P.sd_v = SD[g.isd]; r.P.var_w = VW[g.ivw];
    r.P.zp_copy = ZP[g.izp]; r.P.zp_copy_het = ZPH[g.izph];
    r.P.p_d = PD[g.ipd]; r.P.p_e = PE[g.ipe]; r.P.err_shape = SS[g.is];
    r.nll = g.nll;
    return r;
}

bool kg_is_hom(const KGParams &P, int x) {               // Classifier.hpp:59-80
    auto zh = zeta_weights(P.zp_copy, P.max_copy);
    auto zt = zeta_weights(P.zp_copy_het, P.max_copy);
    double fe = derr_old_val(x, P.err_shape);
    double fhet = f_het_x(x, P.u_v, P.var_w, zt, P.max_copy);
    double fhom = f_hom_x(x, P.u_v, P.sd_v, zh, P.max_copy);
    double a = P.p_e * fe;
    double b = (1.0 - P.p_e) * P.p_d * fhet;
    double c = (1.0 - P.p_e) * (1.0 - P.p_d) * fhom;
    double Z = std::max(a + b + c, 1e-300);
    double phet = b / Z, phom = c / Z;
    return !(x == 1 || phet >= phom);
}

}  // namespace dg
