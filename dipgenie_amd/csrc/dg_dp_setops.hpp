// Set operations on sorted colour lists (approximator.cpp:269-311) as device functions: 4-way merges that never build
// the unions.  Shared by the score-delta kernels and the traceback's s_het pass.
#pragma once
#include "dg_dp.hpp"

namespace dgi {

template <bool SYMDIFF>
__device__ __forceinline__ int union2x2(const int32_t *A, int na, const int32_t *B, int nb,
                                        const int32_t *C, int nc, const int32_t *D, int nd) {
    if (SYMDIFF) { if ((na | nb | nc | nd) == 0) return 0; }
    else { if ((na | nb) == 0 || (nc | nd) == 0) return 0; }
    int i = 0, j = 0, k = 0, m = 0, cnt = 0;
    while (i < na || j < nb || k < nc || m < nd) {
        int x = INT32_MAX;
        if (i < na) x = min(x, A[i]);
        if (j < nb) x = min(x, B[j]);
        if (k < nc) x = min(x, C[k]);
        if (m < nd) x = min(x, D[m]);
        bool inL = false, inR = false;
        while (i < na && A[i] == x) { inL = true; ++i; }
        while (j < nb && B[j] == x) { inL = true; ++j; }
        while (k < nc && C[k] == x) { inR = true; ++k; }
        while (m < nd && D[m] == x) { inR = true; ++m; }
        if (SYMDIFF ? (inL != inR) : (inL && inR)) ++cnt;
    }
    return cnt;
}


__device__ __forceinline__ int score_inter(const ColourCsr &c, int u1, int v1, int u2, int v2) {
    const int64_t a = c.hom_off[u1], b = c.hom_off[v1], d = c.hom_off[u2], e = c.hom_off[v2];
    return union2x2<false>(c.hom_col + a, (int)(c.hom_off[u1 + 1] - a), c.hom_col + b, (int)(c.hom_off[v1 + 1] - b),
                           c.hom_col + d, (int)(c.hom_off[u2 + 1] - d), c.hom_col + e, (int)(c.hom_off[v2 + 1] - e));
}
__device__ __forceinline__ int score_symd(const ColourCsr &c, int u1, int v1, int u2, int v2) {
    const int64_t a = c.het_off[u1], b = c.het_off[v1], d = c.het_off[u2], e = c.het_off[v2];
    return union2x2<true>(c.het_col + a, (int)(c.het_off[u1 + 1] - a), c.het_col + b, (int)(c.het_off[v1 + 1] - b),
                          c.het_col + d, (int)(c.het_off[u2 + 1] - d), c.het_col + e, (int)(c.het_off[v2 + 1] - e));
}

}  // namespace dgi
