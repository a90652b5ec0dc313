// Score-delta precompute (approximator.cpp:604-624): delta[e_u][e_v] = inter + symd for every coloured transition.
// The deltas depend neither on r nor on other levels, so they are filled for a whole window of levels in one launch,
// off the sweep's dependency chain.
#include <algorithm>

#include "dg_dp_setops.hpp"

namespace dgi {

// Per in-edge (source -> destination), once per graph: which of the four colour lists that an edge pair can contribute are
// non-empty -- bit 0 Hom(source), bit 1 Hom(destination), bit 2 Het(source), bit 3 Het(destination).  Most edges of a
// coloured transition carry no colour at all; with the flags the delta kernel answers those pairs from two byte loads
// instead of sixteen offset loads, and pairs with ONE coloured edge from that edge's own score (self[]).
__global__ __launch_bounds__(64) void dp_edge_flags_kernel(const LevelDesc *__restrict__ descs, int L, const uint32_t *__restrict__ in_edge,
                                                           const int32_t *__restrict__ in_dst, ColourCsr col, uint8_t *__restrict__ flags,
                                                           uint16_t *__restrict__ self) {
    const int l = (int)blockIdx.x + 1;
    if (l >= L) return;
    const LevelDesc d = descs[l];
    for (int e = (int)threadIdx.x; e < d.T; e += 64) {
        const int u1 = d.a0 + (int)(in_edge[d.in_base + e] & 0x7FFFFFFFu), u2 = in_dst[d.in_base + e];
        const uint32_t f = (col.hom_off[u1 + 1] > col.hom_off[u1] ? 1u : 0u) | (col.hom_off[u2 + 1] > col.hom_off[u2] ? 2u : 0u) |
                           (col.het_off[u1 + 1] > col.het_off[u1] ? 4u : 0u) | (col.het_off[u2 + 1] > col.het_off[u2] ? 8u : 0u);
        flags[d.in_base + e] = (uint8_t)f;
        // score of this edge paired with an edge that carries no colour: |Hom(u1) n Hom(u2)| + |Het(u1) /\ Het(u2)|
        int sc = 0;
        if ((f & 3u) == 3u) {
            const int64_t a = col.hom_off[u1], b = col.hom_off[u2];
            sc += union2x2<false>(col.hom_col + a, (int)(col.hom_off[u1 + 1] - a), nullptr, 0, col.hom_col + b, (int)(col.hom_off[u2 + 1] - b), nullptr, 0);
        }
        if (f & 12u) {
            const int64_t a = col.het_off[u1], b = col.het_off[u2];
            sc += union2x2<true>(col.het_col + a, (int)(col.het_off[u1 + 1] - a), nullptr, 0, col.het_col + b, (int)(col.het_off[u2 + 1] - b), nullptr, 0);
        }
        self[d.in_base + e] = (uint16_t)sc;
    }
}

__global__ __launch_bounds__(256) void dp_delta_kernel(const LevelDesc *__restrict__ descs,
                                                       const int32_t *__restrict__ dtrans,      // coloured transition -> level
                                                       const int64_t *__restrict__ dblk_first,  // first block of each coloured transition
                                                       int n_dtrans, const uint32_t *__restrict__ in_edge,
                                                       const int32_t *__restrict__ in_dst, ColourCsr col,
                                                       uint16_t *__restrict__ delta /* biased like SweepArgs::delta */, int64_t block0,
                                                       const uint8_t *__restrict__ eflags, const uint16_t *__restrict__ eself) {
    __shared__ int s_t;
    if (threadIdx.x == 0) {   // binary search: last transition whose first block <= blockIdx.x
        int lo = 0, hi = n_dtrans - 1;
        const int64_t b = block0 + blockIdx.x;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (dblk_first[mid] <= b) lo = mid; else hi = mid - 1; }
        s_t = lo;
    }
    __syncthreads();
    const int t = s_t;
    const LevelDesc d = descs[dtrans[t]];
    const int64_t n = (int64_t)d.T * d.T;
    const int64_t first = (block0 + (int64_t)blockIdx.x - dblk_first[t]) * DELTA_PER_BLOCK;
    uint16_t *out = delta + d.delta_off;
    // Pass 1, eight entries per thread at a time (independent index arithmetic and flag loads go out back to back): pairs whose
    // four colour lists cannot contribute -- the vast majority -- are answered with 0 on the spot, the others are queued in
    // LDS.  Pass 2 works the queue off densely: the sorted-list merges are the expensive part (16 dependent loads and
    // data-dependent loops), and with one coloured in-edge in ten nearly every wave used to hold a few of them, so every
    // wave paid for them (34 ms on MHC-24; 2.6 ms with the merges switched off).
    __shared__ uint16_t s_q[DELTA_PER_BLOCK];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const bool small = n < ((int64_t)1 << 31);
    for (int q0 = 0; q0 < DELTA_PER_BLOCK / 256; q0 += 8) {
        int64_t ee[8];
        uint32_t f[8];                                                    // bit 8: entry exists; bits 0-7 / 16-23: flags of e_u / e_v
        int eu8[8], ev8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t e = first + (int64_t)(q0 + u) * 256 + threadIdx.x;
            const bool ok = e < n;
            const int64_t ec = ok ? e : 0;
            const int eu = small ? (int)((uint32_t)ec / (uint32_t)d.T) : (int)(ec / d.T);
            const int ev = (int)(ec - (int64_t)eu * d.T);
            ee[u] = e; eu8[u] = eu; ev8[u] = ev;
            f[u] = ok ? ((uint32_t)eflags[d.in_base + eu] | ((uint32_t)eflags[d.in_base + ev] << 16) | 0x100u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!(f[u] & 0x100u)) continue;
            const uint32_t fu = f[u] & 0xFFu, fv = f[u] >> 16;
            if ((fu | fv) == 0u) out[ee[u]] = 0;                            // no colour on either edge
            else if (fv == 0u) out[ee[u]] = eself[d.in_base + eu8[u]];      // one coloured edge: its own score
            else if (fu == 0u) out[ee[u]] = eself[d.in_base + ev8[u]];
            else if (eu8[u] <= ev8[u]) s_q[atomicAdd(&s_n, 1)] = (uint16_t)((q0 + u) * 256 + (int)threadIdx.x);   // both coloured: merge (pass 2);
            // the score is symmetric in (e_u, e_v) -- unions of the two sources' lists against unions of the two destinations' -- so the
            // pair with e_u > e_v is left to the workgroup that holds its mirror image, which writes both entries
        }
    }
    __syncthreads();
    const int nq = s_n;
    for (int i = (int)threadIdx.x; i < nq; i += 256) {
        const int64_t e = first + (int64_t)s_q[i];
        const int eu = small ? (int)((uint32_t)e / (uint32_t)d.T) : (int)(e / d.T);
        const int ev = (int)(e - (int64_t)eu * d.T);
        const uint32_t ff = (uint32_t)eflags[d.in_base + eu] | (uint32_t)eflags[d.in_base + ev];
        const uint32_t pu = in_edge[d.in_base + eu], pv = in_edge[d.in_base + ev];
        const int u1 = d.a0 + (int)(pu & 0x7FFFFFFFu), v1 = d.a0 + (int)(pv & 0x7FFFFFFFu);
        const int u2 = in_dst[d.in_base + eu], v2 = in_dst[d.in_base + ev];
        int sc = 0;
        if ((ff & 3u) == 3u) sc += score_inter(col, u1, v1, u2, v2);
        if ((ff & 12u) != 0u) sc += score_symd(col, u1, v1, u2, v2);
        out[e] = (uint16_t)sc;
        if (eu != ev) out[(int64_t)ev * d.T + eu] = (uint16_t)sc;
    }
}

void delta_launch_edge_flags(const DpState &S, hipStream_t s) {
    hipLaunchKernelGGL(dp_edge_flags_kernel, dim3((unsigned)std::max(S.L - 1, 1)), dim3(64), 0, s, S.d_descs.as<LevelDesc>(), S.L, S.d_in_edge.as<uint32_t>(),
                       S.d_in_dst.as<int32_t>(), colour_csr(S), S.d_eflag.as<uint8_t>(), S.d_eself.as<uint16_t>());
}

const uint16_t *delta_launch_window(DpState &S, int w, hipStream_t s) {
    const int t0 = S.dwin_t[w], t1 = S.dwin_t[w + 1];
    const uint16_t *biased = S.d_delta.as<uint16_t>();
    if (t1 > t0) {
        const int64_t b0 = S.dblk_first_host[t0], b1 = S.dblk_first_host[t1];
        const int64_t base_off = S.descs[S.dtrans_host[t0]].delta_off;          // global entry offset of the window's first matrix
        uint16_t *out = S.d_delta.as<uint16_t>() + DELTA_PAD - base_off;
        hipLaunchKernelGGL(dp_delta_kernel, dim3((unsigned)(b1 - b0)), dim3(256), 0, s, S.d_descs.as<LevelDesc>(), S.d_dtrans.as<int32_t>(),
                           S.d_dblk_first.as<int64_t>(), (int)S.dtrans_host.size(), S.d_in_edge.as<uint32_t>(), S.d_in_dst.as<int32_t>(), colour_csr(S),
                           out, b0, S.d_eflag.as<uint8_t>(), S.d_eself.as<uint16_t>());
        biased = out;
    }
    S.cur_win = w;
    return biased;
}

// The single-window case beside the sweep.  The deltas depend on nothing the sweep produces, and the sweep of level l needs the
// matrix of level l only: the first piece (the transitions of the first DELTA_HEAD levels) runs on the sweep's stream, the
// remaining pieces on a side stream, each followed by an event that the sweep waits for before it issues the piece's first
// level (dg_dp_run.hip: sweep_range).  The delta kernel fills ~35 levels' worth of matrices in the time the sweep takes for
// one level, so those waits are satisfied long before they are reached; they are what makes the overlap correct.
// MHC-24: 13.0 ms of delta before the sweep become 1.1-1.3 ms with a head of 2,000 levels (0.1 ms with 250 and fourfold growing pieces), the sweep beside the pieces runs 5 ms longer: pass 634 -> 626 ms
// (a lowest-priority side stream: 628-630 ms, no better).
constexpr int DELTA_PIECES = 8, DELTA_HEAD = 250;

const uint16_t *delta_launch_overlapped(DpState &S, hipStream_t s) {
    const int nt = (int)S.dtrans_host.size();
    const bool forced = S.delta_overlap == 2;                          // (tests: small graphs too)
    // (one DP state per device unless option side_stream says otherwise: with several, a side stream may share a hardware queue with
    // another state's sweep and its pieces would queue behind whole batches of that sweep)
    if (!S.delta_overlap || nt < 2 * DELTA_PIECES || (!forced && (S.L < 32000 || !S.side_stream_ok()))) return delta_launch_window(S, 0, s);
    const int head = forced ? std::max(2, S.L / 256) : DELTA_HEAD;
    // ONE side stream for these pieces and for the L2 table prefetcher (dg_dp_sweep.hip).  With a stream each, the prefetcher -- a kernel
    // that lives as long as its range's sweep -- and the pieces could meet in one hardware queue (HIP maps streams onto a few of them):
    // the sweep then waited for a piece that waited behind a prefetcher that waited for the sweep.  Seen with GPU_MAX_HW_QUEUES=8: 4.1 s
    // per sweep.  In one stream the pieces always precede the prefetcher.
    if (!S.pf_stream && hipStreamCreateWithFlags(&S.pf_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); S.pf_stream = nullptr; S.delta_overlap = 0; return delta_launch_window(S, 0, s); }
    if (S.delta_piece_ev.empty()) {
        S.delta_piece_ev.assign(DELTA_PIECES, nullptr);
        for (auto &e : S.delta_piece_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); S.delta_overlap = 0; return delta_launch_window(S, 0, s); }
    }
    hipStream_t side = S.pf_stream;
    // piece boundaries in transitions: pieces that grow fourfold from the head on (a piece must be finished before the sweep reaches
    // its first level: the sweep needs ~4 us per level, the delta kernel ~0.1-0.15 us), then equal shares of the rest
    std::vector<int> cut(DELTA_PIECES + 1, nt);
    cut[0] = 0;
    int lvl_cut = head;
    for (int k = 1; k <= 4; ++k, lvl_cut *= 4) {
        cut[k] = (int)(std::lower_bound(S.dtrans_host.begin(), S.dtrans_host.end(), lvl_cut) - S.dtrans_host.begin());
        cut[k] = std::max(cut[k - 1] + 1, std::min(cut[k], nt - (DELTA_PIECES - k)));
    }
    for (int k = 5; k < DELTA_PIECES; ++k) cut[k] = cut[4] + (int)((int64_t)(nt - cut[4]) * (k - 4) / (DELTA_PIECES - 4));
    const int64_t base_off = S.descs[S.dtrans_host[0]].delta_off;
    uint16_t *out = S.d_delta.as<uint16_t>() + DELTA_PAD - base_off;
    S.delta_piece_level.assign(DELTA_PIECES, 0);
    hipEvent_t &gate = S.delta_piece_ev[0];                           // the side stream starts after what precedes on s (previous pass's readers)
    (void)hipEventRecord(gate, s);
    (void)hipStreamWaitEvent(side, gate, 0);
    for (int k = 0; k < DELTA_PIECES; ++k) {
        const int t0 = cut[k], t1 = cut[k + 1];
        S.delta_piece_level[k] = t0 < nt ? S.dtrans_host[t0] : S.L;
        if (t1 <= t0) continue;
        const int64_t b0 = S.dblk_first_host[t0], b1 = S.dblk_first_host[t1];
        hipStream_t q = k == 0 ? s : side;
        hipLaunchKernelGGL(dp_delta_kernel, dim3((unsigned)(b1 - b0)), dim3(256), 0, q, S.d_descs.as<LevelDesc>(), S.d_dtrans.as<int32_t>(),
                           S.d_dblk_first.as<int64_t>(), nt, S.d_in_edge.as<uint32_t>(), S.d_in_dst.as<int32_t>(), colour_csr(S),
                           out, b0, S.d_eflag.as<uint8_t>(), S.d_eself.as<uint16_t>());
        if (k > 0) (void)hipEventRecord(S.delta_piece_ev[k], side);
    }
    S.delta_piece_next = 1;
    S.cur_win = 0;
    return out;
}

void delta_overlap_free(DpState &S) {                                  // (the side stream itself: sweep_prefetch_free, called first)
    for (auto &e : S.delta_piece_ev) if (e) (void)hipEventDestroy(e);
    S.delta_piece_ev.clear();
}

}  // namespace dgi
