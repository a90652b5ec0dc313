// Level sweep of the diploid DP (approximator.cpp:627-701), edge-pair form.  State layout [i][r][j] (j fastest).
//
// The work of one transition is the T x T grid of in-edge pairs (e_u, e_v) -- exactly the index space of the delta
// matrix.  One wave owns one task = (destination row i2, column group g, chunk of RC recombination counts): a group is
// a run of <= 64 consecutive in-edges e_v that covers whole destination columns (host-built), so lane <-> e_v and
// every destination cell's candidates sit in adjacent lanes.  The wave walks the row's in-edges e_u (wave-uniform),
// and per step every lane does one delta load (coalesced along e_v) and RC value loads (coalesced along the source
// column j) -- no per-lane inner loop, so a vertex with in-degree 24 costs 24 steps instead of 24 x 24.  Each lane
// keeps, per recombination count, the best candidate as the pair (value, ord) with ord built from the in-edge ranks so
// that a plain lexicographic max IS the reference's take-if order (value desc, pred_i asc, pred_j asc,
// approximator.cpp:657-659).  A log-step segmented max over lanes of equal destination column finishes the cell; the
// segment head stores the value and the back-pointer.
#include <algorithm>
#include <cmath>

#include "dg_dp_sweep_dev.hpp"

namespace dgi {

__device__ __forceinline__ uint32_t ord_word(int i, int j, int wu, int wv) {
    return ((uint32_t)(0x7FFF - i) << 17) | ((uint32_t)(0x7FFF - j) << 2) | ((uint32_t)wu << 1) | (uint32_t)wv;
}
__device__ __forceinline__ uint32_t bp_from_ord(uint32_t o) {
    const uint32_t i = 0x7FFFu - (o >> 17), j = 0x7FFFu - ((o >> 2) & 0x7FFFu);
    return i | (j << 15) | (((o >> 1) & 1u) << 30) | ((o & 1u) << 31);
}
// ---------------------------------------------------------------------------------------------
// generic form: group offsets read at run time; any in-degree, any level size, wide back-pointers
// ---------------------------------------------------------------------------------------------
template <int RC, bool DIGEST>
__device__ __forceinline__ void sweep_level_pairs(const SweepArgs &A, int lvl, int wave_id, int n_waves) {
    const LevelDesc d = A.descs[lvl];
    const int RP = A.RP;
    const int32_t *__restrict__ cur = (const int32_t *)(A.ring + (size_t)DG_SLOT(A, lvl - 1) * A.slot_bytes + A.pad_bytes);
    int32_t *__restrict__ nxt = (int32_t *)(A.ring + (size_t)DG_SLOT(A, lvl) * A.slot_bytes + A.pad_bytes);
    const int lane = threadIdx.x & 63;
    const int nchunk = (RP + RC - 1) / RC;
    const int64_t ntask = (int64_t)d.k2 * d.ngroups * nchunk;
    const bool has_delta = d.delta_off >= 0;
    const uint16_t *dm = has_delta ? A.delta + d.delta_off : A.delta_zero;   // (A.delta is biased by the resident delta window)
    const int dT = has_delta ? d.T : 0, dmask = has_delta ? -1 : 0;
    const uint32_t *gb = A.grp_begin + d.grp_first;
    unsigned long long dsum = 0;
    for (int64_t task = wave_id; task < ntask; task += n_waves) {
        const int g = (int)(task % d.ngroups);
        const int64_t rest = task / d.ngroups;
        const int rc = (int)(rest % nchunk), i2 = (int)(rest / nchunk);
        const int r0 = rc * RC;
        const uint32_t gbeg = gb[g], gend = gb[g + 1];
        const uint32_t eu0 = A.in_off[d.b0 + i2], eu1 = A.in_off[d.b0 + i2 + 1];
        int bval[RC];
        uint32_t bord[RC];
#pragma unroll
        for (int q = 0; q < RC; ++q) { bval[q] = NEG_INF; bord[q] = 0; }
        int j2 = -1 - lane;                                             // inactive lanes: unique negative ids
        // a group wider than 64 is one giant column (host guarantee): lanes accumulate over its chunks
        for (uint32_t cb = gbeg; cb < gend; cb += 64) {
            const uint32_t ev = cb + lane;
            const bool act = ev < gend;
            int j = 0, wv = 0, evr = 0;
            if (act) {
                const uint32_t pv = A.in_edge[ev];
                j = (int)(pv & 0x7FFFFFFFu); wv = (int)(pv >> 31);
                const int cv = A.in_dst[ev];
                j2 = cv - d.b0;
                if (!d.bp_wide) evr = (int)(ev - A.in_off[cv]);            // rank inside the column's in-edge list
            }
            const int dcol = (int)(ev - d.in_base) & dmask;
            for (uint32_t eu = eu0; eu < eu1; ++eu) {
                const uint32_t pu = A.in_edge[eu];
                const int i = (int)(pu & 0x7FFFFFFFu), wu = (int)(pu >> 31);
                if (act) {
                    const int w = wu + wv;
                    const int dl = (int)dm[(int64_t)(eu - d.in_base) * dT + dcol];
                    const uint32_t ord = d.bp_wide ? ord_word(i, j, wu, wv) : ord_rank((int)(eu - eu0), evr);
                    // rows r = r2 - w; the buffers carry front/tail padding so r = -1, -2 (and r2 >= RP in a
                    // ragged last chunk) are legal reads that the select discards: RC loads back to back
                    const int32_t *base = cur + ((int64_t)i * RP + (r0 - w)) * d.k + j;
                    int vals[RC];
#pragma unroll
                    for (int q = 0; q < RC; ++q) vals[q] = base[q * d.k];
#pragma unroll
                    for (int q = 0; q < RC; ++q) {
                        const int cand = vals[q] + dl;
                        const bool ok = (r0 + q < RP) & (r0 + q - w >= 0) & (vals[q] != NEG_INF);       // :633, :646-647
                        const bool take = ok & ((cand > bval[q]) | ((cand == bval[q]) & (ord > bord[q])));   // :657-659
                        bval[q] = take ? cand : bval[q];
                        bord[q] = take ? ord : bord[q];
                    }
                }
            }
        }
        // segmented max over lanes with equal destination column (lanes of a column are adjacent)
        const int span = (int)min(gend - gbeg, 64u);
        for (int s = 1; s < span; s <<= 1) {
            const int oj2 = __shfl_down(j2, s);
            const bool same = (lane + s < 64) & (oj2 == j2);
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const int ov = __shfl_down(bval[q], s);
                const uint32_t oo = (uint32_t)__shfl_down((int)bord[q], s);
                const bool take = same & ((ov > bval[q]) | ((ov == bval[q]) & (oo > bord[q])));
                bval[q] = take ? ov : bval[q];
                bord[q] = take ? oo : bord[q];
            }
        }
        const int pj2 = __shfl_up(j2, 1);
        const bool head = (j2 >= 0) & ((lane == 0) | (pj2 != j2));
        if (head) {
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const int r2 = r0 + q;
                if (r2 < RP) {
                    const int64_t idx = ((int64_t)i2 * RP + r2) * d.k2 + j2;
                    nxt[idx] = bval[q];
                    if (A.bp) {
                        if (d.bp_wide) __builtin_nontemporal_store((uint32_t)(bval[q] == NEG_INF ? BP_NONE : bp_from_ord(bord[q])), (uint32_t *)(A.bp + d.bp_off + 2 * idx));
                        else __builtin_nontemporal_store((uint16_t)~bord[q], &A.bp[d.bp_off + idx]);
                    }
                    if (DIGEST && bval[q] != NEG_INF) {
                        const unsigned long long o = ((unsigned long long)r2 * d.k2 + i2) * d.k2 + j2;   // oracle's r-major index
                        uint32_t pi, pj;
                        if (d.bp_wide) { const uint32_t h = bp_from_ord(bord[q]); pi = h & 0x7FFFu; pj = (h >> 15) & 0x7FFFu; }
                        else {
                            const uint32_t ru = BP_MAX_RANK - (bord[q] >> 8), rv = BP_MAX_RANK - (bord[q] & 0xFFu);
                            pi = A.in_edge[eu0 + ru] & 0x7FFFFFFFu;
                            pj = A.in_edge[A.in_off[d.b0 + j2] + rv] & 0x7FFFFFFFu;
                        }
                        dsum += digest_term(bval[q], o, pi, pj);
                    }
                }
            }
        }
        // destination columns without any in-edge are unreachable: nobody owns them, clear them here
        if (g == 0 && d.ndead > 0) {
            for (int t = lane; t < d.ndead * RC; t += 64) {
                const int q = t % RC, c = A.dead_cols[d.dead_first + t / RC];
                if (r0 + q < RP) {
                    const int64_t idx = ((int64_t)i2 * RP + r0 + q) * d.k2 + c;
                    nxt[idx] = NEG_INF;
                    if (A.bp) {
                        if (d.bp_wide) __builtin_nontemporal_store((uint32_t)(BP_NONE), (uint32_t *)(A.bp + d.bp_off + 2 * idx));
                        else __builtin_nontemporal_store((uint16_t)0xFFFFu, &A.bp[d.bp_off + idx]);
                    }
                }
            }
        }
    }
    if (DIGEST && dsum) atomicAdd(&A.digest[lvl], dsum);
}

template <int RC, bool DIGEST>
__global__ __launch_bounds__(256) void dp_sweep_kernel(SweepArgs A, int lvl) {
    sweep_level_pairs<RC, DIGEST>(A, lvl, (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), (int)(gridDim.x * 4));
}

__global__ void dp_init_kernel(int32_t *cur, int RP) {   // level 0: k = 1, every r starts at 0 (:534-535)
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r < RP) cur[r] = 0;
}

// ---------------------------------------------------------------------------------------------
// Fast form of the same sweep.  A level's kernel starts with cold caches (kernel boundary), so its
// duration is a chain of dependent memory round trips (~0.5-1 us each) -- the fast form cuts the chain to
//   kernel arguments (LevelDesc by value)  ->  {row record, slot record, row in-edge words}  ->  {delta, RC values}  ->  stores
// * slot table: every column group is padded to exactly 64 records {j | wv<<15 | j2<<16, ev_local |
//   rank<<20 | steps<<28}, so a lane finds its in-edge without reading group offsets;
// * row record {first in-edge, in-degree, in-edge 0, in-edge 1} serves 97 % of the rows in one load;
// * rows with more in-edges read them, one per lane, from the level's row in-edge matrix (LevelDesc::rowx_*) whose
//   address needs no record -- the same load round -- and broadcast them with readlane; only levels without a matrix
//   (in-degree > 64, or beyond the matrix budget) fetch the list from in_edge[] after the row record, one round later;
// * the row's in-edges are processed U per step so U sets of RC loads are in flight;
// * the segmented max runs only ceil(log2(max column in-degree of the group)) steps.
// GENERAL variant (levels tagged fast_ok == 2): a column with in-degree > 64 (more than 64 haplotypes recombining into
// one vertex) spans several slot blocks that one wave walks in turn, and rows of any in-degree fetch their in-edges 64
// at a time.  Only sizes beyond the 3-D grid or 2^20 in-edges per level fall back to the generic kernel above.
// COOP: 0 = every row; 1 = rows with more than COOP_MIN in-edges return (they are done by the cooperative region of
// the same launch); 2 = cooperative: the four waves of the workgroup walk a quarter of the row's in-edges each, wave 0
// merges the partial bests through LDS and finishes the task.
// ---------------------------------------------------------------------------------------------

template <int RC>
__device__ __forceinline__ void relax_select(const int (&vals)[RC], int dl, uint32_t ord, int r0, int w, int RP,
                                             int (&bval)[RC], uint32_t (&bord)[RC]) {
#pragma unroll
    for (int q = 0; q < RC; ++q) {
        const int cand = vals[q] + dl;
        const bool ok = (r0 + q < RP) & (r0 + q - w >= 0) & (vals[q] != NEG_INF);                   // :633, :646-647
        const bool take = ok & ((cand > bval[q]) | ((cand == bval[q]) & (ord > bord[q])));           // :657-659
        bval[q] = take ? cand : bval[q];
        bord[q] = take ? ord : bord[q];
    }
}

__device__ __forceinline__ void merge_best(int ov, uint32_t oo, bool same, int &bv, uint32_t &bo) {
    const bool take = same & ((ov > bv) | ((ov == bv) & (oo > bo)));
    bv = take ? ov : bv;
    bo = take ? oo : bo;
}

// Measurement build only (-DDG_SWEEP_PROBE, tools/probe build; never in the product library): one wave per destination row
// stamps the 100 MHz real-time counter at the task's phase boundaries.
#ifdef DG_SWEEP_PROBE
#define DG_PROBE_BEGIN unsigned long long pq[6] = {0, 0, 0, 0, 0, 0}; \
    const bool probe_on = A.probe != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x >> 6) == 0 && (int)blockIdx.z < d.k2;
#define DG_PROBE(q) do { if (probe_on) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pq[q] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define DG_PROBE_END do { if (probe_on && (threadIdx.x & 63) == 0) { \
        atomicMin(&A.probe[(size_t)lvl * 8 + 0], pq[0]); atomicMax(&A.probe[(size_t)lvl * 8 + 1], pq[5]); \
        if ((int)blockIdx.z == d.k2 / 2) for (int q = 0; q < 6; ++q) A.probe[(size_t)lvl * 8 + 2 + q] = pq[q]; } } while (0)
#else
#define DG_PROBE_BEGIN
#define DG_PROBE(q) do { } while (0)
#define DG_PROBE_END do { } while (0)
#endif

template <int RC, bool DIGEST, bool GENERAL, int COOP>
__device__ __forceinline__ void sweep_task(const LevelHead &H, const FastArgs &A, const LevelDesc &d, __amdgpu_buffer_rsrc_t cur_rsrc,
                                           int32_t *__restrict__ nxt, int i2, int g, int r0, int lvl, int part = 0, uint2 *ex = nullptr) {
    const int lane = threadIdx.x & 63;
    const int RP = H.RP;
    DG_PROBE_BEGIN
    DG_PROBE(0);
    // first load round: every address below comes from kernel arguments and the block index alone
    const uint4 rr = H.rowrec_l[i2];                                    // {eu0, du, pu0, pu1}
    uint2 sl = H.slots_l[g * 64 + lane];
    const bool rowx = !GENERAL && H.rowx_stride > 0;
    uint32_t mypu = 0;                                                  // in-edge words of the row, one per lane
    if (rowx && lane < H.rowx_stride) mypu = H.rowx_l[i2 * H.rowx_stride + lane];
    // steps field: 0..6 = log2 steps of the segmented max; 15 = first block of a giant column (in-degree > 64: its
    // in-edges fill several consecutive blocks, all walked by THIS wave); 14 = continuation block (nothing to do)
    int steps = __builtin_amdgcn_readfirstlane((int)(sl.y >> 28));
    const int j2 = (sl.x != 0xFFFFFFFFu) ? (int)((sl.x >> 16) & 0x7FFFu) : -1 - lane;
    int nblk = 1;
    if (GENERAL) {
        if (steps == 14) return;                                        // (workgroup-uniform in the cooperative region: same g)
        if (steps == 15) {
            const int col = __builtin_amdgcn_readfirstlane(j2);
            nblk = ((int)H.rowrec_l[col].y + 63) >> 6;
            steps = 6;
        }
    }
    const uint16_t *dm = H.dm;
    const int dT = H.dT;
    const bool has_delta = dT != 0;
    const int du = (int)rr.y;
    if (COOP == 1 && du > COOP_MIN) return;
    DG_PROBE(1);
    const int t_lo = COOP == 2 ? (du * part) >> 2 : 0, t_hi = COOP == 2 ? (du * (part + 1)) >> 2 : du;   // this wave's share of the row's in-edges
    int bval[RC];
    uint32_t bord[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q) { bval[q] = NEG_INF; bord[q] = 0; }
    const int64_t erow0 = (int64_t)rr.x * dT;                           // (dm is biased by the level's first in-edge)
    const int rowbytes = H.k * 4;
    bool act = false;
    // Source rows are r = r2 - w.  Byte offsets are relative to the padded buffer start (the resource), so rows
    // r = -1, -2 land in the front padding and r2 >= RP (ragged last chunk) in the tail padding; the select
    // discards what is out of range, which lets every load be issued unconditionally, back to back.
    for (int blk = 0; blk < (GENERAL ? nblk : 1); ++blk) {
        if (GENERAL && blk > 0) sl = H.slots_l[(g + blk) * 64 + lane];
        const bool actb = sl.x != 0xFFFFFFFFu;
        act |= actb;
        const int j = (int)(sl.x & 0x7FFFu), wv = (int)((sl.x >> 15) & 1u);
        const int dcol = has_delta ? (int)(sl.y & 0x000FFFFFu) : 0;
        const int evr = (int)((sl.y >> 20) & 0xFFu);                   // rank of this lane's in-edge inside its column's list
        if (du <= 2) {
            // 97 % of the rows: both in-edges came with the row record, no loop, no extra load
            if (actb && du > 0) {
                const int ia = (int)(rr.z & 0x7FFFu), wa = (int)(rr.z >> 31) + wv;    // (bit 16: flag for the chain walk)
                const int ib = (int)(rr.w & 0x7FFFu), wb = (int)(rr.w >> 31) + wv;
                const int offa = ((ia * RP + (r0 - wa)) * H.k + j) * 4 + H.pad_bytes;
                const int offb = ((ib * RP + (r0 - wb)) * H.k + j) * 4 + H.pad_bytes;
                int va[RC], vb[RC];
                const int dla = (int)dm[erow0 + dcol];
                int dlb = 0;
#pragma unroll
                for (int q = 0; q < RC; ++q)
                    va[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offa + q * rowbytes, 0, 0);
                if (du == 2) {
                    dlb = (int)dm[erow0 + dT + dcol];
#pragma unroll
                    for (int q = 0; q < RC; ++q)
                        vb[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offb + q * rowbytes, 0, 0);
                }
                relax_select<RC>(va, dla, ord_rank(0, evr), r0, wa, RP, bval, bord);
                if (du == 2) relax_select<RC>(vb, dlb, ord_rank(1, evr), r0, wb, RP, bval, bord);
            }
        } else {
            // heavy rows (recombination fan-in): U in-edges per step -- all their loads (U deltas + U*RC values) go out
            // back to back, then the selects run; (value, ord) max is associative and commutative, so the order inside
            // a step is irrelevant.  Small RC leaves registers for a deep step: in-degree 23 takes 2 steps at RC = 1.
            constexpr int U = GENERAL ? (RC >= 4 ? 2 : 4) : (RC >= 8 ? 2 : (RC >= 4 ? 4 : 8));
            // lean variant: one trip (in-degree <= 64, lane t of mypu = in-edge t); general: 64 in-edges per trip
            for (int c0 = GENERAL ? t_lo : 0; c0 < t_hi; c0 += GENERAL ? 64 : (1 << 30)) {
                const int tb = GENERAL ? c0 : t_lo, te = GENERAL ? min(c0 + 64, t_hi) : t_hi;
                if (!rowx) { mypu = 0; if (c0 + lane < te) mypu = A.in_edge[rr.x + c0 + lane]; }
                for (int t = tb; t < te; t += U) {
                    // the in-edge words live in SGPRs only until the load offset is formed; the select needs just their
                    // weight bits, kept in one mask (a deep step would otherwise hold U scalars and spill)
                    uint32_t wmask = 0;
                    if (actb) {
                        int vals[U][RC], dl[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (t + u < te) {                                 // wave-uniform
                                const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)mypu, t + u - c0);
                                wmask |= (p >> 31) << u;
                                const int iu = (int)(p & 0x7FFFFFFFu), w = (int)(p >> 31) + wv;
                                const int off = ((iu * RP + (r0 - w)) * H.k + j) * 4 + H.pad_bytes;
                                dl[u] = (int)dm[erow0 + (int64_t)(t + u) * dT + dcol];
#pragma unroll
                                for (int q = 0; q < RC; ++q)
                                    vals[u][q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, off + q * rowbytes, 0, 0);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (t + u < te) {
                                const int wu = (int)((wmask >> u) & 1u);
                                relax_select<RC>(vals[u], dl[u], ord_rank(t + u, evr), r0, wu + wv, RP, bval, bord);
                            }
                        }
                    }
                }
            }
        }
    }
    DG_PROBE(2);
    if (COOP == 2) {                                                    // partial bests of waves 1..3 -> wave 0
        if (part > 0) {
#pragma unroll
            for (int q = 0; q < RC; ++q) ex[((part - 1) * RC + q) * 64 + lane] = make_uint2((uint32_t)bval[q], bord[q]);
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const uint2 o = ex[(p * RC + q) * 64 + lane];
                merge_best((int)o.x, o.y, true, bval[q], bord[q]);
            }
        }
    }
    // segmented max over lanes with equal destination column (lanes of a column are adjacent)
    if (steps > 0) {                                                    // first step: distance 1, DPP
        const int oj2 = lane_down1(j2);
        const bool same = (lane + 1 < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) merge_best(lane_down1(bval[q]), (uint32_t)lane_down1((int)bord[q]), same, bval[q], bord[q]);
    }
    for (int st = 1, sh = 2; st < steps; ++st, sh <<= 1) {
        const int oj2 = __shfl_down(j2, sh);
        const bool same = (lane + sh < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) merge_best(__shfl_down(bval[q], sh), (uint32_t)__shfl_down((int)bord[q], sh), same, bval[q], bord[q]);
    }
    const int pj2 = lane_up1(j2);
    const bool head = act & ((lane == 0) | (pj2 != j2));
    unsigned long long dsum = 0;
    DG_PROBE(3);
    if (head) {
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int r2 = r0 + q;
            if (r2 < RP) {
                const int idx = (i2 * RP + r2) * d.k2 + j2;            // fast form: a state buffer is < 2 GB, 32-bit indices
                nxt[idx] = bval[q];                                    // (plain store: write-through costs +5 %, non-temporal +4.4 % on MHC-24)
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + idx], ~bord[q]); else A.bp[d.bp_off + idx] = (uint16_t)~bord[q]; }
                if (DIGEST && bval[q] != NEG_INF) {                    // (parity runs only: the extra loads are off the product path)
                    const unsigned long long o = ((unsigned long long)r2 * d.k2 + i2) * d.k2 + j2;   // oracle's r-major index
                    const uint32_t ru = BP_MAX_RANK - (bord[q] >> 8), rv = BP_MAX_RANK - (bord[q] & 0xFFu);
                    const uint32_t pi = A.in_edge[rr.x + ru] & 0x7FFFFFFFu;
                    const uint32_t pj = A.in_edge[H.rowrec_l[j2].x + rv] & 0x7FFFFFFFu;
                    dsum += digest_term(bval[q], o, pi, pj);
                }
            }
        }
    }
    if (g == 0 && d.ndead > 0) {                                        // columns nobody owns: unreachable
        for (int t = lane; t < d.ndead * RC; t += 64) {
            const int q = t % RC, c = A.dead_cols[d.dead_first + t / RC];
            if (r0 + q < RP) {
                const int64_t idx = ((int64_t)i2 * RP + r0 + q) * d.k2 + c;
                nxt[idx] = NEG_INF;
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + idx], 0xFFFFu); else A.bp[d.bp_off + idx] = (uint16_t)0xFFFFu; }
            }
        }
    }
    if (DIGEST && dsum) atomicAdd(&A.digest[lvl], dsum);
#ifdef DG_SWEEP_PROBE
    if (probe_on) pq[4] = __builtin_amdgcn_s_memrealtime();
    DG_PROBE(5);
    DG_PROBE_END;
#endif
}

#ifndef DG_PLAIN_WG
#define DG_PLAIN_WG 256
#endif
#define DG_PLAIN_GW ((DG_PLAIN_WG / 64) < 4 ? (DG_PLAIN_WG / 64) : 4)        // column blocks per workgroup
#define DG_PLAIN_CH ((DG_PLAIN_WG / 64) / DG_PLAIN_GW)                      // r chunks per workgroup
template <int RC, bool DIGEST, bool GENERAL>
__global__ __launch_bounds__(DG_PLAIN_WG) void dp_sweep_fast_kernel(const uint4 *rowrec_l, const uint2 *slots_l, const uint32_t *rowx_l, const int32_t *cur, const uint16_t *dm,
                                                            int rowx_stride, int nblocks, int rp_k, int pad_bytes, int dT, uint32_t buf_bytes,   // 16 dwords: preloaded
                                                            FastArgs A, LevelDesc d, int lvl) {
    const LevelHead H{rowrec_l, slots_l, rowx_l, cur, dm, rowx_stride, rp_k & 0x1FFF, rp_k >> 13, pad_bytes, dT, buf_bytes};
    publish_level(A.progress, lvl);
    const int wv = (int)(threadIdx.x >> 6);
    const int g = (int)blockIdx.x * DG_PLAIN_GW + (wv % DG_PLAIN_GW);    // always launched with DG_PLAIN_WG threads (reading blockDim would be a kernel-argument load)
    const int r0 = ((int)blockIdx.y * DG_PLAIN_CH + wv / DG_PLAIN_GW) * RC;
    if (g >= nblocks || r0 >= (rp_k & 0x1FFF)) return;                  // wave-uniform; no block barrier below
    int32_t *nxt = (int32_t *)(A.ring + (size_t)DG_SLOT(A, lvl) * A.slot_bytes + A.pad_bytes);
    sweep_task<RC, DIGEST, GENERAL, 0>(H, A, d, state_rsrc(cur, buf_bytes), nxt, (int)blockIdx.z, g, r0, lvl);
}

// the same with a cooperative region: workgroup z < 4 n_heavy = slot block 4 x + (z & 3) of heavy row z >> 2, its four waves
// walk a quarter of the row's in-edges each.  The region is dispatched FIRST: its tasks (load round + LDS merge) are the
// longest of the launch, and started last they were the launch's tail (-3.6 % on the MHC-24 sweep against z >= k2).
template <int RC, bool DIGEST, bool GENERAL>
__global__ __launch_bounds__(256) void dp_sweep_coop_kernel(const uint4 *rowrec_l, const uint2 *slots_l, const uint32_t *rowx_l, const int32_t *cur,
                                                            int rowx_stride, int nblocks, int n_heavy, int rp_k, unsigned long long heavy_lo,
                                                            unsigned long long heavy_hi,                                                  // 16 dwords: preloaded
                                                            FastArgs A, LevelDesc d, int lvl, const uint16_t *dm, int dT, const int32_t *__restrict__ heavy_rows) {
    const LevelHead H{rowrec_l, slots_l, rowx_l, cur, dm, rowx_stride, rp_k & 0x1FFF, rp_k >> 13, A.pad_bytes, dT, A.buf_bytes};
    publish_level(A.progress, lvl);
    int32_t *nxt = (int32_t *)(A.ring + (size_t)DG_SLOT(A, lvl) * A.slot_bytes + A.pad_bytes);
    const int r0 = (int)blockIdx.y * RC;                                // chunks of a row are neighbours in dispatch order: they share its delta row
    const int zc = 4 * n_heavy;
    if ((int)blockIdx.z < zc) {
        __shared__ uint2 ex[3 * RC * 64];
        const int hz = (int)blockIdx.z;
        const int g = (int)blockIdx.x * 4 + (hz & 3);
        if (g >= nblocks) return;                                       // workgroup-uniform: nobody is left at the barrier
        const int h = hz >> 2;
        int i2;                                                         // the first heavy rows ride in the kernel arguments (16 bits each): no dependent load
        if (h < 4) i2 = (int)((heavy_lo >> (16 * h)) & 0xFFFFu);
        else if (h < HEAVY_INLINE) i2 = (int)((heavy_hi >> (16 * (h - 4))) & 0xFFFFu);
        else i2 = heavy_rows[d.heavy_first + h];
        sweep_task<RC, DIGEST, GENERAL, 2>(H, A, d, state_rsrc(cur, A.buf_bytes), nxt, i2, g, r0, lvl, (int)(threadIdx.x >> 6), ex);
        return;
    }
    const int g = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (g >= nblocks) return;                                           // wave-uniform; no block barrier below
    sweep_task<RC, DIGEST, GENERAL, 1>(H, A, d, state_rsrc(cur, A.buf_bytes), nxt, (int)blockIdx.z - zc, g, r0, lvl);
}

// Sweep look-ahead: streams the graph tables (row records, slot records, in-edges, score deltas) of a batch of upcoming
// levels through the memory-side Infinity Cache.  Every table byte is read exactly once per pass, so without this each
// level's two dependent load rounds go all the way to HBM; the batch is a few MB, read at full chip bandwidth.
constexpr int WARM_RANGES = 5;
struct WarmRanges { const char *p[WARM_RANGES]; long long n16[WARM_RANGES]; };   // start (16-byte aligned down) and length in 16-byte units
__global__ __launch_bounds__(256) void dp_warm_tables_kernel(WarmRanges W) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
#pragma unroll
    for (int q = 0; q < WARM_RANGES; ++q) {
        const uint4 *p = (const uint4 *)W.p[q];
        for (long long i = tid; i < W.n16[q]; i += nth) { uint4 v = p[i]; asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
    }
}

// L2 table prefetcher.  The look-ahead above gets a level's tables as far as the memory-side Infinity Cache; the two load
// rounds of a task then cost an Infinity-Cache round trip each.  The L2 is per XCD, so this kernel -- PF_WORKGROUPS one-wave
// workgroups on a side stream, two per XCD by round-robin dispatch -- reads one word of every 128-byte line of the next
// levels' tables (row records, slot records, in-edges, row matrices, score deltas) on EVERY XCD, l2_prefetch levels ahead of
// the level the sweep publishes (workgroup 0 of every sweep launch stores its level).
// Measured on MHC-24 (sweep, ms): off 589; publishing alone 595; 8 / 16 / 64 workgroups 584.5 / 581.9 / 622-627 (64 pollers
// on the published word cost more than the prefetch gives: poll-only 638); 3 / 6 / 12 levels ahead 585.6 / 584.5 / 588.9.
// (A per-level prefetch LAUNCH in the chain made the traced sweep kernels 9 % shorter, 5.26 -> 4.79 us -- most of that was
// the extra launch absorbing the kernel boundary, not the cache.)
// Hints only: nothing waits for it; it leaves when the host marks the range done, when a newer range's sequence number
// appears, or after 0.3 ms without progress.
constexpr int PF_WORKGROUPS = 16, PF_FAR_WORKGROUPS = 16;
constexpr unsigned long long PF_IDLE_TICKS = 30000;      // 0.3 ms
struct PfCtl { int seq, stop, levels_done, probe, probe_ok, pad_[27]; int level; int pad2_[31]; };      // level: a line of its own

__global__ void dp_pf_ctl_kernel(PfCtl *c, int seq, int level, int stop) {
    __hip_atomic_store(&c->level, level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&c->stop, stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&c->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Does the side stream run beside the sweep's stream?  HIP maps streams onto a handful of hardware queues; a prefetcher that
// shares the sweep's queue would sit in front of the levels it is waiting for until it gives up.  Probed once per state: a
// kernel on the side stream waits up to ~2 ms (100 MHz counter) for a word that a kernel issued AFTER it on the sweep's
// stream sets -- it sees it only if the two really run concurrently.
__global__ void dp_pf_probe_wait_kernel(PfCtl *c, int token) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int ok = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < 200000ULL) {
        if (__hip_atomic_load(&c->probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == token) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    c->probe_ok = ok;
}
__global__ void dp_pf_probe_set_kernel(PfCtl *c, int token) { __hip_atomic_store(&c->probe, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64) void dp_l2_prefetch_kernel(const LevelDesc *__restrict__ descs, FastArgs F, PfCtl *ctl, int seq, int lb, int le, int ahead_near, int delta_resident, int far) {
    __shared__ int dump[64];
    // a load whose data nobody wants: straight into an LDS dump word per lane (no destination register, nothing to wait for)
#define DG_DROP_LOAD(PTR) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(PTR), (__attribute__((address_space(3))) void *)dump, 4, 0, 0)
    // blocks [0, PF_WORKGROUPS): the L2 role (every XCD reads every line of the level `ahead_near` levels down);
    // blocks beyond (far > 0): the Infinity-Cache role -- each line once chip-wide, `far` levels down (what the periodic
    // look-ahead launches in the level chain did)
    const bool far_role = (int)blockIdx.x >= PF_WORKGROUPS;
    const int lane = (int)threadIdx.x;
    const int slot = far_role ? (int)blockIdx.x - PF_WORKGROUPS : (int)(blockIdx.x >> 3), nslot = far_role ? (int)gridDim.x - PF_WORKGROUPS : PF_WORKGROUPS >> 3;
    const int ahead = far_role ? far : ahead_near, behind = far_role ? far - 64 : 0;      // the far role stays within [far - 64, far] levels of the sweep
    int last = INT32_MIN, done = 0;
    unsigned long long t_last = __builtin_amdgcn_s_memrealtime();
    int *level_word = &ctl->level;
    auto lines = [&](const char *base, long long bytes) {               // this workgroup's share of the 128-byte lines of [base, base + bytes)
        if (bytes <= 0) return;
        const uintptr_t a = (uintptr_t)base & ~(uintptr_t)127;
        const int n = (int)(((uintptr_t)base + (uintptr_t)bytes - a + 127) >> 7);
        for (int t = slot * 64 + lane; t < n; t += nslot * 64) DG_DROP_LOAD((const char *)a + ((size_t)t << 7));
    };
    int polls = 0;
    for (int lp = lb; lp < le;) {
        // one word is polled (the line the sweep's workgroup 0 stores to: every poll competes with that store); the range's
        // sequence number and stop flag, in another line, every 32nd time
        if ((polls++ & 31) == 0 &&
            (__hip_atomic_load(&ctl->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq || __hip_atomic_load(&ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
        const int lv = __builtin_amdgcn_readfirstlane(__hip_atomic_load(level_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        // Leaves after PF_IDLE_TICKS (100 MHz) without progress.  HIP maps streams and graph launches onto a few hardware queues;
        // if this kernel ever lands in front of the sweep in one queue, the sweep cannot start until it is gone (seen with
        // GPU_MAX_HW_QUEUES=8: every range waited out the former 10^5-poll limit, a 6.5x slower sweep).  0.3 ms per range bounds
        // that case at well under 1 % of a pass; a longer legitimate gap (the capturing pass) merely ends the prefetching of the range.
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (lv != last) { last = lv; t_last = now; } else if (now - t_last > PF_IDLE_TICKS) break;
        if (lp <= lv + behind) lp = lv + behind + 1;                    // overtaken (L2 role: level lv is running)
        if (lp >= le) break;
        if (lp - lv > ahead) { __builtin_amdgcn_s_sleep(48); continue; }
        const LevelDesc &d = descs[lp];
        const int k2 = __builtin_amdgcn_readfirstlane(d.k2), T = __builtin_amdgcn_readfirstlane(d.T);
        lines((const char *)(F.rowrec + d.b0), 16LL * k2);
        lines((const char *)(F.slots + d.slot_first), 512LL * d.nblocks);
        lines((const char *)(F.in_edge + d.in_base), 4LL * T);
        if (d.rowx_stride > 0) lines((const char *)(F.rowx + d.rowx_off), 4LL * k2 * d.rowx_stride);
        if (delta_resident && d.delta_off >= 0 && (long long)T * T <= (128 << 10)) lines((const char *)(F.delta + d.delta_off), 2LL * T * T);
        ++lp; ++done;
    }
    if (lane == 0 && blockIdx.x == 0) atomicAdd(&ctl->levels_done, done);
    (void)far_role;
#undef DG_DROP_LOAD
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void sweep_prepare(const DpState &S, SweepLaunch &X) {
    X.rc_sel = S.RP <= 8 ? 8 : (S.RP <= 19 ? 19 : 33);
    X.small_state = S.state_alloc_bytes < ((size_t)1 << 31);            // 32-bit buffer offsets inside a slot
    SweepArgs &A = X.A;
    A.descs = S.d_descs.as<LevelDesc>(); A.in_off = S.d_in_off.as<uint32_t>(); A.in_edge = S.d_in_edge.as<uint32_t>();
    A.grp_begin = S.d_grp.as<uint32_t>(); A.in_dst = S.d_in_dst.as<int32_t>(); A.dead_cols = S.d_dead.as<int32_t>();
    A.delta = A.delta_zero = S.d_delta.as<uint16_t>();
    A.ring = S.d_ring.as<char>(); A.slot_bytes = S.state_alloc_bytes; A.pad_bytes = 4 * (size_t)S.pad_front;
    A.bp = nullptr; A.digest = S.d_digest.as<unsigned long long>(); A.RP = S.RP;
    FastArgs &F = X.F;
    F.rowrec = S.d_rowrec.as<uint4>(); F.slots = S.d_slots.as<uint2>(); F.in_edge = A.in_edge; F.rowx = S.d_rowx.as<uint32_t>(); F.dead_cols = A.dead_cols;
    F.delta = F.delta_zero = A.delta; F.bp = nullptr; F.digest = A.digest; F.RP = S.RP;
    F.ring = S.d_ring.as<char>(); F.slot_bytes = (uint32_t)S.state_alloc_bytes;
#ifdef DG_SWEEP_PROBE
    F.probe = S.d_probe.as<unsigned long long>();
#endif
    F.progress = A.progress = &S.d_pfctl.as<PfCtl>()->level;
    F.pad_bytes = (int)(4 * S.pad_front);
    F.buf_bytes = (uint32_t)std::min<size_t>(S.state_alloc_bytes, 0x7FFFFFFFu);
}

void sweep_init_state(const DpState &S, hipStream_t s) {
    hipLaunchKernelGGL(dp_init_kernel, dim3((unsigned)((S.RP + 255) / 256)), dim3(256), 0, s, S.d_ring.as<int32_t>() + S.pad_front, S.RP);
}

// A lone wave retires ~1 instruction per 4-8 cycles, so an RC-fold unrolled task is the level's critical path: while
// the chip has idle wave slots, give each wave fewer recombination counts.  Cost model fitted on MHC-24 (R = 18) and
// the 100-walk chr22-style panel (R = 32):
//   T(RC) = max(1, W / cap) * (t0 + dmax * RC * tg) + W * tw,   W = tasks * ceil(RP / RC) waves.
// First factor: rounds of resident waves; second: a wave's dependent chain (the row with the largest in-degree walks
// dmax in-edges with RC gathers each); last: per-wave issue overhead.  Cooperative variant (RC <= 4): rows above
// COOP_MIN in-edges are walked by four waves, so the chain is a quarter (at least COOP_MIN) while four extra
// workgroup slots per heavy row are launched.
static void choose_rc(const DpState &S, const LevelDesc &d, int l, int rc_sel, int &rc, bool &coop) {
    rc = rc_sel;
    coop = false;
    if (S.adaptive_rc == 0) return;
    const int cand[11] = {1, 2, 3, 4, 5, 6, 8, 10, 11, 16, rc_sel};
    const bool coop_ok = S.use_coop && d.n_heavy > 0 && d.k2 + 4 * d.n_heavy <= 65535;
    const double dmax = (double)std::max(1, S.level_dmax[l]);
    double best = 1e300;
    for (int pass = (coop_ok && S.use_coop == 2) ? 1 : 0; pass < (coop_ok ? 2 : 1); ++pass) {     // coop = 2 (tests): whenever possible
        for (int q = 0; q < 11; ++q) {
            if (cand[q] > rc_sel || (q < 10 && cand[q] == rc_sel)) continue;
            if (pass == 1 && cand[q] > 4) continue;
            const double rows = pass ? (double)d.k2 + 4.0 * d.n_heavy : (double)d.k2;
            const double chain = pass ? std::max((double)COOP_MIN, std::ceil(dmax / 4.0)) + 1.0 : dmax;
            const double W = rows * d.nblocks * ((S.rp_active + cand[q] - 1) / cand[q]);
            const double T = std::max(1.0, W / (double)S.rc_cap) * ((double)S.rc_t0_ns + chain * cand[q] * (double)S.rc_tg_ps * 1e-3) + W * (double)S.rc_tw_ps * 1e-3;
            if (T <= best) { best = T; rc = cand[q]; coop = pass == 1; }   // ties: the larger RC (fewer waves)
        }
    }
}

void sweep_launch_level(DpState &S, SweepLaunch &X, int l, hipStream_t s) {
    LevelDesc &d = S.descs[l];
    // small and mid-sized levels end sooner with write-back back-pointer stores (3.6 vs 4.2 us per level on MHC_4; threshold
    // 16 K / 64 K / 256 K / 1 M / 4 M cells: MHC-24 sweep 580 / 575 / 574 / 579 / 586 ms), big ones with non-temporal ones that
    // keep the once-written lattice out of the L2
    d.bp_nt = (int64_t)d.k2 * d.k2 * S.RP >= S.bp_nt_min_cells ? 1 : 0;
#ifdef DG_SYM
    if (sweep_launch_sym(S, X, l, s)) return;                           // (measurement build only: dg_dp_sweep_sym.hip)
#endif
    if (d.fast_ok && X.small_state && S.RP <= 65535 && S.use_fast) {
        int rc;
        bool coop;
        choose_rc(S, d, l, X.rc_sel, rc, coop);
        S.launch_hist[(rc & 63) * 4 + (d.fast_ok == 2 ? 2 : 0) + (coop ? 1 : 0)]++;
        const int nch = (S.rp_active + rc - 1) / rc;              // (rp_active < RP: re-sweep of a segment whose path stays below that plane, dg_dp_run.hip)
        const dim3 grid((unsigned)((d.nblocks + 3) / 4), (unsigned)nch, (unsigned)(d.k2 + (coop ? 4 * d.n_heavy : 0)));
        const dim3 pgrid((unsigned)((d.nblocks + DG_PLAIN_GW - 1) / DG_PLAIN_GW), (unsigned)((nch + DG_PLAIN_CH - 1) / DG_PLAIN_CH), grid.z);
        const int32_t *hv = S.d_heavy.as<int32_t>();
        const FastArgs &F = X.F;
        const uint4 *rowrec_l = F.rowrec + d.b0;
        const uint2 *slots_l = F.slots + d.slot_first;
        const uint32_t *rowx_l = F.rowx + d.rowx_off;
        const int32_t *cur = (const int32_t *)(F.ring + (size_t)DG_SLOT(F, l - 1) * F.slot_bytes);
        const int dT = d.delta_off >= 0 ? d.T : 0;
        const uint16_t *dm = dT ? F.delta + d.delta_off - (int64_t)d.in_base * dT : F.delta_zero;   // (F.delta is biased by the resident delta window)
        const int rp_k = S.RP | (d.k << 13);
        unsigned long long hlo = 0, hhi = 0;
        for (int q = 0; q < 4; ++q) { hlo |= (unsigned long long)(uint16_t)d.heavy_in[q] << (16 * q); hhi |= (unsigned long long)(uint16_t)d.heavy_in[4 + q] << (16 * q); }
#define DG_FAST(RCV, DG) do { if (d.fast_ok == 2) hipLaunchKernelGGL((dp_sweep_fast_kernel<RCV, DG, true>), pgrid, dim3(DG_PLAIN_WG), 0, s, rowrec_l, slots_l, rowx_l, cur, dm, d.rowx_stride, d.nblocks, rp_k, F.pad_bytes, dT, F.buf_bytes, F, d, l); \
                              else hipLaunchKernelGGL((dp_sweep_fast_kernel<RCV, DG, false>), pgrid, dim3(DG_PLAIN_WG), 0, s, rowrec_l, slots_l, rowx_l, cur, dm, d.rowx_stride, d.nblocks, rp_k, F.pad_bytes, dT, F.buf_bytes, F, d, l); } while (0)
#define DG_COOP(RCV, DG) do { if (d.fast_ok == 2) hipLaunchKernelGGL((dp_sweep_coop_kernel<RCV, DG, true>), grid, dim3(256), 0, s, rowrec_l, slots_l, rowx_l, cur, d.rowx_stride, d.nblocks, d.n_heavy, rp_k, hlo, hhi, F, d, l, dm, dT, hv); \
                              else hipLaunchKernelGGL((dp_sweep_coop_kernel<RCV, DG, false>), grid, dim3(256), 0, s, rowrec_l, slots_l, rowx_l, cur, d.rowx_stride, d.nblocks, d.n_heavy, rp_k, hlo, hhi, F, d, l, dm, dT, hv); } while (0)
#define DG_FAST_RC(DG) do { if (coop) { switch (rc) { case 1: DG_COOP(1, DG); break; case 2: DG_COOP(2, DG); break; case 3: DG_COOP(3, DG); break; \
                                                    default: DG_COOP(4, DG); break; } break; } \
                            switch (rc) { case 1: DG_FAST(1, DG); break; case 2: DG_FAST(2, DG); break; case 3: DG_FAST(3, DG); break; \
                                        case 4: DG_FAST(4, DG); break; case 5: DG_FAST(5, DG); break; case 6: DG_FAST(6, DG); break; \
                                        case 8: DG_FAST(8, DG); break; case 10: DG_FAST(10, DG); break; case 11: DG_FAST(11, DG); break; \
                                        case 16: DG_FAST(16, DG); break; case 19: DG_FAST(19, DG); break; \
                                        default: DG_FAST(33, DG); break; } } while (0)
        if (S.want_digest) DG_FAST_RC(true); else DG_FAST_RC(false);
#undef DG_COOP
#undef DG_FAST_RC
#undef DG_FAST
    } else {
        const int nchunk = (S.RP + X.rc_sel - 1) / X.rc_sel;
        const int64_t ntask = (int64_t)d.k2 * d.ngroups * nchunk;
        const unsigned grid = (unsigned)std::min<int64_t>((ntask + 3) / 4, S.max_blocks);
        const SweepArgs &A = X.A;
        S.launch_hist[0]++;
#define DG_SWEEP(RCV, DG) hipLaunchKernelGGL((dp_sweep_kernel<RCV, DG>), dim3(grid), dim3(256), 0, s, A, l)
        if (S.want_digest) { if (X.rc_sel == 8) DG_SWEEP(8, true); else if (X.rc_sel == 19) DG_SWEEP(19, true); else DG_SWEEP(33, true); }
        else { if (X.rc_sel == 8) DG_SWEEP(8, false); else if (X.rc_sel == 19) DG_SWEEP(19, false); else DG_SWEEP(33, false); }
#undef DG_SWEEP
    }
}

void sweep_warm_tables(const DpState &S, const SweepLaunch &X, int q0, int q1, hipStream_t s) {   // graph tables of destination levels [q0, q1) -> Infinity Cache
    const LevelDesc &da = S.descs[q0], &db = S.descs[q1 - 1];
    const FastArgs &F = X.F;
    WarmRanges W{};
    auto put = [&](int q, const void *base, int64_t b0, int64_t b1) {       // byte range [b0, b1) behind base
        const uintptr_t a = ((uintptr_t)base + (uintptr_t)b0) & ~(uintptr_t)15, e = ((uintptr_t)base + (uintptr_t)b1) & ~(uintptr_t)15;
        W.p[q] = (const char *)a; W.n16[q] = e > a ? (long long)((e - a) >> 4) : 0;
    };
    put(0, F.rowrec, 16 * (int64_t)da.b0, 16 * ((int64_t)db.b0 + db.k2));
    put(1, F.slots, 8 * da.slot_first, 8 * (db.slot_first + (int64_t)db.nblocks * 64));
    put(2, F.in_edge, 4 * (int64_t)da.in_base, 4 * ((int64_t)db.in_base + db.T));
    int64_t d0 = -1, d1 = -1, x0 = -1, x1 = -1;
    for (int q = q0; q < q1; ++q) {
        const LevelDesc &dq = S.descs[q];
        if (dq.rowx_stride > 0) {                                           // (matrices are laid out in level order)
            if (x0 < 0) x0 = dq.rowx_off;
            x1 = dq.rowx_off + (int64_t)dq.k2 * dq.rowx_stride;
        }
        if (dq.delta_off < 0 || S.level_win[q] != S.cur_win) continue;
        const int64_t e = dq.delta_off + (int64_t)dq.T * dq.T;
        d0 = d0 < 0 ? dq.delta_off : std::min(d0, dq.delta_off);
        d1 = std::max(d1, e);
    }
    if (d0 >= 0) put(3, F.delta, 2 * d0, 2 * d1);
    if (x0 >= 0) put(4, F.rowx, 4 * x0, 4 * x1);
    long long tot = 0;
    for (int q = 0; q < WARM_RANGES; ++q) tot += W.n16[q];
    if (tot <= 0) return;
    const unsigned grid = (unsigned)std::min<long long>((tot + 255) / 256, 2048);
    hipLaunchKernelGGL(dp_warm_tables_kernel, dim3(grid), dim3(256), 0, s, W);
}

// L2 prefetcher of the sweep range [lb, le): control words set in stream order on s, the prefetcher itself on the side stream
int sweep_prefetch_begin(DpState &S, const SweepLaunch &X, int lb, int le, bool delta_resident, hipStream_t s) {
    S.pf_active = false;
    if (S.l2_prefetch <= 0 || !X.small_state || !S.use_fast || le - lb < 64) return DG_OK;
    if (!S.pf_stream && hipStreamCreateWithFlags(&S.pf_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); S.pf_stream = nullptr; S.l2_prefetch = 0; return DG_OK; }
    if (!S.pf_ev && hipEventCreateWithFlags(&S.pf_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); S.pf_ev = nullptr; S.l2_prefetch = 0; return DG_OK; }
    // One DP state per DEVICE (option side_stream: -1 = this rule, 0 / 1 = the caller decides): a second instance's sweep could share a
    // hardware queue with THIS state's prefetcher and would then wait behind it for a whole range (three concurrent instances on
    // one GPU: 78 -> 52 G cells/s before this rule).
    if (!S.side_stream_ok()) return DG_OK;
    if (!S.pf_tested) {
        S.pf_tested = 1;
        DG_HIP(hipStreamSynchronize(s));
        hipLaunchKernelGGL(dp_pf_probe_wait_kernel, dim3(1), dim3(1), 0, S.pf_stream, S.d_pfctl.as<PfCtl>(), 0x5EED);
        hipLaunchKernelGGL(dp_pf_probe_set_kernel, dim3(1), dim3(1), 0, s, S.d_pfctl.as<PfCtl>(), 0x5EED);
        PfCtl h;
        if (hipStreamSynchronize(S.pf_stream) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpy(&h, S.d_pfctl.p, sizeof h, hipMemcpyDeviceToHost) != hipSuccess || !h.probe_ok) {
            (void)hipGetLastError();
            if (getenv("DG_DEBUG")) fprintf(stderr, "[dipgenie_hip] L2 prefetcher off: its stream does not run beside the sweep's\n");
            S.l2_prefetch = 0;
            return DG_OK;
        }
    }
    const int seq = ++S.pf_seq;
    hipLaunchKernelGGL(dp_pf_ctl_kernel, dim3(1), dim3(1), 0, s, S.d_pfctl.as<PfCtl>(), seq, lb - 1, 0);
    DG_HIP(hipEventRecord(S.pf_ev, s));
    DG_HIP(hipStreamWaitEvent(S.pf_stream, S.pf_ev, 0));
    const int far = (int)S.pf_far;
    hipLaunchKernelGGL(dp_l2_prefetch_kernel, dim3(PF_WORKGROUPS + (far > 0 ? PF_FAR_WORKGROUPS : 0)), dim3(64), 0, S.pf_stream, S.d_descs.as<LevelDesc>(), X.F,
                       S.d_pfctl.as<PfCtl>(), seq, lb, le, (int)S.l2_prefetch, delta_resident ? 1 : 0, far);
    S.pf_active = true;
    return DG_OK;
}
void sweep_prefetch_end(DpState &S, int le, hipStream_t s) {
    if (S.pf_stream && S.l2_prefetch > 0) hipLaunchKernelGGL(dp_pf_ctl_kernel, dim3(1), dim3(1), 0, s, S.d_pfctl.as<PfCtl>(), S.pf_seq, le, 1);
}
void sweep_prefetch_free(DpState &S) {
    if (S.pf_stream) { (void)hipStreamSynchronize(S.pf_stream); (void)hipStreamDestroy(S.pf_stream); S.pf_stream = nullptr; }
    if (S.pf_ev) { (void)hipEventDestroy(S.pf_ev); S.pf_ev = nullptr; }
}

}  // namespace dgi
