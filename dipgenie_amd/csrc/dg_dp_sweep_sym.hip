// MEASUREMENT BUILD ONLY (make -C dipgenie_amd/csrc sym -> bin/libdipgenie_hip_sym.so, -DDG_SYM): the symmetric form of the level sweep.
// Built in round 4, bit-exact on every level digest (tests/test_gpu_parity.py::test_dp_symmetric_form, MHC-24, the
// chr22-style panel), and measured SLOWER than the plain form on the wide levels it was meant for (profiles/r04_sym_*.txt, DESIGN.md
// s3.3): a wide level's time is launch floor + stores (both triangles must be written) + gathers, and only the last term halves.
// The product library does not contain it.
#include <algorithm>

#include "dg_dp_sweep_dev.hpp"

namespace dgi {

// ---------------------------------------------------------------------------------------------
// Symmetric form, for the wide (throughput-bound) levels.
// The score of an in-edge pair is symmetric in (e_u, e_v) (approximator.cpp:604-624: both set sizes are symmetric in the two
// paths) and so is r2 = r + wu + wv, hence by induction over the levels value[r][i][j] = value[r][j][i]: cell (j2, i2) reduces over
// exactly the transposed candidates of cell (i2, j2).  Only the take-if tie-break (:657-659: pred_i asc, then pred_j asc) is not
// symmetric: the mirror cell prefers the smallest e_v rank first, then the smallest e_u rank.  So a task computes the cells
// (i2, j2 >= i2) only -- half the gathers, delta loads and selects -- and keeps BOTH orders beside the one running value: the packed
// word ord1 | ord2 << 16 with ord1 = (255 - eu) << 8 | (255 - ev) (this cell) and ord2 = (255 - ev) << 8 | (255 - eu) (its mirror);
// "equal value: larger order word wins" is one v_pk_max_u16 for the two halves.  Both cells get their value and their own
// back-pointer: lattice, traceback, digests and the state layout are those of the plain form.
// Workgroup = SYM_ROWS consecutive destination rows (one wave each) x one slot block x RC recombination counts, so that the
// mirror cells (j2, r2, i2) of a column j2 are SYM_ROWS consecutive words: they go through an LDS tile and leave as 64-byte
// (values) / 32-byte (back-pointers) segments instead of one cache line per lane.  A tile whose rows all lie beyond the block's last
// column has nothing to do and leaves after its first load; rows with more than COOP_MIN in-edges are taken out of the tiles and
// get a workgroup of their own per (block, chunk): its SYM_ROWS waves walk a sixteenth of the row's in-edges each and merge
// through LDS (their mirror is one column: strided stores, a few rows per level).  Columns without in-edges belong to nobody:
// the tiles of block 0 clear row AND column of every dead vertex.
// ---------------------------------------------------------------------------------------------
constexpr int SYM_ROWS = 16;
typedef unsigned short dg_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(dg_u16x2, a), __builtin_bit_cast(dg_u16x2, b)));
}
// packed order word = lane part (rank of the lane's in-edge e_v inside its column's list) | row part (rank t of the row's in-edge e_u)
__device__ __forceinline__ uint32_t symord_lane(int evr) { const uint32_t c = (uint32_t)(BP_MAX_RANK - evr); return c | (c << 24); }
__device__ __forceinline__ uint32_t symord_row(int t) { const uint32_t c = (uint32_t)(BP_MAX_RANK - t); return (c << 8) | (c << 16); }

template <int RC>
__device__ __forceinline__ void relax_select_sym(const int (&vals)[RC], int dl, uint32_t ordp, int r0, int w, int RP, int (&bval)[RC], uint32_t (&bord)[RC]) {
#pragma unroll
    for (int q = 0; q < RC; ++q) {
        const int cand = vals[q] + dl;
        const bool ok = (r0 + q < RP) & (r0 + q - w >= 0) & (vals[q] != NEG_INF);                   // :633, :646-647
        const bool gt = ok & (cand > bval[q]), eq = ok & (cand == bval[q]);                         // :657-659, both orders at once
        const uint32_t m = pk_max_u16(ordp, bord[q]);
        bord[q] = gt ? ordp : (eq ? m : bord[q]);
        bval[q] = gt ? cand : bval[q];
    }
}
__device__ __forceinline__ void merge_best_sym(int ov, uint32_t oo, bool same, int &bv, uint32_t &bo) {
    const bool gt = same & (ov > bv), eq = same & (ov == bv);
    const uint32_t m = pk_max_u16(oo, bo);
    bo = gt ? oo : (eq ? m : bo);
    bv = gt ? ov : bv;
}

template <int RC>
struct SymShared {
    union {
        struct { int32_t v[RC][64][SYM_ROWS + 1]; uint16_t b[RC][64][SYM_ROWS + 1]; } tile;   // mirror cells [count][column of the block][row of the tile]
        uint2 ex[SYM_ROWS - 1][RC][64];                                                      // fan-in rows: partial bests of waves 1 .. 15
    };
    int32_t col_j2[64];
    int32_t row_on[SYM_ROWS];
    int32_t n_cols;
};

// gathers of one destination row over the in-edges [t_lo, t_hi) of the row and the lanes' in-edges of the block (lanes of columns
// below the diagonal stay idle); the row's in-edge words: rr.z / rr.w (in-degree <= 2), else lane t of mypu (row matrix) or in_edge[]
template <int RC, bool GENERAL>
__device__ __forceinline__ void sym_gather(const LevelHead &H, const FastArgs &A, __amdgpu_buffer_rsrc_t cur_rsrc, const uint4 rr, uint2 sl, uint32_t mypu, int g, int nblk,
                                           int i2, int r0, int t_lo, int t_hi, int (&bval)[RC], uint32_t (&bord)[RC]) {
    const int lane = threadIdx.x & 63;
    const int RP = H.RP, dT = H.dT;
    const bool has_delta = dT != 0;
    const uint16_t *dm = H.dm;
    const int du = (int)rr.y;
    const bool rowx = !GENERAL && H.rowx_stride > 0;
    const int64_t erow0 = (int64_t)rr.x * dT;                           // (dm is biased by the level's first in-edge)
    const int rowbytes = H.k * 4;
    for (int blk = 0; blk < (GENERAL ? nblk : 1); ++blk) {
        if (GENERAL && blk > 0) sl = H.slots_l[(g + blk) * 64 + lane];
        const bool actb = (sl.x != 0xFFFFFFFFu) & ((int)((sl.x >> 16) & 0x7FFFu) >= i2);
        const int j = (int)(sl.x & 0x7FFFu), wv = (int)((sl.x >> 15) & 1u);
        const int dcol = has_delta ? (int)(sl.y & 0x000FFFFFu) : 0;
        const uint32_t ol = symord_lane((int)((sl.y >> 20) & 0xFFu));
        if (du <= 2 && t_lo == 0 && t_hi == du) {
            if (actb && du > 0) {
                const int ia = (int)(rr.z & 0x7FFFu), wa = (int)(rr.z >> 31) + wv;
                const int ib = (int)(rr.w & 0x7FFFu), wb = (int)(rr.w >> 31) + wv;
                const int offa = ((ia * RP + (r0 - wa)) * H.k + j) * 4 + H.pad_bytes;
                const int offb = ((ib * RP + (r0 - wb)) * H.k + j) * 4 + H.pad_bytes;
                int va[RC], vb[RC];
                const int dla = (int)dm[erow0 + dcol];
                int dlb = 0;
#pragma unroll
                for (int q = 0; q < RC; ++q) va[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offa + q * rowbytes, 0, 0);
                if (du == 2) {
                    dlb = (int)dm[erow0 + dT + dcol];
#pragma unroll
                    for (int q = 0; q < RC; ++q) vb[q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, offb + q * rowbytes, 0, 0);
                }
                relax_select_sym<RC>(va, dla, ol | symord_row(0), r0, wa, RP, bval, bord);
                if (du == 2) relax_select_sym<RC>(vb, dlb, ol | symord_row(1), r0, wb, RP, bval, bord);
            }
        } else {
            constexpr int U = RC >= 6 ? 1 : (RC >= 3 ? 2 : 4);           // (in-edges per step: U x RC values in flight)
            for (int c0 = GENERAL ? (t_lo & ~63) : 0; c0 < t_hi; c0 += GENERAL ? 64 : (1 << 30)) {
                const int tb = GENERAL ? max(c0, t_lo) : t_lo, te = GENERAL ? min(c0 + 64, t_hi) : t_hi;
                if (!rowx) { mypu = 0; if (c0 + lane < te) mypu = A.in_edge[rr.x + c0 + lane]; }
                for (int t = tb; t < te; t += U) {
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
                                for (int q = 0; q < RC; ++q) vals[u][q] = __builtin_amdgcn_raw_buffer_load_b32(cur_rsrc, off + q * rowbytes, 0, 0);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (t + u < te) {
                                const int wu = (int)((wmask >> u) & 1u);
                                relax_select_sym<RC>(vals[u], dl[u], ol | symord_row(t + u), r0, wu + wv, RP, bval, bord);
                            }
                        }
                    }
                }
            }
        }
    }
}

// segmented max over the lanes of one destination column (adjacent lanes), both orders
template <int RC>
__device__ __forceinline__ void sym_column_max(int steps, int j2, int (&bval)[RC], uint32_t (&bord)[RC]) {
    const int lane = threadIdx.x & 63;
    if (steps > 0) {
        const int oj2 = lane_down1(j2);
        const bool same = (lane + 1 < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) merge_best_sym(lane_down1(bval[q]), (uint32_t)lane_down1((int)bord[q]), same, bval[q], bord[q]);
    }
    for (int st = 1, sh = 2; st < steps; ++st, sh <<= 1) {
        const int oj2 = __shfl_down(j2, sh);
        const bool same = (lane + sh < 64) & (oj2 == j2);
#pragma unroll
        for (int q = 0; q < RC; ++q) merge_best_sym(__shfl_down(bval[q], sh), (uint32_t)__shfl_down((int)bord[q], sh), same, bval[q], bord[q]);
    }
}

// digest terms of cell (i2, j2) and, off the diagonal, of its mirror (parity runs only)
__device__ __forceinline__ unsigned long long sym_digest(const LevelHead &H, const FastArgs &A, const uint4 rr, int k2, int i2, int j2, int r2, int value, uint32_t bordp) {
    const uint32_t o1 = bordp & 0xFFFFu, o2 = bordp >> 16;
    const uint32_t col0 = H.rowrec_l[j2].x;
    const uint32_t pi = A.in_edge[rr.x + (BP_MAX_RANK - (o1 >> 8))] & 0x7FFFFFFFu, pj = A.in_edge[col0 + (BP_MAX_RANK - (o1 & 0xFFu))] & 0x7FFFFFFFu;
    unsigned long long sum = digest_term(value, ((unsigned long long)r2 * k2 + i2) * k2 + j2, pi, pj);
    if (j2 > i2) {
        const uint32_t mi = A.in_edge[col0 + (BP_MAX_RANK - (o2 >> 8))] & 0x7FFFFFFFu, mj = A.in_edge[rr.x + (BP_MAX_RANK - (o2 & 0xFFu))] & 0x7FFFFFFFu;
        sum += digest_term(value, ((unsigned long long)r2 * k2 + j2) * k2 + i2, mi, mj);
    }
    return sum;
}

// one (tile of SYM_ROWS rows, slot block, chunk): every wave one row
template <int RC, bool DIGEST, bool GENERAL>
__device__ __forceinline__ void sym_tile(const LevelHead &H, const FastArgs &A, const LevelDesc &d, SymShared<RC> &sh, __amdgpu_buffer_rsrc_t cur_rsrc, int32_t *__restrict__ nxt,
                                         int tile, int g, int r0, int lvl, int n_heavy, bool clear_dead, int dbg) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int RP = H.RP, k2 = d.k2;
    // vertices without in-edges: unreachable as row and as column; the tile's first workgroup clears both for its rows
    if (clear_dead && d.ndead > 0) {
        const int row0 = tile * SYM_ROWS, nrow = min(SYM_ROWS, k2 - row0);
        for (int t = (int)threadIdx.x; t < nrow * d.ndead * RC; t += SYM_ROWS * 64) {
            const int q = t % RC, rest = t / RC;
            const int c = A.dead_cols[d.dead_first + rest % d.ndead], i2 = row0 + rest / d.ndead;
            if (r0 + q < RP) {
                const int ia = (i2 * RP + r0 + q) * k2 + c, ib = (c * RP + r0 + q) * k2 + i2;
                nxt[ia] = NEG_INF; nxt[ib] = NEG_INF;
                if (A.bp) { A.bp[d.bp_off + ia] = (uint16_t)0xFFFFu; A.bp[d.bp_off + ib] = (uint16_t)0xFFFFu; }
            }
        }
    }
    if (g < 0) return;                                                  // (a tile whose rows are all dead vertices behind the last block: clearing only)
    // first load round: the block's slot records (the same in every wave of the workgroup) and the wave's row record
    const uint2 sl = H.slots_l[g * 64 + lane];
    const int i2 = tile * SYM_ROWS + wave;
    const uint4 rr = H.rowrec_l[min(i2, k2 - 1)];
    if (dbg & 32) { asm volatile("" ::"v"(sl.x), "v"(rr.x)); return; }
    int steps = __builtin_amdgcn_readfirstlane((int)(sl.y >> 28));
    const bool act0 = sl.x != 0xFFFFFFFFu;
    const int j2 = act0 ? (int)((sl.x >> 16) & 0x7FFFu) : -1 - lane;
    const unsigned long long am = __builtin_amdgcn_ballot_w64(act0);
    if (am == 0) return;                                                // (a level without in-edges: all dead, cleared above)
    int nblk = 1;
    if (GENERAL) {
        if (steps == 14) return;                                        // continuation block of a giant column: walked by the wave of its first block
        if (steps == 15) { nblk = ((int)H.rowrec_l[__builtin_amdgcn_readfirstlane(j2)].y + 63) >> 6; steps = 6; }
    }
    const int cmax = __builtin_amdgcn_readlane(j2, 63 - __builtin_clzll(am));   // last column of the block (columns ascend with the lanes)
    if (cmax < tile * SYM_ROWS) return;                                 // every row of the tile lies beyond the block's last column (workgroup-uniform)
    const int pj2 = lane_up1(j2);
    const bool head0 = act0 & ((lane == 0) | (pj2 != j2));
    int bval[RC];
    uint32_t bord[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q) { bval[q] = NEG_INF; bord[q] = 0; }
    unsigned long long dsum = 0;
    const bool row_on = (i2 < k2) & (i2 <= cmax) & !((n_heavy > 0) & ((int)rr.y > COOP_MIN));
    if (row_on) {
        uint32_t mypu = 0;
        if (!GENERAL && H.rowx_stride > 0 && lane < H.rowx_stride) mypu = H.rowx_l[i2 * H.rowx_stride + lane];
        if (!(dbg & 8)) sym_gather<RC, GENERAL>(H, A, cur_rsrc, rr, sl, mypu, g, nblk, i2, r0, 0, (int)rr.y, bval, bord);
        sym_column_max<RC>(steps, j2, bval, bord);
    }
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(head0);
    const int cidx = __builtin_popcountll(hm & ((1ULL << lane) - 1ULL));     // this lane's column inside the block
    if (wave == 0) { if (head0) sh.col_j2[cidx] = j2; if (lane == 0) sh.n_cols = __builtin_popcountll(hm); }
    if (lane == 0) sh.row_on[wave] = row_on ? 1 : 0;
    if (row_on && head0 && j2 >= i2) {
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int r2 = r0 + q;
            if (r2 < RP) {
                const int ia = (i2 * RP + r2) * k2 + j2;
                nxt[ia] = bval[q];
                if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + ia], ~bord[q]); else A.bp[d.bp_off + ia] = (uint16_t)~bord[q]; }
                sh.tile.v[q][cidx][wave] = bval[q];
                sh.tile.b[q][cidx][wave] = (uint16_t)(~bord[q] >> 16);
                if (DIGEST && bval[q] != NEG_INF) dsum += sym_digest(H, A, rr, k2, i2, j2, r2, bval[q], bord[q]);
            }
        }
    }
    if (DIGEST && dsum) atomicAdd(&A.digest[lvl], dsum);
    if (dbg & 4) return;
    __syncthreads();
    // mirror cells (j2, r2, i2): lanes <-> (column of the block, row of the tile), rows fastest: SYM_ROWS consecutive words per column
    const int tt = lane & (SYM_ROWS - 1), c = wave * (64 / SYM_ROWS) + (lane / SYM_ROWS);
    if (c < sh.n_cols && sh.row_on[tt]) {
        const int jj = sh.col_j2[c], ii = tile * SYM_ROWS + tt;
        if (jj > ii) {
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const int r2 = r0 + q;
                if (r2 < RP) {
                    const int ib = (jj * RP + r2) * k2 + ii;
                    nxt[ib] = sh.tile.v[q][c][tt];
                    if (A.bp) { if (d.bp_nt) store_bp_nt(&A.bp[d.bp_off + ib], (uint32_t)sh.tile.b[q][c][tt]); else A.bp[d.bp_off + ib] = sh.tile.b[q][c][tt]; }
                }
            }
        }
    }
}

// fold = 1 (lean levels with at most 64 blocks): the (tile, block) pairs above the diagonal are dealt to the grid without holes.  Tile t needs
// the blocks g >= gmin(t), gmin(t) = number of blocks that end in front of column 16 t's first in-edge (one vector load of the level's block
// boundaries, two ballots); workgroup (x, z) takes pair x of tile z, then of tile NT - 1 - z: together about NB + 1 pairs whatever z.
template <int RC, bool DIGEST, bool GENERAL>
__global__ __launch_bounds__(SYM_ROWS * 64) void dp_sweep_sym_kernel(const uint4 *rowrec_l, const uint2 *slots_l, const uint32_t *rowx_l, const int32_t *cur, const uint16_t *dm,
                                                                      int rowx_stride, int nblocks, int rp_k, int pad_bytes, int dT, uint32_t buf_bytes,   // 16 dwords: preloaded
                                                                      FastArgs A, LevelDesc d, int lvl, int n_heavy, const int32_t *__restrict__ heavy_rows, int dbg,
                                                                      int fold, const uint32_t *__restrict__ gb_l) {
    __shared__ SymShared<RC> sh;
    if (dbg & 16) return;
    const LevelHead H{rowrec_l, slots_l, rowx_l, cur, dm, rowx_stride, rp_k & 0x1FFF, rp_k >> 13, pad_bytes, dT, buf_bytes};
    publish_level(A.progress, lvl);
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int r0 = (int)blockIdx.y * RC, RP = H.RP, k2 = d.k2;
    int32_t *__restrict__ nxt = (int32_t *)(A.ring + (size_t)DG_SLOT(A, lvl) * A.slot_bytes + A.pad_bytes);
    const __amdgpu_buffer_rsrc_t cur_rsrc = state_rsrc(cur, buf_bytes);
    if ((int)blockIdx.z >= n_heavy) {
        const int z = (int)blockIdx.z - n_heavy;
        if (GENERAL || !fold) { sym_tile<RC, DIGEST, GENERAL>(H, A, d, sh, cur_rsrc, nxt, z, (int)blockIdx.x, r0, lvl, n_heavy, blockIdx.x == 0, dbg); return; }
        const int NT = (k2 + SYM_ROWS - 1) / SYM_ROWS, t1 = z, t2 = NT - 1 - z;
        const uint32_t e1 = rowrec_l[t1 * SYM_ROWS].x, e2 = rowrec_l[min(t2 * SYM_ROWS, k2 - 1)].x;
        const uint32_t gend = lane < nblocks ? gb_l[lane + 1] : 0xFFFFFFFFu;       // first in-edge behind block `lane`
        const int gmin1 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(gend <= e1)), gmin2 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(gend <= e2));
        const int need1 = nblocks - gmin1, need2 = t2 > t1 ? nblocks - gmin2 : 0;
        const int n1 = max(need1, 1), n2 = t2 > t1 ? max(need2, 1) : 0;            // (a tile without a block of its own still has its dead vertices cleared)
        bool first = true;
        for (int x = (int)blockIdx.x; x < n1 + n2; x += (int)gridDim.x) {            // (one trip, unless a pair of tiles needs more blocks than the grid is wide)
            if (!first) __syncthreads();                                            // the tile in LDS is free again
            first = false;
            if (x < n1) sym_tile<RC, DIGEST, GENERAL>(H, A, d, sh, cur_rsrc, nxt, t1, x < need1 ? gmin1 + x : -1, r0, lvl, n_heavy, x == 0, dbg);
            else sym_tile<RC, DIGEST, GENERAL>(H, A, d, sh, cur_rsrc, nxt, t2, x - n1 < need2 ? gmin2 + (x - n1) : -1, r0, lvl, n_heavy, x == n1, dbg);
        }
        return;
    }
    // ---- one fan-in row, its in-edges dealt to the SYM_ROWS waves ----
    if (dbg & 1) return;
    const int g = (int)blockIdx.x;
    if (g >= nblocks) return;
    const uint2 sl = slots_l[g * 64 + lane];
    const int h = (int)blockIdx.z;
    const int i2 = h < HEAVY_INLINE ? (int)d.heavy_in[h] : heavy_rows[d.heavy_first + h];
    const uint4 rr = rowrec_l[i2];
    int steps = __builtin_amdgcn_readfirstlane((int)(sl.y >> 28));
    const bool act0 = sl.x != 0xFFFFFFFFu;
    const int j2 = act0 ? (int)((sl.x >> 16) & 0x7FFFu) : -1 - lane;
    const unsigned long long am = __builtin_amdgcn_ballot_w64(act0);
    if (am == 0) return;
    int nblk = 1;
    if (GENERAL) {
        if (steps == 14) return;
        if (steps == 15) { nblk = ((int)rowrec_l[__builtin_amdgcn_readfirstlane(j2)].y + 63) >> 6; steps = 6; }
    }
    const int cmax = __builtin_amdgcn_readlane(j2, 63 - __builtin_clzll(am));
    if (cmax < i2) return;                                              // the whole block lies below the diagonal (workgroup-uniform)
    const int pj2 = lane_up1(j2);
    const bool head0 = act0 & ((lane == 0) | (pj2 != j2));
    int bval[RC];
    uint32_t bord[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q) { bval[q] = NEG_INF; bord[q] = 0; }
    unsigned long long dsum = 0;
    const int du = (int)rr.y;
    uint32_t mypu = 0;
    if (!GENERAL && rowx_stride > 0 && lane < rowx_stride) mypu = rowx_l[i2 * rowx_stride + lane];
    sym_gather<RC, GENERAL>(H, A, cur_rsrc, rr, sl, mypu, g, nblk, i2, r0, (du * wave) / SYM_ROWS, (du * (wave + 1)) / SYM_ROWS, bval, bord);
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < RC; ++q) sh.ex[wave - 1][q][lane] = make_uint2((uint32_t)bval[q], bord[q]);
    }
    __syncthreads();
    if (wave > 0) return;
    for (int p = 0; p < SYM_ROWS - 1; ++p) {
#pragma unroll
        for (int q = 0; q < RC; ++q) { const uint2 o = sh.ex[p][q][lane]; merge_best_sym((int)o.x, o.y, true, bval[q], bord[q]); }
    }
    sym_column_max<RC>(steps, j2, bval, bord);
    if (head0 && j2 >= i2) {
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int r2 = r0 + q;
            if (r2 < RP) {
                const int ia = (i2 * RP + r2) * k2 + j2, ib = (j2 * RP + r2) * k2 + i2;
                nxt[ia] = bval[q];
                if (A.bp) A.bp[d.bp_off + ia] = (uint16_t)~bord[q];
                if (j2 > i2) { nxt[ib] = bval[q]; if (A.bp) A.bp[d.bp_off + ib] = (uint16_t)(~bord[q] >> 16); }
                if (DIGEST && bval[q] != NEG_INF) dsum += sym_digest(H, A, rr, k2, i2, j2, r2, bval[q], bord[q]);
            }
        }
    }
    if (DIGEST && dsum) atomicAdd(&A.digest[lvl], dsum);
}

// host side: launches the symmetric form for level l where it applies; false = the caller launches the plain form
bool sweep_launch_sym(DpState &S, SweepLaunch &X, int l, hipStream_t s) {
    LevelDesc &d = S.descs[l];
    // wide levels: the symmetric form (upper triangle computed, both triangles stored); sym 2 = wherever the form exists (tests)
    if (d.fast_ok && X.small_state && S.RP <= 8191 && S.use_fast && S.use_sym && (S.use_sym >= 2 || (d.k2 >= S.sym_min_k2 && d.nblocks >= 2)) &&
        (d.k2 + SYM_ROWS - 1) / SYM_ROWS + d.n_heavy <= 65535 && d.nblocks <= 0x7FFFFFFF / 64) {
        int rc = (int)std::min<int64_t>(std::max<int64_t>(S.sym_rc, 1), 8);
        rc = rc == 5 ? 4 : (rc == 7 ? 6 : rc);
        rc = std::min(rc, S.RP <= 8 ? std::max(1, S.RP) : 8);
        if (rc == 5) rc = 4; if (rc == 7) rc = 6;
        const int nh = S.use_coop ? d.n_heavy : 0;
        S.launch_hist[(40 + rc) * 4 + (d.fast_ok == 2 ? 2 : 0)]++;
        const int NT = (d.k2 + SYM_ROWS - 1) / SYM_ROWS;
        const int fold = (S.sym_fold && d.fast_ok == 1 && d.nblocks <= 64 && d.ngroups == d.nblocks) ? 1 : 0;   // lean levels: groups are the blocks
        // (fold: x = pair of a tile and its mirror tile, z = the lower half of the tiles; a fan-in row needs the first nblocks of the x range)
        const dim3 grid((unsigned)(fold ? d.nblocks + 2 : d.nblocks), (unsigned)((S.RP + rc - 1) / rc), (unsigned)((fold ? (NT + 1) / 2 : NT) + nh));
        const uint32_t *gb_l = X.A.grp_begin + d.grp_first;
        const FastArgs &F = X.F;
        const uint4 *rowrec_l = F.rowrec + d.b0;
        const uint2 *slots_l = F.slots + d.slot_first;
        const uint32_t *rowx_l = F.rowx + d.rowx_off;
        const int32_t *cur = (const int32_t *)(F.ring + (size_t)DG_SLOT(F, l - 1) * F.slot_bytes);
        const int dT = d.delta_off >= 0 ? d.T : 0;
        const uint16_t *dm = dT ? F.delta + d.delta_off - (int64_t)d.in_base * dT : F.delta_zero;
        const int rp_k = S.RP | (d.k << 13);
        const int32_t *hv = S.d_heavy.as<int32_t>();
#define DG_SYM(RCV, DG) do { if (d.fast_ok == 2) hipLaunchKernelGGL((dp_sweep_sym_kernel<RCV, DG, true>), grid, dim3(SYM_ROWS * 64), 0, s, rowrec_l, slots_l, rowx_l, cur, dm, d.rowx_stride, d.nblocks, rp_k, F.pad_bytes, dT, F.buf_bytes, F, d, l, nh, hv, (int)S.sym_dbg, fold, gb_l); \
                             else hipLaunchKernelGGL((dp_sweep_sym_kernel<RCV, DG, false>), grid, dim3(SYM_ROWS * 64), 0, s, rowrec_l, slots_l, rowx_l, cur, dm, d.rowx_stride, d.nblocks, rp_k, F.pad_bytes, dT, F.buf_bytes, F, d, l, nh, hv, (int)S.sym_dbg, fold, gb_l); } while (0)
#define DG_SYM_RC(DG) do { switch (rc) { case 1: DG_SYM(1, DG); break; case 2: DG_SYM(2, DG); break; case 3: DG_SYM(3, DG); break; case 4: DG_SYM(4, DG); break; \
                                         case 6: DG_SYM(6, DG); break; default: DG_SYM(8, DG); break; } } while (0)
        if (S.want_digest) DG_SYM_RC(true); else DG_SYM_RC(false);
#undef DG_SYM_RC
#undef DG_SYM
        return true;
    }
    return false;
}

}  // namespace dgi
