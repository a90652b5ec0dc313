// Traceback of the diploid DP (approximator.cpp:757-785).
// Chain kernel (one wave): walk the back-pointer lattice downwards from a known cell.  The chain is one dependent HBM
// load per level, so everything else is kept off it: the level descriptors of the next 64 levels are fetched one per
// lane and broadcast with readlane, and the hop words are parked in path[].  In segmented mode it is called once per
// chunk, last chunk first, carrying the cell in ChainState.
// Finish kernel (whole grid, levels in parallel): re-derive s_het from the colour lists of the winning edge pairs
// (:662) and emit the weighted edges (:673-692; both final edges unconditionally) as (level, from, to, which) records;
// the host orders them by level.
#include <algorithm>

#include "dg_dp_setops.hpp"

namespace dgi {

// Pulls the row records of a range of levels into the memory-side Infinity Cache right before the chain walk reads two
// of them per level (they were last touched by the sweep, hundreds of GB of lattice writes ago).
__global__ __launch_bounds__(256) void dp_warm_kernel(const uint4 *__restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint4 v = p[i]; asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
}

// Speculative chain walk.  A plain step (load the back-pointer, decode, move) costs one HBM round trip because the next
// cell's address needs this level's back-pointer.  But the candidates are few: with at most two in-edges into the row
// and into the column (97 % of the vertices) the predecessor is one of four cells, all known as soon as the two row
// records are in.  Lanes 0..3 (mirrored by the other lanes) therefore load the back-pointer AND the row records of
// "their" candidate one level ahead; when this level's back-pointer arrives it only selects the lane whose loads are
// already in flight, so two levels' HBM round trips overlap.  A rank above 1 or a wide level resolves the predecessor
// with extra loads and re-issues exact loads for it.
// Shape of the code (it matters: the compiler derives its s_waitcnt from it): every path through a step ends with the
// three loads (row record, column record, back-pointer word) of the next level as the most recent memory operations, and
// a step consumes only what the previous step issued; the step is instantiated twice with the two register sets swapped,
// because copying a register that still has a load in flight would wait for that load.
struct ChainWalk {
    int i, j, r, value, csel;
};
struct ChainRegs { uint4 row, col; uint32_t b; };
struct ChainDesc { long long bo; int k2, bw, kp; };    // bp offset, width, first vertex | wide << 31, source width

// loads of one cell (level descriptor bo/k2/bw): row records + the aligned 32-bit word that holds its back-pointer
__device__ __forceinline__ void chain_issue(ChainRegs &Y, const uint16_t *__restrict__ bp, const uint4 *__restrict__ rowrec, long long bo, int k2, int bw,
                                            int RP, int ci, int cj, int cr) {
    const int b0 = bw & 0x7FFFFFFF;
    const long long unit = bo + (((long long)ci * RP + cr) * k2 + cj) * (bw < 0 ? 2 : 1);
    Y.row = rowrec[b0 + ci];
    Y.col = rowrec[b0 + cj];
    Y.b = *(const uint32_t *)(bp + (unit & ~1LL));
}

__device__ __forceinline__ void chain_spec_step(ChainWalk &W, int l, int l_lo, int RP, const ChainDesc &D, const ChainDesc &N,
                                                const uint16_t *__restrict__ bp, const uint4 *__restrict__ rowrec, const uint32_t *__restrict__ in_edge,
                                                uint32_t &hops, int slot, int lane, ChainRegs &X, ChainRegs &Y) {
    const int ca = lane & 1, cb = (lane >> 1) & 1;                      // this lane's candidate: rank ca of the row, cb of the column
    const long long bo_l = D.bo, nbo = N.bo;
    const int k2 = D.k2, bw = D.bw, kprev = D.kp, nk2 = N.k2, nbw = N.bw;
    const int cs = W.csel;
    const uint32_t rix = (uint32_t)__builtin_amdgcn_readlane((int)X.row.x, cs), riy = (uint32_t)__builtin_amdgcn_readlane((int)X.row.y, cs);
    const uint32_t riz = (uint32_t)__builtin_amdgcn_readlane((int)X.row.z, cs), riw = (uint32_t)__builtin_amdgcn_readlane((int)X.row.w, cs);
    const uint32_t rjx = (uint32_t)__builtin_amdgcn_readlane((int)X.col.x, cs), rjy = (uint32_t)__builtin_amdgcn_readlane((int)X.col.y, cs);
    const uint32_t rjz = (uint32_t)__builtin_amdgcn_readlane((int)X.col.z, cs), rjw = (uint32_t)__builtin_amdgcn_readlane((int)X.col.w, cs);
    // one level ahead: the four candidate cells of level l - 1; a lane without a valid candidate re-reads this level's cell
    const bool have_next = l - 1 >= l_lo;
    const uint32_t wa = ca ? riw : riz, wb = cb ? rjw : rjz;
    const int ci = (int)(wa & 0x7FFFu), cj = (int)(wb & 0x7FFFu), cr = W.r - (int)(wa >> 31) - (int)(wb >> 31);
    const bool ok = have_next & ((uint32_t)ca < riy) & ((uint32_t)cb < rjy) & (cr >= 0) & (ci < kprev) & (cj < kprev);
    chain_issue(Y, bp, rowrec, ok ? nbo : bo_l, ok ? nk2 : k2, ok ? nbw : bw, RP, ok ? ci : W.i, ok ? cj : W.j, ok ? cr : W.r);
    // this level's back-pointer word (issued one step ago)
    const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)X.b, cs);
    uint32_t hop;
    bool exact_next = false;                                            // the predecessor is not among the candidates in flight
    if (bw < 0) {                                                       // wide level: the word is the hop
        hop = word;
        exact_next = true;
    } else {
        const long long unit = bo_l + ((long long)W.i * RP + W.r) * k2 + W.j;
        const uint32_t bv = (unit & 1) ? (word >> 16) : (word & 0xFFFFu);
        const uint32_t eu = bv >> 8, ev = bv & 0xFFu;
        if (eu >= riy || ev >= rjy) { W.value = CHAIN_CORRUPT; W.i = W.j = 0; W.r = 0; return; }   // (0xFFFF = unreachable lands here too)
        const uint32_t pu = eu == 0 ? riz : (eu == 1 ? riw : in_edge[rix + eu]);
        const uint32_t pv = ev == 0 ? rjz : (ev == 1 ? rjw : in_edge[rjx + ev]);
        hop = (pu & 0x7FFFu) | ((pv & 0x7FFFu) << 15) | ((pu >> 31) << 30) | ((pv >> 31) << 31);
        exact_next = (eu > 1) | (ev > 1);
        W.csel = (int)((eu & 1u) | ((ev & 1u) << 1));
    }
    hops = lane == slot ? hop : hops;                                   // lane t keeps the hop of the batch's t-th level: one store per batch,
    W.i = (int)(hop & 0x7FFFu); W.j = (int)((hop >> 15) & 0x7FFFu);     // none inside the walk (stores share the loads' counter)
    W.r -= (int)((hop >> 30) & 1u) + (int)(hop >> 31);
    // a hop that leaves the source level means the lattice is corrupt (a level nobody swept): stop before the next
    // load goes wild -- the host reports DG_ERR_STATE instead of the GPU faulting
    if ((W.i >= kprev) | (W.j >= kprev) | (W.r < 0)) { W.value = CHAIN_CORRUPT; W.i = W.j = 0; W.r = 0; return; }
    if (have_next && (exact_next || nbw < 0)) {                         // rare: exact loads of the predecessor replace the candidates
        chain_issue(Y, bp, rowrec, nbo, nk2, nbw, RP, W.i, W.j, W.r);
        W.csel = 0;
    }
}

__global__ __launch_bounds__(64) void dp_trace_chain_spec_kernel(const LevelDesc *__restrict__ descs, int l_hi, int l_lo, int RP, int R,
                                                                 const uint16_t *__restrict__ bp /* biased by the segment's first unit */,
                                                                 const int32_t *__restrict__ final_val /* non-null on the first call */,
                                                                 const uint4 *__restrict__ rowrec, const uint32_t *__restrict__ in_edge,
                                                                 uint2 *__restrict__ path, ChainState *st) {
    const int lane = threadIdx.x & 63;
    ChainWalk W;
    W.csel = 0;
    if (final_val) { W.value = final_val[(int64_t)R * descs[l_hi].k2]; W.i = 0; W.j = 0; W.r = R; }   // sink level, layout [i][r][j]: cell (0, R, 0)
    else { W.i = st->i; W.j = st->j; W.r = st->r; W.value = st->value; }
    if (W.value != NEG_INF && W.value != CHAIN_CORRUPT) {
        ChainRegs A, B;
        bool first = true;
        for (int base = l_hi; base >= l_lo && W.value != CHAIN_CORRUPT; base -= 56) {
            const int my_l = base - lane;                               // 64 descriptors, 56 levels (a multiple of 8) per batch
            long long bo = 0;
            int kk = 1, bb = 0, kp = 1;                                 // bb = first vertex of the level | wide << 31; kp = source width
            if (my_l >= l_lo) { bo = descs[my_l].bp_off; kk = descs[my_l].k2; kp = descs[my_l].k; bb = descs[my_l].b0 | (descs[my_l].bp_wide << 31); }
            asm volatile("" ::"v"(bo), "v"(kk), "v"(bb), "v"(kp));      // descriptors complete before the walk (no wait inside the loop)
            const int n = min(56, base - l_lo + 1);
            uint32_t hops = 0;
#define DG_DESC(T) ChainDesc{((long long)__builtin_amdgcn_readlane((int)(bo >> 32), (T)) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)bo, (T)), \
                            __builtin_amdgcn_readlane(kk, (T)), __builtin_amdgcn_readlane(bb, (T)), __builtin_amdgcn_readlane(kp, (T))}
            ChainDesc D0 = DG_DESC(0), D1;                              // descriptor of the level at hand / one level ahead, rotated like the registers
            if (first) {                                                // prologue: exact loads of the starting cell
                chain_issue(A, bp, rowrec, D0.bo, D0.k2, D0.bw, RP, W.i, W.j, W.r);
                first = false;
            }
            // unrolled by hand (8 steps per trip): the loop's back edge copies the registers of the loads in flight, and so
            // waits for them -- one un-overlapped step per trip
#define DG_WALK_STOP(T) (W.value == CHAIN_CORRUPT || (T) >= n)
#define DG_WALK_STEP(T, X, Y, DC, DN) DN = DG_DESC((T) + 1); chain_spec_step(W, base - (T), l_lo, RP, DC, DN, bp, rowrec, in_edge, hops, (T), lane, X, Y)
            for (int t = 0; t < n; t += 8) {
                DG_WALK_STEP(t, A, B, D0, D1);     if (DG_WALK_STOP(t + 1)) break;
                DG_WALK_STEP(t + 1, B, A, D1, D0); if (DG_WALK_STOP(t + 2)) break;
                DG_WALK_STEP(t + 2, A, B, D0, D1); if (DG_WALK_STOP(t + 3)) break;
                DG_WALK_STEP(t + 3, B, A, D1, D0); if (DG_WALK_STOP(t + 4)) break;
                DG_WALK_STEP(t + 4, A, B, D0, D1); if (DG_WALK_STOP(t + 5)) break;
                DG_WALK_STEP(t + 5, B, A, D1, D0); if (DG_WALK_STOP(t + 6)) break;
                DG_WALK_STEP(t + 6, A, B, D0, D1); if (DG_WALK_STOP(t + 7)) break;
                DG_WALK_STEP(t + 7, B, A, D1, D0); if (DG_WALK_STOP(t + 8)) break;
            }
#undef DG_WALK_STOP
#undef DG_WALK_STEP
#undef DG_DESC
            if (lane < n && W.value != CHAIN_CORRUPT)               // the path holds in-edge style words: source | weight << 31
                path[base - lane] = make_uint2((hops & 0x7FFFu) | (((hops >> 30) & 1u) << 31), ((hops >> 15) & 0x7FFFu) | ((hops >> 31) << 31));
        }
    }
    if (lane == 0) { st->i = W.i; st->j = W.j; st->r = W.r; st->value = W.value; }
}


// Lean chain walk: the same speculation, for lattices without wide levels whose per-level offsets fit 32 bits (every
// graph the fast sweep takes).  One wave issues one instruction every four cycles whatever its kind, and the step above
// costs ~250 of them (0.5 us) -- more than the two overlapped HBM round trips it hides.  This one is written around the
// instruction count (~1/3): 16-bit loads of the back-pointer itself (no word/parity selection), 24-bit multiply-adds on
// clamped candidate coordinates instead of selects between "candidate" and "fallback" address sets (a clamped address
// is always inside the level's block, whatever the lattice holds), scalar-base + 32-bit-offset addressing, the source
// words parked unpacked, and no per-step validity checks: a back-pointer rank beyond the in-degree reads the all-ones
// guard word, whose source id is out of range at every level, and the finish kernel (all levels in parallel) checks
// every hop's source ids against its level's width.
struct LeanRegs { uint4 row, col; uint32_t b; };

__device__ __forceinline__ void lean_issue(LeanRegs &Y, const uint16_t *__restrict__ bp_level /* wave-uniform: first unit of the level */, const char *__restrict__ rowrec, uint32_t b0, uint32_t k2,
                                           uint32_t rp_k2, uint32_t ci, uint32_t cj, uint32_t cr) {
    const uint32_t off = __umul24(ci, rp_k2) + __umul24(cr, k2) + cj;
    Y.row = *(const uint4 *)(rowrec + ((b0 + ci) << 4));
    Y.col = *(const uint4 *)(rowrec + ((b0 + cj) << 4));
    Y.b = *(const uint16_t *)((const char *)bp_level + (off << 1));
}

struct LeanDesc { const uint16_t *bp_level; uint32_t k2, b0; };
struct LeanWalk { int r, cs; uint32_t pu, pv; bool bad; };

__device__ __forceinline__ void lean_step(LeanWalk &W, int RP, const LeanDesc &N, const char *rowrec, const uint32_t *__restrict__ in_edge,
                                          uint32_t &park_u, uint32_t &park_v, int slot, int lane, LeanRegs &X, LeanRegs &Y) {
    const int cs = __builtin_amdgcn_readfirstlane(W.cs);
    // guard: rank 1 of a vertex with a single in-edge is not a predecessor
    const uint32_t xw = X.row.y > 1u ? X.row.w : 0xFFFFFFFFu, yw = X.col.y > 1u ? X.col.w : 0xFFFFFFFFu;
    const uint32_t riz = (uint32_t)__builtin_amdgcn_readlane((int)X.row.z, cs), riw = (uint32_t)__builtin_amdgcn_readlane((int)xw, cs);
    const uint32_t rjz = (uint32_t)__builtin_amdgcn_readlane((int)X.col.z, cs), rjw = (uint32_t)__builtin_amdgcn_readlane((int)yw, cs);
    // one level ahead: this lane's candidate (rank lane&1 of the row, (lane>>1)&1 of the column), clamped into the level
    const uint32_t wa = (lane & 1) ? riw : riz, wb = (lane & 2) ? rjw : rjz;
    const uint32_t ci = min(wa & 0x7FFFu, N.k2 - 1u), cj = min(wb & 0x7FFFu, N.k2 - 1u);
    const uint32_t cr = (uint32_t)max(W.r - (int)(wa >> 31) - (int)(wb >> 31), 0);
    lean_issue(Y, N.bp_level, rowrec, N.b0, N.k2, (uint32_t)RP * N.k2, ci, cj, cr);
    // the candidates' loads are issued HERE, before this level's back-pointer (issued one step ago) is waited for
    // (the broadcast is spelled out so that it -- and the wait in front of it -- cannot be scheduled above the loads; cs is written
    // by scalar instructions only, so the VALU-writes-lane-select hazard does not apply)
    uint32_t bv;
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(bv) : "v"(X.b), "s"(cs) : "memory");
    const uint32_t eu = bv >> 8, ev = bv & 0xFFu;
    uint32_t pu, pv;
    if ((eu | ev) > 1u) {                                               // rare: a rank above 1 (or 0xFFFF = unreachable)
        const uint32_t rix = (uint32_t)__builtin_amdgcn_readlane((int)X.row.x, cs), riy = (uint32_t)__builtin_amdgcn_readlane((int)X.row.y, cs);
        const uint32_t rjx = (uint32_t)__builtin_amdgcn_readlane((int)X.col.x, cs), rjy = (uint32_t)__builtin_amdgcn_readlane((int)X.col.y, cs);
        if (eu >= riy || ev >= rjy) { W.bad = true; pu = pv = 0; }
        else {
            pu = eu == 0 ? riz : (eu == 1 ? riw : in_edge[rix + eu]);
            pv = ev == 0 ? rjz : (ev == 1 ? rjw : in_edge[rjx + ev]);
            pu = (uint32_t)__builtin_amdgcn_readfirstlane((int)pu); pv = (uint32_t)__builtin_amdgcn_readfirstlane((int)pv);
        }
        W.r -= (int)(pu >> 31) + (int)(pv >> 31);
        // exact loads of the predecessor replace the candidates in flight.  (The candidates are "used" here so that the
        // compiler cannot treat their loads as dead on this path and sink the issue below the branch -- behind the wait.)
        asm volatile("" :: "v"(Y.b), "v"(Y.row.x), "v"(Y.col.x));
        uint32_t ei = min(pu & 0x7FFFu, N.k2 - 1u), ej = min(pv & 0x7FFFu, N.k2 - 1u);
        asm volatile("" : "+v"(ei), "+v"(ej));                          // vector loads like the candidates' (same registers, same extension)
        lean_issue(Y, N.bp_level, rowrec, N.b0, N.k2, (uint32_t)RP * N.k2, ei, ej, (uint32_t)max(W.r, 0));
        W.cs = 0;
    } else {
        pu = eu ? riw : riz;
        pv = ev ? rjw : rjz;
        W.r -= (int)(pu >> 31) + (int)(pv >> 31);
        W.cs = (int)(eu | (ev << 1));
    }
    W.pu = pu; W.pv = pv;
    park_u = lane == slot ? pu : park_u;                                // lane t keeps the t-th level's words: one store per batch
    park_v = lane == slot ? pv : park_v;
}

// Prefetch workgroups.  With ~90 instructions per level the walk is bound by memory latency: the HBM round trip of its
// back-pointer load (two in flight) and the row-record load that leads to the next candidates (~0.34 us per level).  The
// walker cannot know its cell any earlier, but r only ever decreases along the walk, and rarely: the cell of a level
// some tens of levels down lies in plane r or r - 1 of that level, 2 * k2 rows of k2 back-pointers -- 16 KB on a 64-wide
// level.  Helper workgroups ON THE WALKER'S XCD (the L2 is per XCD; a helper wave on the walker's own CU was measured
// and is worse than none: 80 vs 54 ms, its misses queue in front of the walker's loads in the CU's in-order memory
// pipeline) read those planes, one load per 128-byte line with the data dropped, and the level's row records, keeping
// up to LEAN_AHEAD_MAX levels ahead of the position the walker publishes every 16 levels; the walker's loads then
// hit the L2.  A wrong guess (a third recombination inside the window) costs an ordinary miss, nothing else.
// Protocol (ChainSync, in global memory; relaxed agent-scope atomics -- hints, nothing depends on their order): the
// walker (block 0) publishes its XCC id and seq << 32 | level; every other block of the launch whose XCC id matches
// takes a ticket, the first LEAN_PREFETCHERS tickets prefetch, everybody else leaves.  seq (a per-launch number) keeps a
// helper from acting on the previous launch's words; whatever it reads, it only ever touches levels of [l_lo, l_hi].
constexpr int LEAN_AHEAD_MAX = 96, LEAN_AHEAD_BYTES = 3 << 19;      // window: at most 96 levels and ~1.5 MB of planes
constexpr int LEAN_PREFETCHERS = 8, LEAN_BLOCKS = 80, LEAN_POLL_EVERY = 4;
constexpr int LEAN_PUBLISH_MASK = 8;             // the walker publishes its position every 16 levels (every 8: 47.0 ms, 16: 44.5, 32: 44.9 on MHC-24)
struct ChainSync { unsigned long long pos; int r, xcc; int ticket[2]; int n_helpers, n_levels, pad_[2]; };   // (n_helpers, n_levels: DG_DEBUG statistics)

__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | 20); }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void lean_prefetch(const LevelDesc *__restrict__ descs, int l_hi, int l_lo, int RP, const uint16_t *__restrict__ bp,
                                              const char *__restrict__ rowrec, ChainSync *sy, int seq, int *dump /* LDS, 64 words */, int lane) {
    // a load whose data nobody wants: straight into an LDS dump word per lane -- no destination register that a later value
    // could be sharing when the data arrives, nothing to wait for
#define DG_DROP_LOAD(PTR) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(PTR), (__attribute__((address_space(3))) void *)dump, 4, 0, 0)
    unsigned long long pos;
    int spins = 0;
    do {                                                                // the walker's first words of THIS launch
        pos = __hip_atomic_load(&sy->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(pos >> 32) != seq) { __builtin_amdgcn_s_sleep(32); if (++spins > (1 << 20)) return; }
    } while ((int)(pos >> 32) != seq);
    if ((int)(uint32_t)pos == INT32_MIN) return;
    if (__hip_atomic_load(&sy->xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc_id()) return;
    int me = 0;
    if (lane == 0) me = atomicAdd(&sy->ticket[seq & 1], 1);
    me = __builtin_amdgcn_readfirstlane(me);
    if (me >= LEAN_PREFETCHERS) return;
    // this block's levels: l_hi - me - LEAN_PREFETCHERS * q.  Their descriptors are fetched 64 at a time, one per lane, so that no
    // level waits for a dependent descriptor load of its own; the position is polled every LEAN_POLL_EVERY levels (an L2 round trip)
    int done = 0, lw = l_hi, rw = RP - 1;
    bool live = true;
    for (int lp = l_hi - me; live && lp >= l_lo; lp -= LEAN_PREFETCHERS * 64) {
        const int my_l = max(lp - LEAN_PREFETCHERS * lane, l_lo);
        const LevelDesc &dd = descs[my_l];
        const int64_t bo = dd.bp_off * 2;
        int bo_lo = (int)bo, bo_hi = (int)(bo >> 32), kk = dd.k2, bb = dd.b0;
        for (int j = 0; j < 64 && live; ++j) {
            const int l = lp - LEAN_PREFETCHERS * j;
            if (l < l_lo) break;
            const int k2 = __builtin_amdgcn_readlane(kk, j), b0 = __builtin_amdgcn_readlane(bb, j);
            const char *lvl = (const char *)bp + (((int64_t)__builtin_amdgcn_readlane(bo_hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane(bo_lo, j));
            const int ahead = max(8, min(LEAN_AHEAD_MAX, LEAN_AHEAD_BYTES / (4 * k2 * k2)));
            for (bool first = true; live; first = false) {              // fresh position: every few levels, and while too far ahead
                if (!first || j % LEAN_POLL_EVERY == 0) {
                    pos = __hip_atomic_load(&sy->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lw = __builtin_amdgcn_readfirstlane((int)(uint32_t)pos);
                    rw = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sy->r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (lw == INT32_MIN || (int)(pos >> 32) != seq) live = false;
                }
                if (!live || lw - l <= ahead) break;
                __builtin_amdgcn_s_sleep(16);
            }
            if (!live) break;
            if (l > lw) continue;                                       // overtaken
            const int r_hi = min(max(rw, 0), RP - 1);
            const int row_bytes = 2 * k2, lines = (row_bytes + 127) >> 7;   // 128-byte lines of one (i, r) row of back-pointers
            const int n = 2 * k2 * lines;
            for (int t = lane; t < n; t += 64) {
                const int ln = t % lines, q = t / lines, i = q >> 1, r = max(r_hi - (q & 1), 0);
                const uint32_t off = (uint32_t)(((uint32_t)i * (uint32_t)RP + (uint32_t)r) * (uint32_t)k2) * 2u + (uint32_t)min(ln << 7, row_bytes - 2);
                DG_DROP_LOAD(lvl + (off & ~3u));
            }
            for (int t = lane; t < ((k2 * 16 + 127) >> 7); t += 64) {
                const uint32_t off = ((uint32_t)b0 << 4) + (uint32_t)min(t << 7, k2 * 16 - 4);
                DG_DROP_LOAD(rowrec + off);
            }
            ++done;
        }
    }
    if (lane == 0) { atomicAdd(&sy->n_helpers, 1); atomicAdd(&sy->n_levels, done); }
#undef DG_DROP_LOAD
}

__global__ __launch_bounds__(64) void dp_trace_chain_lean_kernel(const LevelDesc *__restrict__ descs, int l_hi, int l_lo, int RP, int R,
                                                                 const uint16_t *__restrict__ bp /* biased by the segment's first unit */,
                                                                 const int32_t *__restrict__ final_val /* non-null on the first call */,
                                                                 const uint4 *__restrict__ rowrec_, const uint32_t *__restrict__ in_edge,
                                                                 uint2 *__restrict__ path, ChainState *st, ChainSync *sy, int seq) {
    __shared__ int dump_s[64];
    const int lane = threadIdx.x & 63;
    const char *rowrec = (const char *)rowrec_;
    if (blockIdx.x != 0) { lean_prefetch(descs, l_hi, l_lo, RP, bp, rowrec, sy, seq, dump_s, lane); return; }
    int value, si, sj;
    LeanWalk W;
    W.cs = 0; W.bad = false; W.pu = W.pv = 0;
    if (final_val) { value = final_val[(int64_t)R * descs[l_hi].k2]; si = 0; sj = 0; W.r = R; }       // sink level, layout [i][r][j]: cell (0, R, 0)
    else { si = st->i; sj = st->j; W.r = st->r; value = st->value; }
#define DG_PUBLISH(LV, RV) do { __hip_atomic_store(&sy->r, (RV), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                                __hip_atomic_store(&sy->pos, ((unsigned long long)(uint32_t)seq << 32) | (uint32_t)(LV), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    if (lane == 0) {
        __hip_atomic_store(&sy->ticket[(seq + 1) & 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the next launch's counter
        __hip_atomic_store(&sy->xcc, xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        DG_PUBLISH((value != NEG_INF && value != CHAIN_CORRUPT) ? l_hi : INT32_MIN, W.r);
    }
    if (value != NEG_INF && value != CHAIN_CORRUPT) {
        LeanRegs A, B;
        bool first = true;
        for (int base = l_hi; base >= l_lo && !W.bad; base -= 56) {
            const int my_l = max(base - lane, l_lo);                   // 64 descriptors, 56 levels (a multiple of 8) per batch; the range's
            const LevelDesc &dd = descs[my_l];                          // first level stands in for what lies below it (clamped loads only)
            const int64_t ba = dd.bp_off * 2;                           // byte offset of the level's first unit from the biased base
            int ba_lo = (int)ba, ba_hi = (int)(ba >> 32), kk = dd.k2, bb = dd.b0;
            asm volatile("" ::"v"(ba_lo), "v"(ba_hi), "v"(kk), "v"(bb)); // descriptors complete before the walk (no wait inside the loop)
            const int n = min(56, base - l_lo + 1);
            uint32_t park_u = 0, park_v = 0;
#define DG_DESC(T) LeanDesc{(const uint16_t *)((const char *)bp + (((int64_t)__builtin_amdgcn_readlane(ba_hi, (T)) << 32) | (uint32_t)__builtin_amdgcn_readlane(ba_lo, (T)))), \
                           (uint32_t)__builtin_amdgcn_readlane(kk, (T)), (uint32_t)__builtin_amdgcn_readlane(bb, (T))}
            if (first) {                                                // prologue: exact loads of the starting cell
                const LeanDesc D0 = DG_DESC(0);
                lean_issue(A, D0.bp_level, rowrec, D0.b0, D0.k2, (uint32_t)RP * D0.k2, min((uint32_t)si, D0.k2 - 1u), min((uint32_t)sj, D0.k2 - 1u), (uint32_t)max(W.r, 0));
                first = false;
            }
            // unrolled by hand (8 steps per trip): the loop's back edge copies the registers of the loads in flight, and so
            // waits for them -- one un-overlapped step per trip
#define DG_LEAN_STEP(T, X, Y) { const LeanDesc DN = DG_DESC((T) + 1); lean_step(W, RP, DN, rowrec, in_edge, park_u, park_v, (T), lane, X, Y); }
#define DG_LEAN_STOP(T) (W.bad || (T) >= n)
            for (int t = 0; t < n; t += 8) {
                if (lane == 0 && (t & LEAN_PUBLISH_MASK) == 0) DG_PUBLISH(base - t, W.r);
                DG_LEAN_STEP(t, A, B);     if (DG_LEAN_STOP(t + 1)) break;
                DG_LEAN_STEP(t + 1, B, A); if (DG_LEAN_STOP(t + 2)) break;
                DG_LEAN_STEP(t + 2, A, B); if (DG_LEAN_STOP(t + 3)) break;
                DG_LEAN_STEP(t + 3, B, A); if (DG_LEAN_STOP(t + 4)) break;
                DG_LEAN_STEP(t + 4, A, B); if (DG_LEAN_STOP(t + 5)) break;
                DG_LEAN_STEP(t + 5, B, A); if (DG_LEAN_STOP(t + 6)) break;
                DG_LEAN_STEP(t + 6, A, B); if (DG_LEAN_STOP(t + 7)) break;
                DG_LEAN_STEP(t + 7, B, A); if (DG_LEAN_STOP(t + 8)) break;
            }
#undef DG_LEAN_STOP
#undef DG_LEAN_STEP
#undef DG_DESC
            if (lane < n && !W.bad) path[base - lane] = make_uint2(park_u, park_v);
            si = (int)(W.pu & 0x7FFFu); sj = (int)(W.pv & 0x7FFFu);
        }
        if (W.bad || W.r < 0) value = CHAIN_CORRUPT;
    }
    if (lane == 0) { DG_PUBLISH(INT32_MIN, 0); st->i = si; st->j = sj; st->r = W.r; st->value = value; }
#undef DG_PUBLISH
}

// levels in parallel over the whole grid; *out is zeroed by the host before the launch (value is written by block 0)
__global__ __launch_bounds__(256) void dp_trace_finish_kernel(const LevelDesc *__restrict__ descs, int L, const uint2 *__restrict__ path,
                                                              ColourCsr col, int cap_e, int32_t *__restrict__ edges /* 4*cap_e */,
                                                              const ChainState *st, TraceOut *out) {
    const int value = st->value;
    if (blockIdx.x == 0 && threadIdx.x == 0) out->value = value;
    if (value == NEG_INF || value == CHAIN_CORRUPT) return;
    int shet = 0, sinter = 0;
    for (int l = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x); l < L; l += (int)(gridDim.x * blockDim.x)) {
        const uint2 b = path[l];
        int i = 0, j = 0;                                             // destination cell at level l = predecessor recorded at l+1
        if (l < L - 1) { const uint2 nb = path[l + 1]; i = (int)(nb.x & 0x7FFFu); j = (int)(nb.y & 0x7FFFu); }
        const int pi = (int)(b.x & 0x7FFFu), pj = (int)(b.y & 0x7FFFu);
        const int wu = (int)(b.x >> 31), wv = (int)(b.y >> 31);
        const LevelDesc d = descs[l];
        // a hop that leaves its level (the lean walk does not check per step): the source ids of this level's hop AND the destination
        // ids taken from the hop recorded one level up, before either indexes a colour list
        if (pi >= d.k || pj >= d.k || i >= d.k2 || j >= d.k2) { out->corrupt = 1; continue; }
        const int u1 = d.a0 + pi, v1 = d.a0 + pj, u2 = d.b0 + i, v2 = d.b0 + j;
        if (d.delta_off >= 0) { shet += score_symd(col, u1, v1, u2, v2); sinter += score_inter(col, u1, v1, u2, v2); }
        const int reps = (l == L - 1) ? 1 : 0;
        for (int q = 0; q < reps + wu; ++q) {
            const int e = atomicAdd(&out->n_e, 1);
            if (e < cap_e) { edges[e] = l; edges[cap_e + e] = u1; edges[2 * cap_e + e] = u2; edges[3 * cap_e + e] = 0; }
        }
        for (int q = 0; q < reps + wv; ++q) {
            const int e = atomicAdd(&out->n_e, 1);
            if (e < cap_e) { edges[e] = l; edges[cap_e + e] = v1; edges[2 * cap_e + e] = v2; edges[3 * cap_e + e] = 1; }
        }
    }
    // the walked path re-scored from the colour lists (approximator.cpp:614-618): its total must be the DP value -- a check of the
    // whole chain sweep -> lattice -> walk on every run
    for (int sft = 32; sft > 0; sft >>= 1) { shet += __shfl_down(shet, sft); sinter += __shfl_down(sinter, sft); }
    if ((threadIdx.x & 63) == 0 && shet) atomicAdd(&out->s_het, shet);
    if ((threadIdx.x & 63) == 0 && (shet | sinter)) atomicAdd(&out->path_score, shet + sinter);
}

void trace_launch_warm_rows(const DpState &S, int lb, int le, hipStream_t s) {   // row records of destination levels [lb, le), at most ~200 MB worth
    const int64_t v0 = S.descs[lb].b0, v1 = (int64_t)S.descs[le - 1].b0 + S.descs[le - 1].k2;
    const int64_t n = std::min<int64_t>(v1 - v0, (int64_t)12 << 20);
    if (n > 0) hipLaunchKernelGGL(dp_warm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S.d_rowrec.as<uint4>() + (v1 - n), n);
}

void trace_launch_chain(const DpState &S, int l_hi, int l_lo, const uint16_t *bp_biased, const int32_t *final_val, hipStream_t s) {
    if (S.lean_chain)
        hipLaunchKernelGGL(dp_trace_chain_lean_kernel, dim3(LEAN_BLOCKS), dim3(64), 0, s, S.d_descs.as<LevelDesc>(), l_hi, l_lo, S.RP, S.R, bp_biased, final_val,
                           S.d_rowrec.as<uint4>(), S.d_in_edge.as<uint32_t>(), S.d_path.as<uint2>(), S.d_chain.as<ChainState>(),
                           (ChainSync *)(S.d_chain.as<char>() + 64), ++S.chain_seq);
    else
        hipLaunchKernelGGL(dp_trace_chain_spec_kernel, dim3(1), dim3(64), 0, s, S.d_descs.as<LevelDesc>(), l_hi, l_lo, S.RP, S.R, bp_biased, final_val,
                           S.d_rowrec.as<uint4>(), S.d_in_edge.as<uint32_t>(), S.d_path.as<uint2>(), S.d_chain.as<ChainState>());
}

void trace_debug_report(const DpState &S) {                            // DG_DEBUG: how much of the walk the helpers covered
    ChainSync sy;
    if (hipMemcpy(&sy, S.d_chain.as<char>() + 64, sizeof sy, hipMemcpyDeviceToHost) == hipSuccess)
        fprintf(stderr, "[dg] chain walk helpers: %d block-launches prefetched %d levels (since load)\n", sy.n_helpers, sy.n_levels);
}

void trace_launch_finish(const DpState &S, hipStream_t s) {
    (void)hipMemsetAsync(S.d_trace.p, 0, sizeof(TraceOut), s);
    hipLaunchKernelGGL(dp_trace_finish_kernel, dim3((unsigned)std::min(1024, (S.L + 255) / 256)), dim3(256), 0, s, S.d_descs.as<LevelDesc>(), S.L, S.d_path.as<uint2>(), colour_csr(S), S.cap,
                       S.d_edges.as<int32_t>(), S.d_chain.as<ChainState>(), S.d_trace.as<TraceOut>());
}

}  // namespace dgi
