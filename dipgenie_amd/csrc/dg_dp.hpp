// Diploid pair-of-paths DP: types and entry points shared by the translation units of the DP
// (dg_dp_tables.hip, dg_dp_delta.hip, dg_dp_sweep.hip, dg_dp_trace.hip, dg_dp_run.hip).
//
// Replaces the level loop + sink read-out of Approximator::diploid_dp_approximation_solver
// (/root/reference/src/approximator.cpp:532-716, 757-785).  Design:
//   * gather form: every destination cell (i2, j2, r2) of level l+1 reduces over in(u2) x in(v2) -- one wave per
//     (destination row, group of <= 64 column in-edges, chunk of recombination counts), lanes on the column
//     in-edges; in-edges are stored sorted by source position, so a lexicographic max over (value, in-edge ranks)
//     reproduces the reference's take-if total order (value desc, pred_i asc, pred_j asc, :657-659) without locks
//     or atomics, and no destination is ever "reset" (:565-576 disappears);
//   * rolling value state is 4 B/cell, layout [i][r][j] (j fastest); s_het / edge chains are not carried
//     (reference cell = 40 B).  Every cell streams one 2-byte back-pointer (ranks of the winning in-edges) to HBM
//     and a traceback walks the lattice from the sink, emitting the weighted-edge lists (:757-764, :673-692) and
//     re-deriving s_het from the colour lists of the L winning edge pairs;
//   * score deltas (:604-624) do not depend on r nor on other levels: a launch fills, for every transition that
//     touches a colour, the T x T matrix delta[e_u][e_v] (uint16), T = #in-edges of the destination level;
//   * one launch per level (the levels are a dependency chain); the host picks the chunk size RC per level from a
//     cost model and gives rows with many in-edges cooperative workgroups.  (Several levels per dispatch with row-completion
//     counters in place of the kernel boundary were built and measured 3-6x slower: profiles/r03_chained_dispatch_ab.txt.)
#pragma once
#include <condition_variable>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

#include "dg_internal.hpp"

namespace dgi {

constexpr int32_t NEG_INF = INT32_MIN / 4;              // approximator.cpp:413
constexpr uint32_t BP_NONE = 0xFFFFFFFFu;              // wide (32-bit) back-pointer of an unreachable cell
constexpr int MAX_K = 1 << 15;                          // hop words pack positions in 15 bits
// Back-pointers are 16 bits per cell: (eu << 8) | ev, the ranks of the winning in-edges inside the destination row's
// and column's in-edge lists (sorted by source position, so rank order IS the reference's (pred_i asc, pred_j asc)
// tie order); 0xFFFF = unreachable.  Only a level with an in-degree > 255 keeps the wide word
// pred_i | pred_j << 15 | wu << 30 | wv << 31 (two 16-bit units per cell) and runs on the generic kernel.
constexpr int BP_MAX_RANK = 255;
constexpr int COOP_MIN = 8;                             // rows with more in-edges get cooperative workgroups
constexpr int HEAVY_INLINE = 8;                         // heavy rows of a level whose index rides in the kernel arguments
constexpr int ROWX_MAX = 64;                            // widest row in-edge matrix (in-edges per row)
constexpr int DELTA_PER_BLOCK = 256 * 16;
constexpr int DELTA_PAD = 8;                            // delta[0..8) stays zero: the colourless transitions' slot
constexpr int32_t CHAIN_CORRUPT = INT32_MIN;            // ChainState::value after a hop left its level

struct LevelDesc {                                      // transition (l-1) -> l, indexed by l; passed BY VALUE to the sweep
    int32_t a0, k;                                      // source level: first vertex id, width
    int32_t b0, k2;                                     // destination level
    uint32_t in_base;                                   // first in-edge of the destination level
    int32_t T;                                          // in-edges into the destination level
    int64_t delta_off;                                  // offset of the T*T uint16 matrix, -1 if all zero
    int64_t bp_off;                                     // offset of this level in the bp lattice, in 16-bit units (even)
    int32_t grp_first, ngroups;                         // column groups (runs of <=64 in-edges covering whole columns)
    int32_t dead_first, ndead;                          // destination columns with no in-edge
    int64_t slot_first;                                 // first entry of this level in the 64-wide slot table
    int32_t fast_ok, nblocks;                           // fast kernel usable; number of 64-slot blocks (>= ngroups: giant columns take several)
    int32_t heavy_first, n_heavy;                       // rows with more than COOP_MIN in-edges: slice of the heavy-row table
    int32_t bp_wide, bp_nt;                             // wide back-pointers on this level; stream them with non-temporal stores
    // Row in-edge matrix: rowx[rowx_off + i2 * rowx_stride + t] = t-th in-edge word of destination row i2 (0 beyond its
    // in-degree).  Its address needs nothing but kernel arguments, so the in-edge list of a fan-in row arrives in the
    // task's first load round together with the row and slot records (rowx_stride = 0: the level has no such matrix and
    // rows with more than two in-edges fetch their list from in_edge[] once the row record is in).
    int64_t rowx_off;
    int32_t rowx_stride;
    int16_t heavy_in[HEAVY_INLINE];                     // the level's first heavy rows (cooperative region: no table lookup)
    int32_t dmax;                                       // largest in-degree among the level's vertices (host: choice of RC)
};

struct TraceOut { int32_t value, s_het, n_e, overflow, corrupt, path_score; };   // path_score: sum of the score deltas along the walked path (must equal value)
struct ChainState { int32_t i, j, r, value; };
struct ColourCsr { const int64_t *hom_off, *het_off; const int32_t *hom_col, *het_col; };

struct SweepArgs {                                      // generic sweep kernel
    const LevelDesc *descs;
    const uint32_t *in_off, *in_edge, *grp_begin;
    const int32_t *in_dst, *dead_cols;
    const uint16_t *delta, *delta_zero;                 // delta: biased so that delta[d.delta_off] is valid for the resident window
    char *ring;                                         // the two state slots (ping-pong: level l in slot l & 1), each slot_bytes long, data starts pad_bytes in
    size_t slot_bytes, pad_bytes;
    uint16_t *bp;
    unsigned long long *digest;
    int RP;
    int *progress;                                      // PfCtl::level: the level whose launch is running (dg_dp_sweep.hip: L2 prefetcher)
};

struct FastArgs {                                       // fast sweep kernel
    const uint4 *rowrec;
    const uint2 *slots;
    const uint32_t *in_edge, *rowx;
    const int32_t *dead_cols;
    const uint16_t *delta, *delta_zero;
    char *ring;                                         // padded allocation start of state slot 0; slot s starts at ring + s * slot_bytes
    uint16_t *bp;                                       // fast-form levels always store narrow back-pointers
    unsigned long long *digest;
    int RP, pad_bytes;                                  // pad_bytes: front padding of the state buffers
    uint32_t buf_bytes;                                 // size of one padded state buffer (what a buffer resource covers)
    uint32_t slot_bytes;                                // distance between two state slots
    int *progress;                                      // PfCtl::level: the level whose launch is running (L2 prefetcher)
#ifdef DG_SWEEP_PROBE
    unsigned long long *probe;                          // measurement build: 8 words per level
#endif
};

// DP states alive per device.  The side-stream features (L2 table prefetcher, score deltas beside the sweep) assume that nothing
// else shares the device's hardware queues with the sweep: with a second state on the SAME device a side-stream kernel of one may
// sit in front of the other's sweep.  States on other devices do not matter.
inline std::atomic<int> &dp_states_on_device(int device) { static std::atomic<int> n[64]; return n[device & 63]; }

struct DpState {
    explicit DpState(int dev) : device(dev) { ++dp_states_on_device(device); }
    ~DpState() { --dp_states_on_device(device); }
    const int device;
    int64_t side_stream = -1;                           // side_stream: -1 = on while this is the device's only DP state (and the concurrency probe agrees), 0 = off, 1 = on
    bool side_stream_ok() const { return side_stream < 0 ? dp_states_on_device(device).load() <= 1 : side_stream != 0; }
    DpState(const DpState &) = delete;
    DpState &operator=(const DpState &) = delete;
    int32_t nV = 0, L = 0, R = 0, RP = 0, cap = 0;
    int32_t rp_active = 0;                              // planes [0, rp_active) are swept by the fast kernels (= RP except while a segment is re-swept below its path's plane)
    int64_t plane_limit = 1;                            // plane_limit: beyond HBM, re-sweep every segment only up to the plane its path leaves it on (0: all planes)
    bool loaded = false;
    // ---- options (dg_dp_set_option) ----
    int64_t want_digest = 0;                            // digest: accumulate per-level digests (values + back-pointers)
    int64_t use_fast = 1;                               // fast: 0 forces the generic kernel
    int64_t adaptive_rc = 1;                            // adaptive_rc: 0 = one chunk of all recombination counts per task
    int64_t use_coop = 1;                               // coop: cooperative tasks for rows with many in-edges (2: whenever possible)
    int64_t max_blocks = 1024;                          // max_blocks: grid of the generic kernel
    int64_t segment_cells = 0;                          // segment_cells: force lattice segments of at most this many cells (tests)
    int64_t host_threads = 16;                          // host_threads: threads of dg_dp_load_graph's host table construction
    int64_t test_poison_level = 0, test_poison_byte = 0xFF;   // test_poison_*: overwrite one level of the lattice between sweep and walk (tests of the corrupt-lattice path)
    // (measurement build -DDG_SYM only; the product library has neither the kernels nor the options)
    int64_t use_sym = 1;                                // sym: symmetric form of the sweep on wide levels (1: levels at least sym_min_k2 wide, 2: wherever the form exists, 0: off)
    int64_t sym_fold = 0;                               // sym_fold: 1 = the (tile, block) pairs above the diagonal dealt to the grid without holes (lean levels; measured 2 % slower than the rectangular grid with its idle workgroups)
    int64_t sym_dbg = 0;                                // (experiments: 1 no fan-in workgroups, 4 no mirror stores, 8 no gathers, 16 empty kernel, 32 first load round only -- results void)
    int64_t sym_min_k2 = 160, sym_rc = 4;               // sym_min_k2, sym_rc: width from which a level takes the symmetric form; recombination counts per task there (1-4, 6, 8)
    int64_t host_tables = 0;                            // host_tables: 1 = build the tables on the host and upload them (dg_dp_tables.hip; parity twin of dg_dp_build.hip)
    int64_t bp_nt_min_cells = 262144;                   // bp_nt_min_cells: levels this big stream their back-pointers non-temporally
    int64_t graph_batch = -1;                           // graph_batch: levels per captured hipGraph (0 = plain launches, -1 = the default of 1,000)
    int64_t warm_ahead = 128;                           // warm_ahead: sweep look-ahead, levels per batch (0 = off)
    int64_t sync_every = 0;                             // sync_every: drain the stream every N level launches (profiler aid)
    int64_t l2_prefetch = 6;                            // l2_prefetch: levels the per-XCD table prefetcher runs ahead of the sweep (0: off)
    int64_t use_lean_chain = 1;                         // lean_chain: 1 the lean walk where the lattice allows it, 0 always the general one; next load
    bool lean_chain = false;
    // single-window score deltas computed beside the sweep: piece k (transitions of levels >= delta_piece_level[k]) signals delta_piece_ev[k]
    std::vector<hipEvent_t> delta_piece_ev;
    std::vector<int32_t> delta_piece_level;
    int delta_piece_next = 0;                           // first piece the sweep has not waited for yet
    int64_t delta_overlap = 1;                          // delta_overlap: 0 = the whole window before the sweep (option)
    hipStream_t pf_stream = nullptr;                    // the ONE side stream: score-delta pieces, then the L2 table prefetcher (dg_dp_sweep.hip)
    hipEvent_t pf_ev = nullptr;
    int pf_seq = 0;
    bool pf_active = false;                             // a prefetcher accompanies the sweep range being issued
    int64_t pf_far = 128;                               // pf_far: > 0 = prefetcher blocks also pull the tables pf_far levels ahead into the Infinity Cache (then no periodic look-ahead launches)
    int pf_tested = 0;                                  // 0: the side stream's concurrency with the sweep's stream not yet probed, 1: probed
    mutable int chain_seq = 0;                          // per-launch number of the lean chain walk (ChainSync, dg_dp_trace.hip)
    int64_t use_rowx = 1;                               // rowx: row in-edge matrices (0: every fan-in row fetches its list from in_edge[])
    int64_t delta_cap_entries = (int64_t)4 << 30;       // delta_cap_entries: budget of resident score-delta entries
    int64_t rc_cap = 65536, rc_t0_ns = 3000, rc_tg_ps = 20000, rc_tw_ps = 50;   // rc_*: cost model of the per-level RC choice
    size_t chunk_units_cfg = (size_t)4 << 30;           // lattice_chunk_cells: size of one lattice chunk (16-bit units)
    // ---- lattice segments: destination levels [seg_begin[s], seg_begin[s+1]); one segment = whole lattice resident.
    // More than one = checkpoint + recompute (value-only pass, then each segment re-swept with back-pointers, last first).
    std::vector<int> seg_begin;
    std::vector<int64_t> ckpt_off;                     // element offset of checkpoint s (state of level seg_begin[s]-1)
    bool graph_failed = false;                          // capture or instantiation failed once: plain launches from then on
    typedef std::tuple<int, int, const void *, int> GraphKey;     // (first level, end level, biased lattice pointer, look-ahead launches left to the prefetcher | planes swept << 1)
    std::map<GraphKey, hipGraphExec_t> graphs;                    // -> replayable batch
    std::map<GraphKey, std::vector<int64_t>> graph_hist;          // -> its launches by kernel variant (launch_hist)
    bool all_fast = false;
    size_t state_alloc_bytes = 0;
    std::vector<LevelDesc> descs;
    uint64_t cells = 0, relaxations = 0, edge_pairs = 0, colour_entries = 0;
    int64_t total_units = 0, max_level_units = 0;       // back-pointer lattice, in 16-bit units (1 per cell, 2 on wide levels)
    int64_t max_level_cells = 0, delta_entries = 0, n_delta_blocks = 0, pad_front = 0;
    std::vector<int64_t> level_units;                   // units of every level (even)
    // score-delta windows: coloured transitions [dwin_t[w], dwin_t[w+1]) are resident together (one window = everything
    // unless the matrices outgrow delta_cap_entries; then each window is recomputed right before its first level)
    std::vector<int32_t> dwin_t, level_win;             // level_win[l] = window of level l's transition, -1 if colourless
    int cur_win = -1;                                   // window whose matrices are in d_delta right now
    std::vector<int32_t> dtrans_host;
    std::vector<int64_t> dblk_first_host;
    int64_t delta_buf_entries = 0;
    std::vector<int32_t> level_dmax;                    // largest in-degree among the level's vertices
    int64_t n_grp = 0, n_dead = 0, n_heavy_rows = 0, n_slot_records = 0, n_rowx_words = 0, n_dtrans = 0, n_edges = 0;   // logical table sizes (dg_dp_get_table_digest)
    DevBuf d_descs, d_in_off, d_in_edge, d_in_dst, d_hom_off, d_het_off, d_hom_col, d_het_col, d_eflag, d_eself;
    DevBuf d_delta, d_bp, d_ring, d_digest, d_trace, d_edges, d_dblk_first, d_dtrans, d_grp, d_dead, d_heavy, d_rowrec, d_rowx, d_slots, d_path, d_ckpt, d_chain, d_pfctl;
#ifdef DG_SWEEP_PROBE
    DevBuf d_probe;
#endif
    std::vector<uint64_t> digest_host;
    // launches of the last run per sweep kernel variant: index = rc * 4 + general * 2 + coop (rc 0 = generic kernel)
    int64_t launch_hist[64 * 4] = {};
    dg_dp_timing timing;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // The resident back-pointer lattice lives in a pool of equal chunks that a background thread allocates one by
    // one (a 100+ GB hipMalloc takes seconds, more when another process has just freed HBM): reservation can start
    // before the graph exists (dg_dp_prealloc) without starving the small allocations of the sketch stage.
    // Levels never straddle chunks.
    struct Pool {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<void *> chunks;           // each chunk_units * 2 bytes
        size_t chunk_units = (size_t)4 << 30;  // 8 GB
        size_t target = 0;                     // chunks wanted
        size_t cap_chunks = 0;                 // upper bound for reservations made before the graph is known
        bool running = false, failed = false, paused = false;
        std::thread th;
    } pool;
    std::vector<int> chunk_begin;              // destination levels [chunk_begin[c], chunk_begin[c+1]) live in chunk c
    int seg_chunks = 1;                        // chunks per lattice segment (= all of them when the lattice is resident)
};

inline ColourCsr colour_csr(const DpState &S) {
    return ColourCsr{S.d_hom_off.as<int64_t>(), S.d_het_off.as<int64_t>(), S.d_hom_col.as<int32_t>(), S.d_het_col.as<int32_t>()};
}

// ---- lattice chunk pool (dg_dp_run.hip) ----
void pool_request(DpState &S, int device, size_t target);
void pool_trim(DpState &S, size_t keep);
inline void pool_clear(DpState &S) { pool_trim(S, 0); }
void *pool_wait(DpState &S, size_t c);
struct PoolPause {                       // hipMalloc calls queue behind the one in flight: hold the pool thread between chunks
    DpState &S;
    explicit PoolPause(DpState &s);
    ~PoolPause();
};
void graphs_clear(DpState &S);
double wall_s();

// ---- table construction (dg_dp_tables.hip: entry, host construction, lattice plan; dg_dp_build.hip: device construction) ----
int dp_load(dg_ctx *c, const dg_dp_graph *g);
int dp_build_tables_device(dg_ctx *c, const dg_dp_graph *g, DpState &S, std::vector<int32_t> &dtrans, std::vector<int64_t> &dblk_first, int &max_k);
int dp_table_digest(dg_ctx *c, uint64_t *out, int n);

// ---- score deltas (dg_dp_delta.hip) ----
void delta_launch_edge_flags(const DpState &S, hipStream_t s);
// (re)computes the matrices of window w into d_delta; returns the pointer biased so that ptr[LevelDesc::delta_off] is valid
const uint16_t *delta_launch_window(DpState &S, int w, hipStream_t s);
const uint16_t *delta_launch_overlapped(DpState &S, hipStream_t s);          // one window: first piece on s, the rest on a side stream (events in delta_piece_*)
void delta_overlap_free(DpState &S);

// ---- level sweep (dg_dp_sweep.hip) ----
struct SweepLaunch {                     // per-run launch context
    SweepArgs A;
    FastArgs F;
    bool small_state = true;
    int rc_sel = 19;                     // chunk of "all recombination counts" (8 / 19 / 33 instantiations)
};
void sweep_prepare(const DpState &S, SweepLaunch &X);
void sweep_init_state(const DpState &S, hipStream_t s);                  // level 0: every r starts at 0 (:534-535)
void sweep_launch_level(DpState &S, SweepLaunch &X, int l, hipStream_t s);
bool sweep_launch_sym(DpState &S, SweepLaunch &X, int l, hipStream_t s);    // measurement build (-DDG_SYM, dg_dp_sweep_sym.hip): the symmetric form, where it applies
void sweep_warm_tables(const DpState &S, const SweepLaunch &X, int q0, int q1, hipStream_t s);
int sweep_prefetch_begin(DpState &S, const SweepLaunch &X, int lb, int le, bool delta_resident, hipStream_t s);
void sweep_prefetch_end(DpState &S, int le, hipStream_t s);
void sweep_prefetch_free(DpState &S);

// ---- traceback (dg_dp_trace.hip) ----
void trace_launch_warm_rows(const DpState &S, int lb, int le, hipStream_t s);
void trace_launch_chain(const DpState &S, int l_hi, int l_lo, const uint16_t *bp_biased, const int32_t *final_val, hipStream_t s);
void trace_launch_finish(const DpState &S, hipStream_t s);
void trace_debug_report(const DpState &S);

}  // namespace dgi
