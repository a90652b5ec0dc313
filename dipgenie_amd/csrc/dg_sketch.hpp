// State of the sketching half of libdipgenie_hip.so, shared by dg_sketch.hip (tiles) and dg_sketch_spectrum.hip (buckets).
#pragma once
#include "dg_internal.hpp"

namespace dgi {

constexpr int FILL_PAD = 16;            // words between two buckets' fill counters (a 64-byte line each)
// where the tile kernel puts a read minimizer in bucket form: bucket = top bbits of the hash, slot = fill[bucket]++ (< stride)
// (a bucket that is full spills into one shared list: slot = (*spill_n)++ < spill_cap)
struct BucketEmit { uint32_t *fill = nullptr; uint64_t *bk_hash = nullptr; uint32_t *bk_read = nullptr; int bbits = 1; uint32_t stride = 0;
                    uint32_t *spill_n = nullptr; uint64_t *spill_hash = nullptr; uint32_t *spill_read = nullptr; uint32_t spill_cap = 0; };
struct BucketPlan { bool ok = false, has_multi = false; int bbits = 0, sbits = 0, B = 0, G = 0; uint32_t stride = 0, residual_cap = 0, spill_cap = 0; };

struct SketchState {
    DevBuf d_bases, d_off, d_seq_tiles, d_seq_wins, d_seq_tile0, d_seq_win0, d_tiles, d_tile_cnt, d_tile_base, d_tile_sparse, d_hash, d_aux, d_hash2, d_aux2, d_tmp, d_flag, d_uniq, d_cnt, d_n, d_kmers, d_out;
    // bucketed spectrum (dg_sketch_spectrum.hip): bucket arrays, the (workgroup x bucket) count matrix, per-bucket tables
    DevBuf d_bk_hash, d_bk_read, d_matrix, d_bk_start, d_bk_fill, d_bk_dcount, d_bk_dstart, d_bk_ovf, d_bk_status, d_spill_hash, d_spill_read;
    bool attr_set = false;              // the kernels' dynamic-LDS limits are raised once per ctx
    bool sticky_exact = false;          // a bucket ran over its stride once: this ctx places buckets exactly from then on
    int64_t *h_status = nullptr;        // pinned: {pairs, distinct hashes, buckets left to the host, a bucket ran over its stride}
    // options (dg_sketch_set_option) and what the last dg_sketch_reads* call did (dg_sketch_get_stat)
    int opt_mode = 0, opt_bucket_bits = 0, opt_stride = 0, opt_residual_cap = 0, opt_host_buckets = 0;
    int64_t opt_spill_cap = 0;
    int64_t stat_path = 0, stat_buckets = 0, stat_overflow = 0, stat_spilled = 0;
    dg_sketch_timing timing;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

// dg_sketch_spectrum.hip -- Sp_R (sorted distinct hashes + the number of distinct reads holding each) from bucketed pairs.
// Inputs outside what the bucket path packs (>= 2^31 reads or windows) get plan->ok = false.  n_multi: reads of several tiles.
void bucket_plan(const SketchState &S, int64_t n_reads, int64_t n_tiles, int64_t n_win, int64_t n_multi, int w, BucketPlan *plan);
// tile kernel fills the buckets itself: arrays of B * stride pairs, zeroed fill counters
int bucket_fast_begin(dg_ctx *c, SketchState &S, const BucketPlan &plan, BucketEmit *be);
// buckets placed exactly from the sparse tile output (d_tile_sparse / d_tile_cnt / d_hash2 / d_aux2): count, scan, scatter
int bucket_exact_scatter(dg_ctx *c, SketchState &S, const BucketPlan &plan, int64_t n_tiles);
// per-bucket tables in LDS -> dense sorted output in out_hash / out_cnt (nullptr: d_uniq / d_cnt).  *outcome: 0 done,
// 1 a bucket ran over its stride (fast form only; nothing usable was written), 2 too many buckets need the host (use the generic path)
int bucket_finish(dg_ctx *c, SketchState &S, const BucketPlan &plan, bool fast, uint64_t *out_hash, int32_t *out_cnt, int64_t cap,
                  int64_t *n_distinct, int64_t *n_emitted, int *outcome);

}  // namespace dgi
