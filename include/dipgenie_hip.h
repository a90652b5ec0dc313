/* include/dipgenie_hip.h -- C ABI of libdipgenie_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for DipGenie's diploid hot path.  The reference has no FFI/plugin layer; the
 * seams these entry points replace are C++ member calls (all citations relative to the reference):
 *
 *   dg_dp_solve_diploid     replaces the level loop + sink read-out of
 *                           Approximator::diploid_dp_approximation_solver
 *                           (src/approximator.h:26, src/approximator.cpp:532-716 and :757-785)
 *   dg_dp_solve_haploid     replaces the scatter loop of Approximator::dp_approximation_solver
 *                           (src/approximator.h:25, src/approximator.cpp:44-72)
 *   dg_anchor_*             replace the vertex-span mapping of Solver::index_kmers, Solver::compute_anchors, the
 *                           shared-anchor filter and the occurrence sort (src/solver.cpp:343-357, 415-446, 560-663)
 *   dg_sketch_reads         replaces the per-read Solver::compute_hashes loop and the Sp_R /
 *                           kmer_count maps (src/solver.h:101, src/solver.cpp:526-546, 711-732)
 *   dg_sketch_haplotype     replaces the window loop of Solver::index_kmers
 *                           (src/solver.h:100, src/solver.cpp:302-361); the position -> vertex-span
 *                           mapping (:343-357) stays in the host code
 *   dg_hash_kmers           exposes hash128_to_64_ (src/solver.cpp:16-24) for known-answer tests
 *
 * Conventions: plain pointers and sizes only; `int` return (0 = ok, <0 = error, message from
 * dg_last_error()); no exceptions cross the boundary; one dg_ctx <-> one HIP device + stream; a ctx
 * is not thread-safe, different ctxs are independent.  "host" pointers are ordinary process memory;
 * "_dev" entry points take device pointers (e.g. torch tensors' data_ptr()) and run on the ctx
 * stream without synchronising unless stated.  There is NO CPU fallback: every entry point fails
 * with DG_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef DIPGENIE_HIP_H
#define DIPGENIE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DG_OK              0
#define DG_ERR_ARG        -1
#define DG_ERR_NO_DEVICE  -2
#define DG_ERR_HIP        -3
#define DG_ERR_OOM        -4
#define DG_ERR_UNSUPPORTED -5
#define DG_ERR_STATE      -6

typedef struct dg_ctx dg_ctx;

/* ---- context ---- */
dg_ctx     *dg_create(int device);                 /* NULL on failure (see dg_last_error) */
void        dg_destroy(dg_ctx *);
const char *dg_last_error(void);                   /* thread-local message of the last failure */
int         dg_set_stream(dg_ctx *, void *hip_stream);   /* adopt an external hipStream_t (e.g. torch's) */
int         dg_synchronize(dg_ctx *);
int         dg_device_info(dg_ctx *, char *name, int name_cap, int *n_cu, int64_t *hbm_bytes);
int         dg_hip_versions(int *compiled, int *runtime);   /* HIP_VERSION of the build / of the runtime bound in this process (major * 10^7 + minor * 10^5 + patch) */

/* ---- diploid pair-of-paths DP ---- */
typedef struct dg_dp_graph {          /* levelized expanded graph, vertex ids already level-sorted */
    int32_t n_vertices, n_levels, R;
    const int32_t *level_off;         /* [n_levels+1]; level 0 = {source}, last level = {sink} */
    const int64_t *out_off;           /* [n_vertices+1] out-CSR, adjacency order preserved */
    const int32_t *out_dst;           /* every edge goes from level l to level l+1 */
    const uint8_t *out_w;             /* recombination weight 0/1 */
    const int64_t *hom_off, *het_off; /* [n_vertices+1] sorted-unique colour CSR (HOM / HET colours) */
    const int32_t *hom_col, *het_col;
} dg_dp_graph;

typedef struct dg_dp_result {         /* sink state at r = R (approximator.cpp:774-785) */
    int32_t value, s_het, n_p1, n_p2;
    int32_t *p1_from, *p1_to, *p2_from, *p2_to;   /* caller-provided, capacity `cap` each (>= R+2) */
    int32_t cap;
    uint64_t cells, relaxations;      /* work counters: sum k^2(R+1), sum (sum outdeg)^2 (R+1) */
} dg_dp_result;

typedef struct dg_dp_timing {         /* HIP-event times of the last dg_dp_run, milliseconds */
    float delta_ms;                   /* score-delta precompute kernel(s) */
    float forward_ms;                 /* level sweep (segmented lattice: the value-only first pass) */
    float traceback_ms;               /* back-pointer walk + edge-list extraction (segmented lattice: plus the
                                         re-sweeps of the segments with back-pointers) */
    float total_ms;                   /* first launch -> last kernel done */
    int64_t n_forward_launches;
    uint64_t edge_pairs;              /* sum over levels of (in-edges into level)^2 */
    uint64_t colour_entries;          /* colour list entries read by the delta kernel */
    uint64_t state_bytes, bp_bytes, delta_bytes;   /* device allocations */
    int32_t n_segments;               /* 1: the back-pointer lattice was resident; > 1: checkpoint + recompute in that many segments */
    int32_t n_chunks;                 /* lattice chunks the levels were packed into */
} dg_dp_timing;

/* optional: reserve about `bytes` of back-pointer lattice (bytes <= 0: 60 % of the free HBM) in a background
 * thread, in 8 GB chunks; may be called again with a better figure (the latest call wins, chunks already mapped
 * are kept). Mapping 100+ GB takes seconds and stalls every other HIP call meanwhile, so the CLI issues it where
 * only host work follows; dg_dp_load_graph adopts the chunks and dg_dp_run waits for any still missing. */
int dg_dp_prealloc(dg_ctx *, int64_t bytes);
int dg_dp_load_graph(dg_ctx *, const dg_dp_graph *);   /* validate + upload + build in-CSR; resident until next load */
int dg_dp_run(dg_ctx *, dg_dp_result *);               /* all kernels on the resident graph; synchronises */
int dg_dp_get_timing(dg_ctx *, dg_dp_timing *);
int dg_dp_solve_diploid(dg_ctx *, const dg_dp_graph *, dg_dp_result *);   /* = load_graph + run */
/* debug/parity: copy the per-level digest (same definition as the oracle's level_digest) of the
 * last run; out has n_levels entries, entry 0 unused. Requires dg_dp_set_option("digest",1). */
int dg_dp_get_level_digest(dg_ctx *, uint64_t *out, int64_t n);
/* parity / test / tuning knobs of the DP (none is needed in normal use; unknown keys fail with DG_ERR_ARG):
 *   digest 0|1            accumulate the per-level digests
 *   fast 0|1              0: generic sweep kernel only          adaptive_rc 0|1   0: one chunk of all r per task
 *   coop 0|1|2            cooperative fan-in rows off / by cost model / whenever possible
 *   rowx 0|1              row in-edge matrices (next load)      lean_chain 0|1    1: lean chain walk where the lattice allows it, 0: the general one (next load)
 *   graph_batch n         levels per hipGraph batch (-1: default 1000, 0: plain launches)
 *   l2_prefetch n         levels the per-XCD table prefetcher runs ahead of the sweep (0: off)
 *   pf_far n              levels ahead at which the prefetcher's far blocks pull tables into the Infinity Cache (0: periodic launches instead)
 *   delta_overlap 0|1|2   score deltas beside the sweep: off / on graphs of >= 32,000 levels / whenever possible (tests)
 *   warm_ahead n          levels per Infinity-Cache look-ahead batch (0: off)
 *   segment_cells, lattice_chunk_cells, delta_cap_entries   force checkpoint + recompute / chunk size / delta windows (tests)
 *   plane_limit 0|1       lattice beyond HBM: re-sweep every segment only up to the recombination plane its path leaves it on (default 1; 0: all planes)
 *   sync_every n          drain the stream every n level launches (rocprofv3 --pmc)
 *   side_stream -1|0|1    L2 prefetcher + score deltas beside the sweep: -1 (default) while this is the only DP state on its device, 0 never, 1 always
 *   test_poison_level l, test_poison_byte b   tests: fill level l of the back-pointer lattice with byte b between sweep and walk (dg_dp_run must answer DG_ERR_STATE)
 *   host_tables 0|1       0 (default): the sweep's tables are built by device kernels from the uploaded graph; 1: on the host, then uploaded (parity twin; next load)
 *   rc_t0_ns, rc_tg_ps, rc_tw_ps, rc_cap, bp_nt_min_cells, max_blocks, host_threads   cost model / launch tuning */
int dg_dp_set_option(dg_ctx *, const char *key, int64_t value);
/* parity: FNV-1a digests of the 12 tables built by dg_dp_load_graph (level descriptors, in-CSR offsets / sources / destinations,
 * coloured transitions, their delta blocks, column groups, dead columns, heavy rows, row records, row in-edge matrices, slot
 * records): the device construction and the host construction (option host_tables) must agree.  out has n >= 12 words. */
int dg_dp_get_table_digest(dg_ctx *, uint64_t *out, int n);
/* measurement: which sweep kernel variants the last dg_dp_run launched, as "name:count name:count ..." (the names
 * rocprofv3 reports, abbreviated); lets a profile taken in another process be matched against this run. */
int dg_dp_get_launch_profile(dg_ctx *, char *buf, int cap);

/* ---- haploid (vertex, r) DP ---- */
typedef struct dg_hap_graph {         /* expanded graph after topologically_reorder: every edge u -> v has u < v */
    int32_t n_vertices, R;
    const int64_t *out_off;           /* [n_vertices+1] out-CSR, adjacency order preserved */
    const int32_t *out_dst;
    const uint8_t *out_w;             /* recombination weight 0/1 */
    const int32_t *n_colours;         /* |color[v]| */
} dg_hap_graph;
/* replaces the scatter loop of Approximator::dp_approximation_solver (src/approximator.cpp:44-72): fills the caller's
 * dp / back_vtx / back_r arrays, each [n_vertices * (R+1)], index v * (R+1) + r; the per-r backtracks and the choice of
 * best_r (:74-153, double arithmetic) stay with the caller.  Synchronises. */
int dg_dp_solve_haploid(dg_ctx *, const dg_hap_graph *, int32_t *dp, int32_t *back_vtx, int32_t *back_r);

/* ---- (w,k)-minimizer sketching ---- */
/* reads: concatenated bases + offsets [n_reads+1] (host). Outputs (malloc'ed by the library, release
 * with dg_free): globally sorted distinct minimizer hashes and the number of reads containing each
 * (== Sp_R keys in order / kmer_count values). */
int dg_sketch_reads(dg_ctx *, const char *bases, const int64_t *read_off, int64_t n_reads, int k, int w,
                    uint64_t **hash, int32_t **n_reads_with_hash, int64_t *n_distinct);
/* one haplotype string (host): the emitted-on-hash-change minimizer list in sequence order.
 * pos = start of the winning k-mer. Outputs malloc'ed by the library (dg_free). */
int dg_sketch_haplotype(dg_ctx *, const char *seq, int64_t len, int k, int w,
                        uint64_t **hash, int64_t **pos, int64_t *n);
/* hash n k-mers of length k stored back to back (host) with h1^h2 of MurmurHash3_x64_128, seed 0 */
int dg_hash_kmers(dg_ctx *, const char *kmers, int64_t n, int k, uint64_t *out);
void dg_free(void *);

typedef struct dg_sketch_timing { float kernel_ms, sort_ms, total_ms; int64_t n_emitted; } dg_sketch_timing;
int dg_sketch_get_timing(dg_ctx *, dg_sketch_timing *);
/* parity / test knobs of the read spectrum (Sp_R, src/solver.cpp:526-546; none is needed in normal use):
 *   spectrum_mode m        0 (default): the tile kernel drops every minimizer into the bucket of its hash range, one LDS table
 *                          per bucket resolves it (dg_sketch_spectrum.hip); full buckets spill into one shared list, and only when
 *                          that runs over is the pass repeated with exact placement.  2: exact placement at once.  1: the generic path, a stable 64-bit radix sort of
 *                          all (hash, read) pairs + reduce-by-key.  The output is the same bit for bit.
 *   bucket_bits b          0 (default): buckets sized to the input; 1..15: 2^b buckets
 *   bucket_stride n        0 (default): 12288 slots per bucket in mode 0
 *   spill_cap n            0 (default): 2^20 pairs in the shared spill list of full buckets; -1: none (a full bucket repeats the pass
 *                          with exact placement at once)
 *   residual_cap n         0 (default): 1024 residual entries per bucket (third hashes of a table entry); fewer (-1: none)
 *                          leave more buckets to the host's per-segment finish
 *   host_buckets n         0 (default): up to 256 buckets may be left to the host before the generic path takes over; 1..256
 * dg_sketch_get_stat names, about the last dg_sketch_reads / dg_sketch_reads_dev call: spectrum_path (0 buckets filled by the
 * tile kernel, 1 exact placement, 2 generic), buckets, overflow_buckets (finished by the host per segment), spilled_pairs */
int dg_sketch_set_option(dg_ctx *, const char *name, int64_t value);
int dg_sketch_get_stat(dg_ctx *, const char *name, int64_t *value);

/* Device-resident variants for the read-sharded multi-GPU path (one rank per GPU; collectives are
 * done by the caller over RCCL on the same buffers).
 *  dg_sketch_reads_dev: bases/read_off are DEVICE pointers; writes up to cap distinct (hash,count)
 *  pairs of THIS shard, sorted by hash, into device buffers; *n_distinct on host. Synchronises. */
int dg_sketch_reads_dev(dg_ctx *, const char *bases_dev, const int64_t *read_off_dev, int64_t n_reads,
                        int64_t n_bases, int k, int w, uint64_t *hash_dev, int32_t *count_dev, int64_t cap,
                        int64_t *n_distinct);
/* counts_dev[i] += count of dict_dev[i] in the shard's (hash,count) list (both sorted); the caller
 * then all-reduces counts_dev (uint32 per dictionary minimizer). Asynchronous on the ctx stream. */
int dg_sketch_count_dictionary_dev(dg_ctx *, const uint64_t *dict_dev, int64_t n_dict,
                                   const uint64_t *hash_dev, const int32_t *count_dev, int64_t n,
                                   int32_t *counts_dev);
/* merge several sorted (hash,count) runs stored back to back (device) into one sorted distinct
 * list with summed counts (device, capacity cap). Synchronises.  Entries (0xFFFFFFFFFFFFFFFF, 0) are the padding of
 * a fixed-size exchange: a last entry with that hash and summed count 0 is dropped (a real hash has count >= 1). */
int dg_sketch_merge_runs_dev(dg_ctx *, const uint64_t *hash_dev, const int32_t *count_dev, int64_t n_total,
                             uint64_t *out_hash_dev, int32_t *out_count_dev, int64_t cap, int64_t *n_out);

/* Read-sharded scoring across ranks (semantics of solver.cpp:526-555, 711-755): the uint64 hash space is
 * cut into `world` equal ranges, rank r owns range r of the global spectrum.
 *  dg_sketch_partition_dev: split_dev[r] (r = 0..world) = first index of the sorted list hash_dev[n] owned by a rank
 *  >= r, i.e. the send offsets of the all-to-all of (hash,count) runs.  Asynchronous on the ctx stream. */
int dg_sketch_partition_dev(dg_ctx *, const uint64_t *hash_dev, int64_t n, int world, int64_t *split_dev);
/* rank1_dev[i] += base + idx + 1 for every dictionary hash that is entry idx of this rank's merged range hash_dev[n]
 * (base = number of distinct read hashes in lower ranges): after a sum over ranks rank1 - 1 is the Sp_R id
 * (solver.cpp:541-546), -1 = not a read minimizer.  Asynchronous on the ctx stream. */
int dg_sketch_rank_dictionary_dev(dg_ctx *, const uint64_t *dict_dev, int64_t n_dict, const uint64_t *hash_dev, int64_t n,
                                  int64_t base, int64_t *rank1_dev);
/* dg_sketch_count_dictionary_dev and dg_sketch_rank_dictionary_dev against the SAME list in one pass (one rank: the local
 * spectrum is the global one).  Asynchronous. */
int dg_sketch_count_rank_dictionary_dev(dg_ctx *, const uint64_t *dict_dev, int64_t n_dict, const uint64_t *hash_dev,
                                        const int32_t *count_dev, int64_t n, int64_t base, int32_t *counts_dev, int64_t *rank1_dev);
/* hist_dev[min(count, n_bins-1)] += 1 per entry: this range's share of Hist_kmer (solver.cpp:745-755).  Asynchronous. */
int dg_sketch_histogram_dev(dg_ctx *, const int32_t *count_dev, int64_t n, int n_bins, uint64_t *hist_dev);

/* ---- read-sharded scoring inside ONE process (bin/DipGenie --gpus N; SURVEY.md s8e, BASELINE configs[3]) ----
 * One host thread and one dg_ctx per rank; transport 0 = RCCL (librccl is dlopen-ed here, an in-process communicator over the
 * `devices`, one device per rank), transport 1 = the same exchange staged through host memory between the rank threads (tests on a
 * one-GPU box, where several ranks share a device).  dg_shard_ctx(s, r) is rank r's context: the caller sketches the haplotypes
 * h = r (mod N) on it from N threads of its own (index_kmers is independent per haplotype, solver.cpp:470-473) and runs the rest of
 * the pipeline on rank 0's.
 * dg_shard_score_reads replaces compute_hashes over all reads + Sp_R + kmer_count + Hist_kmer (solver.cpp:526-555, 711-755):
 * every rank sketches its contiguous block of the reads, the hit vector of the haplotype-minimizer dictionary (the sorted distinct
 * set of hap_hash[n_hap_hash], built on every device) is all-reduced over RCCL, the (hash, #reads) runs are exchanged by hash range
 * in one grouped send / receive and merged by their owners.  Out: sp_hash / sp_count (malloc-ed, dg_free) = Sp_R's keys in ascending
 * order with kmer_count; hist[n_bins] (multiplicities >= n_bins - 1 share the last bin); n_dict, dict_hits = size of the dictionary and
 * how many of its hashes some read holds; the slowest rank's sketch and exchange times (HIP events, ms).  Any of the last five may be NULL. */
typedef struct dg_shard dg_shard;
dg_shard *dg_shard_create(int n_ranks, const int *devices, int transport);   /* devices NULL: 0 .. n_ranks-1; NULL on error (dg_last_error) */
void      dg_shard_destroy(dg_shard *);
int       dg_shard_n_ranks(dg_shard *);
dg_ctx   *dg_shard_ctx(dg_shard *, int rank);
int       dg_shard_score_reads(dg_shard *, const char *bases, const int64_t *read_off, int64_t n_reads, int k, int w,
                               const uint64_t *hap_hash, int64_t n_hap_hash, uint64_t **sp_hash, int32_t **sp_count, int64_t *n_sp,
                               int64_t *hist, int n_bins, int64_t *n_dict, int64_t *dict_hits, double *ms_sketch_max, double *ms_exchange_max);

/* ---- haplotype index with vertex spans + anchor join / filter / sort ----
 * Replaces, for all haplotypes at once, Solver::index_kmers including its position -> vertex-list mapping
 * (src/solver.cpp:277-363), Solver::compute_anchors and the Anchor_hits assembly (:415-446, 560-575), the shared-anchor
 * filter (:590-638) and the occurrence sort (:641-663).  Call order: dg_anchor_begin, dg_anchor_add_haplotype for
 * h = 0 .. n_haps-1, dg_sketch_reads (any time), dg_anchor_finish.  All pointers are host memory. */
int dg_anchor_begin(dg_ctx *, int32_t n_haps, int32_t n_vertices, const int32_t *top_order_map /* [n_vertices], solver.cpp:174-199 */,
                    int k, int w);
/* seq = the haplotype's bases (node_seq concatenated along paths[h], :283-288); step_vtx[n_steps] = paths[h];
 * step_start[n_steps + 1] = base offset of every step (step_start[n_steps] = len).  *n_minimizers = |index_kmers(h)|. */
int dg_anchor_add_haplotype(dg_ctx *, int32_t h, const char *seq, int64_t len, const int32_t *step_vtx, const int64_t *step_start,
                            int64_t n_steps, int64_t *n_minimizers);
/* the same for a haplotype whose minimizer list (hash / pos = the output of dg_sketch_haplotype on its sequence, host memory) was
 * computed elsewhere: in a haplotype-sharded run every rank sketches its share of the haplotypes (src/solver.cpp:470-473 runs
 * index_kmers once per haplotype, independently) and the rank that owns the anchor stage imports them. */
int dg_anchor_add_haplotype_sketched(dg_ctx *, int32_t h, int64_t len, const uint64_t *hash, const int64_t *pos, int64_t n,
                                     const int32_t *step_vtx, const int64_t *step_start, int64_t n_steps);
typedef struct dg_anchor_result {     /* Anchor_hits flattened: occurrence i = (occ_id[i], occ_hap[i], vpool[occ_off[i] .. +occ_len[i])), */
    int64_t n_occ, n_vtx;             /* in Anchor_hits order (id asc, haplotype asc, occurrence order of :641-663)                     */
    int32_t *occ_id, *occ_hap;        /* malloc'ed by the library: dg_free each                                                         */
    uint32_t *occ_off, *occ_len;
    int32_t *vpool;
    int64_t n_candidates;             /* occurrences before the shared-anchor filter                                                    */
    int64_t n_unstable_groups;        /* (id, haplotype) groups of > 16 occurrences holding different vertex lists with equal (front,  */
} dg_anchor_result;                   /* back): their order would depend on std::sort's unstable partitioning -- redo the stage on the host */
/* sp_hash[n_sp] = sorted distinct read-minimizer hashes (Sp_R keys, the output of dg_sketch_reads); min_shared =
 * threshold * num_walks as float (:618).  Consumes the index built since dg_anchor_begin. */
int dg_anchor_finish(dg_ctx *, const uint64_t *sp_hash, int64_t n_sp, float min_shared, dg_anchor_result *out);

#ifdef __cplusplus
}
#endif
#endif
