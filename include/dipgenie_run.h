/* include/dipgenie_run.h -- C ABI of libdipgenie_run.so: the host pipeline (mirror of Solver / Approximator for the diploid hot
 * path) as a library, for callers that bring part of the work themselves.
 *
 * It exists for BASELINE configs[3] -- "minimizer index+scoring sharded over 8 x MI355X ..., DP on GPU0" -- as ONE run
 * (dipgenie_amd/run_sharded.py): every rank opens the GFA, sketches its share of the haplotypes and scores its shard of the reads;
 * the rank that owns the DP injects what the others computed and runs the rest, exactly the reference's stage order:
 *
 *   dgr_open                      gfa_read + Solver::read_gfa                         src/main.cpp:117-133, src/solver.cpp:27-227
 *   dgr_haplotype_sequence        the string index_kmers sketches for haplotype h     src/solver.cpp:283-288
 *   dgr_inject_haplotype_sketch   index_kmers' window loop for h, done by its owner   src/solver.cpp:302-361 (= dg_sketch_haplotype)
 *   dgr_inject_spectrum           Sp_R keys + kmer_count (+ Hist_kmer) of ALL reads    src/solver.cpp:526-555, 711-755
 *                                 (= ShardedSketch.gather_spectrum of dipgenie_amd/dist_sketch.py)
 *   dgr_solve                     compute_and_classify_anchors from the join on, Approximator::solve, FASTA
 *                                                                                     src/solver.cpp:560-887, src/approximator.cpp:1014-1331
 * Plain pointers and sizes; int return (0 ok, <0 error, message from dgr_last_error()).  One handle <-> one run; not thread-safe.
 * The device loops go through libdipgenie_hip.so on `device`; there is no CPU fallback (tests link the same source against the
 * oracle: tests/harness/libdg_run_oracle.so). */
#ifndef DIPGENIE_RUN_H
#define DIPGENIE_RUN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dgr_handle dgr_handle;

typedef struct dgr_options {          /* the reference's command line (src/main.cpp:39-111) */
    const char *gfa_file, *reads_file, *out_file;    /* -g -r -o ; reads_file may be NULL when a spectrum is injected */
    int32_t threads, ploidy, R, k, w;                /* -t -p -R -k -w */
    float threshold;                                 /* -T */
    int32_t device;                                  /* HIP device ordinal of the anchor + DP stages */
    int32_t quiet;                                   /* suppress the reference's progress chatter */
} dgr_options;

typedef struct dgr_summary {
    int32_t dp_value, s_het, r1, r2, obj;
    int64_t len1, len2, spectrum, n_levels, n_vertices;
    uint64_t cells, relaxations;
    double seconds;                                  /* dgr_solve wall time */
} dgr_summary;

dgr_handle *dgr_open(const dgr_options *);           /* NULL on failure */
void        dgr_close(dgr_handle *);
const char *dgr_last_error(void);
int32_t     dgr_n_haplotypes(dgr_handle *);
/* *seq points into the handle (valid until the next call for another haplotype or dgr_close) */
int dgr_haplotype_sequence(dgr_handle *, int32_t h, const char **seq, int64_t *len);
/* Solver::read_ip_reads (src/solver.cpp:230-245) on options.reads_file: all reads concatenated + offsets [n + 1]; the arrays live in
 * the handle.  A sharded run hands every rank its contiguous block of them. */
int dgr_load_reads(dgr_handle *, int64_t *n_reads, const char **bases, const int64_t **read_off);
int dgr_inject_haplotype_sketch(dgr_handle *, int32_t h, const uint64_t *hash, const int64_t *pos, int64_t n);
/* hist may be NULL; otherwise hist[min(count, n_bins - 1)] = #distinct hashes with that count */
int dgr_inject_spectrum(dgr_handle *, const uint64_t *sp_hash, const int32_t *sp_count, int64_t n, const int64_t *hist, int32_t n_bins);
int dgr_solve(dgr_handle *, dgr_summary *out);

#ifdef __cplusplus
}
#endif
#endif
