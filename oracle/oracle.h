/* oracle/oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the DipGenie hot path (reference @ /root/reference), used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for the HIP path.
 * Nothing under dipgenie_amd/ may include, link or call this.
 *
 * Pinning: the restatement is checked (tests/test_oracle_golden.py) against
 *   - known-answer vectors produced by the reference's own functions (oracle/ref_harness.cpp linked
 *     against the reference's objects, built by oracle/Makefile into oracle/_ref/), committed under
 *     tests/golden/, and
 *   - end-to-end outputs of the unmodified reference binary (FASTA md5, DP value, r1/r2) on the
 *     reference's test inputs and on seeded synthetic graphs (tests/golden/e2e_*.json).
 *
 * Each function cites the reference file:line it follows.
 */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- sketching (src/solver.cpp:16-24, 277-412; src/MurmurHash3.cpp:255-332; src/misc.cpp:103-115) ---- */

/* MurmurHash3_x64_128(key,len,seed=0) -> h[0]^h[1]   (solver.cpp:16-24) */
uint64_t orc_hash_kmer(const char *s, int len);
void     orc_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2]);

/* Emitted-on-hash-change minimizer list of one sequence, in order (solver.cpp:302-361 / 374-409).
 * `seq` is upper-cased internally on a copy. Returns the number of emitted minimizers (may exceed
 * cap; only the first cap are written). pos[] = start of the winning k-mer (deque front). */
int64_t  orc_minimizers(const char *seq, int64_t len, int k, int w,
                        uint64_t *hash, int64_t *pos, int64_t cap);

/* std::set<uint64_t> of one read (solver.cpp:366-412): sorted distinct hashes. Returns count. */
int64_t  orc_compute_hashes(const char *read, int64_t len, int k, int w, uint64_t *out, int64_t cap);

/* Sp_R / kmer_count (solver.cpp:526-546, 711-732): sorted distinct read hashes and the number of
 * reads containing each. Outputs are malloc'ed; free with orc_free. */
int      orc_sketch_reads(const char *bases, const int64_t *read_off, int64_t n_reads, int k, int w,
                          uint64_t **hash, int32_t **n_reads_with_hash, int64_t *n_distinct);
void     orc_free(void *p);

/* ---- diploid DP (src/approximator.cpp:269-311, 362-785) ---- */

typedef struct orc_dp_graph {       /* levelized expanded graph; vertex ids are level-sorted */
    int32_t n_vertices, n_levels, R;
    const int32_t *level_off;       /* [n_levels+1] */
    const int64_t *out_off;         /* [n_vertices+1], adjacency order preserved */
    const int32_t *out_dst;
    const uint8_t *out_w;           /* 0/1 */
    const int64_t *hom_off, *het_off;   /* [n_vertices+1] sorted-unique colour CSR */
    const int32_t *hom_col, *het_col;
} orc_dp_graph;

typedef struct orc_dp_result {
    int32_t value, s_het, n_p1, n_p2;
    int32_t *p1_from, *p1_to, *p2_from, *p2_to;   /* caller-provided, capacity >= R+2 (+ slack) */
    int32_t cap;
    uint64_t cells, relaxations;
} orc_dp_result;

/* Literal single-thread restatement of the level loop approximator.cpp:532-716 (scatter form,
 * loop order r,i,j, adjacency order, take-if rule :657-659) and of the sink read-out :774-785.
 * If level_digest != NULL it receives one uint64 per level l=1..L-1 (index l): a digest of
 * dp_cur after the roll, = sum over reachable cells (cell_index t, r-major) of
 *   (value+1)*(t+1) + 0x9E3779B97F4A7C15 * ((pred_i << 15 | pred_j) + 1)*(t+1)   mod 2^64,
 * i.e. it pins the value AND the winning predecessor (tie-break :657-659) of every cell. */
int      orc_dp_solve_diploid(const orc_dp_graph *g, orc_dp_result *res, uint64_t *level_digest);

/* ---- haploid DP (src/approximator.cpp:44-72) ---- */
typedef struct orc_hap_graph {      /* expanded graph after topologically_reorder: every edge u -> v has u < v */
    int32_t n_vertices, R;
    const int64_t *out_off;         /* [n_vertices+1], adjacency order preserved */
    const int32_t *out_dst;
    const uint8_t *out_w;           /* 0/1 */
    const int32_t *n_colours;       /* |color[v]| */
} orc_hap_graph;
/* Literal restatement of the scatter loop :47-72: dp, back_vtx, back_r as [v * (R+1) + r]; every state starts at 0 with
 * back pointers -1 (:50-52), updates are strict (>), so the FIRST candidate in (u asc, r asc, adjacency order) wins. */
int      orc_dp_haploid(const orc_hap_graph *g, int32_t *dp, int32_t *back_vtx, int32_t *back_r);

/* inter_size_union2x2 / symdiff_size_union2x2 (approximator.cpp:269-311) on sorted int lists */
int      orc_inter_union2x2(const int32_t *A, int na, const int32_t *B, int nb,
                            const int32_t *C, int nc, const int32_t *D, int nd);
int      orc_symdiff_union2x2(const int32_t *A, int na, const int32_t *B, int nb,
                              const int32_t *C, int nc, const int32_t *D, int nd);

#ifdef __cplusplus
}
#endif
