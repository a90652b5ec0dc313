// Host-side mirror of the reference's Solver / Approximator objects for the diploid hot path.
//
// Everything that is NOT one of the two device loops stays here, restated so that vertex numbering,
// adjacency order and every tie-break match the reference exactly:
//   Solver::read_gfa                       /root/reference/src/solver.cpp:27-227
//   Solver::compute_and_classify_anchors   /root/reference/src/solver.cpp:449-887
//   Approximator::solve                    /root/reference/src/approximator.cpp:1014-1331
//   ExpandedGraph::topologically_reorder   /root/reference/src/ExpandedGraph.hpp:29-102
//   ExpandedGraph::strict_bfs_levelize_and_reorder   /root/reference/src/ExpandedGraph.hpp:269-409
//   haploid dp_approximation_solver        /root/reference/src/approximator.cpp:44-168 (CPU by design)
//   traceback -> sequences, certificate    /root/reference/src/approximator.cpp:720-1004
// The two device loops are reached only through the Backend table (= include/dipgenie_hip.h).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <initializer_list>
#include <utility>
#include <vector>

#include "../../include/dipgenie_hip.h"
#include "fitter.hpp"
#include "gfa_reader.hpp"

namespace dg {

// std::vector that leaves trivially-constructible elements uninitialised on resize: the graph arrays hold tens of millions of
// entries that are all written (in parallel) right after the allocation -- a serial zero fill first costs as much as the fill
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T> using uvec = std::vector<T, NoInitAlloc<T>>;

struct Backend {     // same signatures as the C ABI, plus an opaque ctx
    void *ctx = nullptr;
    int (*sketch_reads)(void *, const char *, const int64_t *, int64_t, int, int, uint64_t **, int32_t **, int64_t *) = nullptr;
    int (*sketch_haplotype)(void *, const char *, int64_t, int, int, uint64_t **, int64_t **, int64_t *) = nullptr;
    int (*dp_solve_diploid)(void *, const dg_dp_graph *, dg_dp_result *) = nullptr;
    int (*dp_solve_haploid)(void *, const dg_hap_graph *, int32_t *dp, int32_t *back_vtx, int32_t *back_r) = nullptr;   // optional
    void (*free_buf)(void *) = nullptr;
    // optional: haplotype index with vertex spans + anchor join / filter / sort behind the boundary
    int (*anchor_begin)(void *, int32_t n_haps, int32_t n_vertices, const int32_t *top_order_map, int k, int w) = nullptr;
    int (*anchor_add_haplotype)(void *, int32_t h, const char *seq, int64_t len, const int32_t *step_vtx, const int64_t *step_start, int64_t n_steps,
                                int64_t *n_minimizers) = nullptr;
    int (*anchor_finish)(void *, const uint64_t *sp_hash, int64_t n_sp, float min_shared, dg_anchor_result *out) = nullptr;
    // optional (sharded runs): a haplotype whose minimizer list was sketched by another rank
    int (*anchor_add_haplotype_sketched)(void *, int32_t h, int64_t len, const uint64_t *hash, const int64_t *pos, int64_t n, const int32_t *step_vtx,
                                         const int64_t *step_start, int64_t n_steps) = nullptr;
    void (*hint_dp_soon)(void *, int64_t est_cells) = nullptr;   // optional, may be called repeatedly (latest wins): the DP will run later with about est_cells cells
    const char *(*last_error)() = nullptr;
};

struct Options {
    int threads = 4;          // -t
    int ploidy = 2;           // -p
    int R = 18;               // -R
    int k = 31, w = 25;       // -k -w
    float threshold = 1.0f;   // -T
    bool debug = false;       // -d
    bool quiet = false;       // (ours) suppress progress chatter
    std::string gfa_file, reads_file, hap_file;   // -g -r -o
    std::string dump_prefix;  // (ours) if set, dump the levelized DP graph to <prefix>.dpg
    bool dump_only = false;   // (ours, tests) stop after the dump
    std::string anchor_dump;  // (ours, tests) if set, write Anchor_hits + homo_bv as text (format of oracle/ref_harness.cpp `anchors`)
    int haploid_mode = 0;        // (ours) haploid (vertex, r) tables: 0 = by graph shape, 1 = host gather loop, 2 = device (if the backend offers it)
    bool leak_at_exit = false;   // (ours) the process exits right after run(): skip the teardown of the big graph objects
    bool host_anchors = false;   // (ours, tests) keep the anchor join / filter / sort on the host even if the backend offers it
};

// ExpandedGraph.hpp:16-26, flattened: CSR adjacency (per-vertex order = the reference's push order),
// CSR colours, and original-vertex lists as (offset, length) into one pool (dummies share their
// parent's list instead of copying it, ExpandedGraph.hpp:330).
struct ExpandedGraph {
    int32_t n = 0;
    uvec<int64_t> adj_off;                      // [n+1]
    uvec<int32_t> adj_dst;
    uvec<uint8_t> adj_w;
    uvec<int32_t> haplotype, level;             // (level: literal route only; the vertex ids are level-sorted, see level_of)
    uvec<uint32_t> orig_off, orig_len;
    uvec<int32_t> orig_pool;
    uvec<int64_t> col_off;                      // [n+1]  (literal route; the fused route writes the HOM / HET split directly)
    uvec<int32_t> col_pool;
    uvec<int32_t> level_off;                    // [L+1] after levelize (vertex ids are level-sorted)
    bool colours_split = false;                 // hom_* / het_* below are filled (Pipeline::build_levelized_fast)
    uvec<int64_t> hom_off, het_off;             // [n+1] sorted-unique HOM / HET colour CSR (approximator.cpp:431-453)
    uvec<int32_t> hom_col, het_col;
    int64_t deg(int v) const { return adj_off[v + 1] - adj_off[v]; }
    int64_t ncol(int v) const { return col_off[v + 1] - col_off[v]; }
    void topologically_reorder(int sink);
    int strict_bfs_levelize_and_reorder();
  private:
    void permute(const uvec<int32_t> &order);   // new vertex i = old vertex order[i]
};

// colour list of an anchor record: nearly always one entry (its own colour), a few more after containment
// propagation -- two inline slots, heap only beyond (millions of records at MHC scale)
class ColourList {
    int inl_[2] = {0, 0};
    uint32_t n_ = 0;
    std::vector<int> *more_ = nullptr;
    void spill() { more_ = new std::vector<int>(inl_, inl_ + n_); }
public:
    ColourList() = default;
    ColourList(std::initializer_list<int> il) { for (int c : il) push_back(c); }
    ColourList(const ColourList &o) : n_(o.n_), more_(o.more_ ? new std::vector<int>(*o.more_) : nullptr) { inl_[0] = o.inl_[0]; inl_[1] = o.inl_[1]; }
    ColourList(ColourList &&o) noexcept : n_(o.n_), more_(o.more_) { inl_[0] = o.inl_[0]; inl_[1] = o.inl_[1]; o.more_ = nullptr; o.n_ = 0; }
    ColourList &operator=(ColourList o) noexcept { std::swap(inl_[0], o.inl_[0]); std::swap(inl_[1], o.inl_[1]); std::swap(n_, o.n_); std::swap(more_, o.more_); return *this; }
    ~ColourList() { delete more_; }
    void push_back(int c) {
        if (!more_ && n_ < 2) { inl_[n_++] = c; return; }
        if (!more_) spill();
        more_->push_back(c); ++n_;
    }
    const int *begin() const { return more_ ? more_->data() : inl_; }
    const int *end() const { return begin() + n_; }
    size_t size() const { return n_; }
};

struct AnchorRec {            // approximator.h:11-18
    int startOrg, endOrg, startExp, endExp;
    ColourList colours;
    int nodeID;
};

// Flattened levelized graph in dg_dp_graph layout (owning storage).
struct DpGraphStorage {
    uvec<int32_t> level_off, out_dst, hom_col, het_col;
    uvec<int64_t> out_off, hom_off, het_off;
    uvec<uint8_t> out_w;
    dg_dp_graph view(int R) const;
    bool save(const std::string &path, int R) const;   // little-endian binary, see pipeline.cpp
    bool load(const std::string &path, int &R);
};

struct Summary {              // what tests and the CLI report
    int32_t dp_value = 0, s_het = 0, r1 = -1, r2 = -1, obj = 0, best_r_haploid = -1;
    int64_t len1 = 0, len2 = 0;
    int64_t spectrum = 0, n_levels = 0, n_vertices = 0, n_colours = 0;
    uint64_t cells = 0, relaxations = 0;
    std::vector<int64_t> minimizers_per_hap, anchors_per_hap;
    KGFitResult fit;
    std::vector<std::pair<std::string, double>> stage_s;
};

// One anchor occurrence: vertex list vpool[off, off+len) of read-minimizer id `a` on haplotype `h`.
struct Occ { int32_t a, h; uint32_t off, len; };

class Pipeline {
  public:
    Options opt;
    Backend be;
    Summary sum;

    // ---- Solver state (solver.h:73-85) ----
    uint32_t n_vtx = 0, num_walks = 0;
    std::vector<std::vector<uint32_t>> adj_list;
    std::vector<std::string> node_seq;
    std::vector<std::vector<uint32_t>> paths;
    std::vector<int32_t> top_order_map;
    std::vector<std::string> hap_id2name;
    std::vector<std::pair<std::string, std::string>> reads;
    int32_t count_sp_r = 0;
    // Anchor_hits[a][h] flattened: occs sorted by (a, h, final occurrence order)
    std::vector<Occ> occs;
    std::vector<int32_t> vpool;
    std::vector<uint8_t> homo_bv;          // per read-minimizer id

    // ---- sharded runs (dipgenie_amd/run_sharded.py, BASELINE configs[3]): what other ranks computed, handed in before run_loaded() ----
    // Sp_R keys in ascending order with kmer_count (solver.cpp:533-555, 711-732) = ShardedSketch.gather_spectrum; optionally the
    // multiplicity histogram Hist_kmer (:745-755) the ranks all-reduced, checked against the one derived from the counts
    bool spectrum_injected = false;
    std::vector<uint64_t> inj_sp_hash;
    std::vector<int32_t> inj_sp_count;
    std::vector<int64_t> inj_hist;
    // index_kmers' window loop per haplotype (hash, position of the winning k-mer: dg_sketch_haplotype), sketched by the owner rank
    struct HapSketch { std::vector<uint64_t> hash; std::vector<int64_t> pos; bool set = false; };
    std::vector<HapSketch> inj_hap;
    std::string haplotype_sequence(uint32_t h) const;   // node_seq concatenated along paths[h] (solver.cpp:283-288)

    int load_graph(std::string &err);      // gfa_read + Solver::read_gfa
    int run_loaded(std::string &err);      // everything after load_graph (main.cpp:163-165)
    int load_reads(std::string &err);      // Solver::read_ip_reads
    int compute_and_classify_anchors(std::string &err);
    int solve(std::string &err);           // Approximator::solve (writes the FASTA)
    int run(std::string &err);             // main.cpp:117-165
    bool dump_anchors(const std::string &path) const;

    // exposed for tests
    DpGraphStorage dpg;

  private:
    void read_gfa_from(const GfaGraph &g);
    std::vector<int> haploid_dp(const ExpandedGraph &g, int R, std::string &err);
    int diploid(ExpandedGraph &g, const std::vector<uint8_t> &color_homo_bv,
                const std::vector<std::vector<AnchorRec>> &anchorsByHap, std::string &err);
    // fused + threaded route from Anchor_hits to the levelized graph (fast_graph.cpp); false: take the literal route
    bool build_levelized_fast(ExpandedGraph &g, std::vector<std::vector<AnchorRec>> &anchorsByHap, std::vector<uint8_t> &color_homo_bv);
    void stamp(const char *name, double t0);
    double t_run0 = 0;
    bool reads_loaded = false;
    // fit + classify on its own thread (compute_and_classify_anchors starts it, wait_fit joins it)
    std::thread fit_thread;
    bool fit_pending = false;
    double fit_t0 = 0;
    int64_t fit_n_hom = 0;
    std::vector<int32_t> fit_sp_count;
    void wait_fit();
  public:
    ~Pipeline() { if (fit_thread.joinable()) fit_thread.join(); }
  private:
};

double now_s();

}  // namespace dg
