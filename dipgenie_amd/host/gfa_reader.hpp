// Minimal GFA 1.1 (S/L/W, optionally gzip) reader for the DipGenie hot path.
//
// Fresh implementation of the *observable* behaviour of the reference's gfatools-derived parser as
// consumed by Solver::read_gfa (/root/reference/src/solver.cpp:27-125):
//   - segment id = order of first appearance on an S- or L-line  (gfa-base.cpp:75 gfa_add_seg,
//     called from gfa-io.cpp:248 and :333-334)
//   - W-lines resolve names with the segments known so far; unknown names are skipped
//     (gfa-io.cpp:399-405)
//   - walks are flipped to the majority strand (gfa-io.cpp:64-93 gfa_walk_flip)
//   - arcs are symmetrised: every link v->w implies w^1->v^1 (gfa-base.cpp:270-305
//     gfa_fix_symm_add); an explicit complementary L-line is not duplicated
//   - segments without sequence/length are dropped together with their arcs
//     (gfa-base.cpp:202-214, 216-234)
// Only what read_gfa consumes is produced: per-segment sequence, the multiset of forward-strand
// successors of every segment, and the walks.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace dg {

struct GfaWalk {
    std::string sample;
    int32_t hap = 0;
    std::vector<uint32_t> v;   // oriented vertex = seg<<1 | is_reverse
};

struct GfaGraph {
    std::vector<std::string> seg_name;
    std::vector<std::string> seg_seq;
    std::vector<uint32_t> seg_len;
    std::vector<uint8_t> seg_del;
    // successors (oriented vertices) of every oriented vertex, after symmetrisation, arbitrary order
    std::vector<std::vector<uint32_t>> arcs;   // size 2*n_seg
    std::vector<GfaWalk> walks;
    uint32_t n_seg() const { return (uint32_t)seg_name.size(); }
};

// Returns false (and sets err) if the file cannot be opened.
bool read_gfa_file(const std::string &path, GfaGraph &g, std::string &err);

}  // namespace dg
