#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): proves the drop-in boundary with the REFERENCE's own host code.

    python3 oracle/ref_hip_patch.py            ->  oracle/_ref/DipGenie_ref_hip

Copies /root/reference/src to a temporary directory OUTSIDE the repository, applies the patch of INTEGRATION.md s2.1-2.3 to the copy
(the three insertions below are OUR code -- the text a DipGenie maintainer would add; the insertion points are found by pattern, no
reference text is kept in this repository), compiles every reference source with oracle/Makefile's REFFLAGS and links the result
against dipgenie_amd/csrc/libdipgenie_hip.so.  Only the binary lands in oracle/_ref/ (git-ignored; it travels to the GPU box like
DipGenie_ref does); the patched sources are deleted with the temporary directory.  tests/test_gpu_ref_hip.py runs the binary on
the MI355X and requires the reference's golden FASTA md5s.  Nothing under dipgenie_amd/ or in bench.py's timed region touches it.

What the patched reference does differently from the unmodified one -- and nothing else:
  s2.1  Approximator::diploid_dp_approximation_solver: the OpenMP level loop (approximator.cpp:532-716) is compiled out; in its place
        the levelized graph is flattened into a dg_dp_graph and dg_dp_solve_diploid runs; the sink entry the rest of the function
        reads (:774-785, :934-935: value, s_het, the two edge chains) is rebuilt from the dg_dp_result, so every line after the
        loop stays as it is.
  s2.2  Solver::compute_and_classify_anchors: the per-read compute_hashes calls (solver.cpp:528-532) are replaced by ONE
        dg_sketch_reads call.  Read_hashes is only used to build Sp_R (:533-546) and kmer_count (:711-732), which need the distinct
        hashes and the number of reads holding each: a hash held by c reads is put into the sets of reads 0 .. c-1.
  s2.3  Solver::index_kmers: the window loop (:302-361) is replaced by dg_sketch_haplotype (one call per haplotype, serialised with
        an omp critical section: a dg_ctx is not thread-safe and index_kmers runs inside an OpenMP loop, :470-473); the anchors are
        built from (hash, position) exactly as :337-358 builds them.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFSRC = "/root/reference/src"
OUT = os.path.join(ROOT, "oracle", "_ref", "DipGenie_ref_hip")
REFFLAGS = ["-O3", "-std=c++17", "-fopenmp", "-pthread", "-march=x86-64-v3", "-w"]          # oracle/Makefile: REFFLAGS
REFLIBS = ["-lm", "-lz", "-lpthread", "-ldl"]                                                  # oracle/Makefile: REFLIBS
REFOBJS = ["main", "gfa-io", "gfa-base", "options", "kalloc", "sys", "approximator", "MurmurHash3", "misc", "solver"]

CTX_DEF = r'''
/* --- dipgenie_hip (INTEGRATION.md s2): one device context for the process --- */
#include "dipgenie_hip.h"
dg_ctx *dg_ref_ctx() {
    static dg_ctx *ctx = dg_create(0);
    if (!ctx) { fprintf(stderr, "[dipgenie_hip] %s\n", dg_last_error()); exit(1); }            /* no gfx950 device: no CPU fallback */
    return ctx;
}
'''

# s2.2: inserted in front of the OpenMP loop that calls compute_hashes per read (solver.cpp:528)
SKETCH_READS = r'''
    {   /* --- dipgenie_hip (INTEGRATION.md s2.2): every read in ONE device call --- */
        std::string dg_bases; std::vector<int64_t> dg_off(num_reads + 1, 0);
        for (int32_t r = 0; r < num_reads; r++) { dg_bases += ip_reads[r].second; dg_off[r + 1] = (int64_t)dg_bases.size(); }
        uint64_t *dg_h = nullptr; int32_t *dg_c = nullptr; int64_t dg_n = 0;
        if (dg_sketch_reads(dg_ref_ctx(), dg_bases.data(), dg_off.data(), num_reads, k_mer, window, &dg_h, &dg_c, &dg_n) != DG_OK) {
            fprintf(stderr, "[dipgenie_hip] %s\n", dg_last_error()); exit(1);
        }
        /* Read_hashes only feeds Sp_R (distinct hashes) and kmer_count (#reads holding a hash): a hash held by c reads goes into the sets of reads 0 .. c-1 */
        for (int64_t i = 0; i < dg_n; i++) for (int32_t r = 0; r < dg_c[i]; r++) Read_hashes[r].insert(dg_h[i]);
        dg_free(dg_h); dg_free(dg_c);
    }
'''

# s2.3: inserted in front of the window loop of index_kmers (solver.cpp:302); returns, so the loop below it never runs
SKETCH_HAP = r'''
    {   /* --- dipgenie_hip (INTEGRATION.md s2.3): the window loop below, on the device --- */
        uint64_t *dg_h = nullptr; int64_t *dg_pos = nullptr; int64_t dg_n = 0; int dg_rc;
        #pragma omp critical(dg_hip_ctx)
        dg_rc = dg_sketch_haplotype(dg_ref_ctx(), haplotype.data(), (int64_t)haplotype.size(), k_mer, window, &dg_h, &dg_pos, &dg_n);
        if (dg_rc != DG_OK) { fprintf(stderr, "[dipgenie_hip] %s\n", dg_last_error()); exit(1); }
        for (int64_t m = 0; m < dg_n; m++) {                                  /* as :337-358 with best_hash = dg_h[m], best_start_idx = dg_pos[m] */
            Anchor anchor; anchor.h = hap;
            std::unordered_set<int32_t> unique_vtx_set; std::vector<int32_t> unique_vtxs_vec;
            for (int64_t j = dg_pos[m]; j < dg_pos[m] + k_mer; j++) {
                int32_t vtx_idx = idx_vtx_map[j];
                if (unique_vtx_set.insert(vtx_idx).second) unique_vtxs_vec.push_back(vtx_idx);
            }
            std::sort(unique_vtxs_vec.begin(), unique_vtxs_vec.end(), [&](int32_t a, int32_t b) { return top_order_map[a] < top_order_map[b]; });
            anchor.k_mers = std::move(unique_vtxs_vec);
            kmer_index.emplace_back(dg_h[m], anchor);
        }
        dg_free(dg_h); dg_free(dg_pos);
        return kmer_index;
    }
'''

# s2.1: inserted behind the (compiled-out) level loop, in front of `for (auto& lk : locks) omp_destroy_lock(&lk);` (approximator.cpp:718)
DP_CALL = r'''
    {   /* --- dipgenie_hip (INTEGRATION.md s2.1): the level loop :532-716 on the device --- */
        /* flatten (vertex ids are already level-sorted by strict_bfs_levelize_and_reorder, ExpandedGraph.hpp:360-407) */
        const int dg_nV = (int)g.adj_list.size();
        std::vector<int32_t> dg_level_off(L + 1, 0), dg_out_dst, dg_hom_col, dg_het_col;
        std::vector<int64_t> dg_out_off(dg_nV + 1, 0), dg_hom_off(dg_nV + 1, 0), dg_het_off(dg_nV + 1, 0);
        std::vector<uint8_t> dg_out_w;
        for (int l = 0; l < L; ++l) dg_level_off[l + 1] = dg_level_off[l] + (int)g.vertices_in_level[l].size();
        for (int v = 0; v < dg_nV; ++v) {
            for (auto &[to, w] : g.adj_list[v]) { dg_out_dst.push_back(to); dg_out_w.push_back((uint8_t)w); }
            dg_out_off[v + 1] = (int64_t)dg_out_dst.size();
            dg_hom_col.insert(dg_hom_col.end(), homo_sorted[v].begin(), homo_sorted[v].end());   dg_hom_off[v + 1] = (int64_t)dg_hom_col.size();
            dg_het_col.insert(dg_het_col.end(), hetero_sorted[v].begin(), hetero_sorted[v].end()); dg_het_off[v + 1] = (int64_t)dg_het_col.size();
        }
        dg_hom_col.push_back(0); dg_het_col.push_back(0);                     /* (never-NULL data pointers for colourless graphs) */
        dg_dp_graph dg_G{dg_nV, L, R, dg_level_off.data(), dg_out_off.data(), dg_out_dst.data(), dg_out_w.data(),
                         dg_hom_off.data(), dg_het_off.data(), dg_hom_col.data(), dg_het_col.data()};
        std::vector<int32_t> dg_p1f(R + 8), dg_p1t(R + 8), dg_p2f(R + 8), dg_p2t(R + 8);
        dg_dp_result dg_res{};
        dg_res.p1_from = dg_p1f.data(); dg_res.p1_to = dg_p1t.data(); dg_res.p2_from = dg_p2f.data(); dg_res.p2_to = dg_p2t.data(); dg_res.cap = R + 8;
        if (dg_dp_solve_diploid(dg_ref_ctx(), &dg_G, &dg_res) != DG_OK) { std::cout << dg_last_error() << std::endl; exit(1); }   /* reference error style (:799) */
        /* the sink entry that :774-785 and :934-935 read: dp_cur[(R * k_sink + 0) * k_sink + 0] with k_sink = 1 */
        static std::deque<EdgeNode> dg_nodes;
        dp_cur.assign((std::size_t)(R + 1), dp_entry(0, 0));
        dp_entry &dg_sink = dp_cur[(std::size_t)R];
        dg_sink.value = dg_res.value; dg_sink.s_het = dg_res.s_het;
        for (int i = 0; i < dg_res.n_p1; ++i) { dg_nodes.emplace_back(dg_p1f[i], dg_p1t[i], dg_sink.p1_tail); dg_sink.p1_tail = &dg_nodes.back(); ++dg_sink.p1_count; }
        for (int i = 0; i < dg_res.n_p2; ++i) { dg_nodes.emplace_back(dg_p2f[i], dg_p2t[i], dg_sink.p2_tail); dg_sink.p2_tail = &dg_nodes.back(); ++dg_sink.p2_count; }
    }
'''


def one(pattern, txt, what):
    m = list(re.finditer(pattern, txt, flags=re.M))
    assert len(m) == 1, f"{what}: pattern found {len(m)} times"
    return m[0]


def patch(src):
    # ---- solver.cpp: s2.2 + s2.3 + the context ----
    p = os.path.join(src, "solver.cpp")
    t = open(p).read()
    m = one(r"^[ \t]*Read_hashes\[r\] = compute_hashes\(ip_reads\[r\]\.second\);[ \t]*\n", t, "compute_hashes call (solver.cpp:531)")
    t = t[:m.start()] + "        (void)r;   /* dipgenie_hip: Read_hashes was filled by the one dg_sketch_reads call above */\n" + t[m.end():]
    m = one(r"^[ \t]*std::map<uint64_t, int32_t> Sp_R;[ \t]*\n", t, "declaration of Sp_R (solver.cpp:527)")
    t = t[:m.end()] + SKETCH_READS + t[m.end():]
    f0 = one(r"^std::vector<std::pair<uint64_t, Anchor>> Solver::index_kmers\(int32_t hap\)", t, "Solver::index_kmers (solver.cpp:277)").start()
    m = list(re.finditer(r"^[ \t]*uint64_t prev_hash = UINT64_MAX;[^\n]*\n", t[f0:], flags=re.M))
    assert m, "start of index_kmers' window loop (solver.cpp:302) not found"
    at = f0 + m[0].end()                                                      # the first one behind the function header
    t = t[:at] + SKETCH_HAP + t[at:]
    m = one(r'^#include "solver\.h"[ \t]*\n', t, "first include of solver.cpp")
    t = t[:m.end()] + "#include <cstdio>\n#include <cstdlib>\n#include <unordered_set>\n" + CTX_DEF + t[m.end():]
    open(p, "w").write(t)
    # ---- approximator.cpp: s2.1 ----
    p = os.path.join(src, "approximator.cpp")
    t = open(p).read()
    m = one(r"^[ \t]*#pragma omp parallel num_threads\(num_threads\)[ \t]*\n(?=[ \t]*\{[ \t]*\n[ \t]*const std::size_t sz0 = \(std::size_t\)\(R \+ 1\);)", t, "OpenMP region of the diploid level loop (approximator.cpp:532)")
    t = t[:m.start()] + "#if 0   /* dipgenie_hip: the level loop runs on the device (below) */\n" + t[m.start():]
    m = one(r"^[ \t]*for \(auto& lk : locks\) omp_destroy_lock\(&lk\);[ \t]*\n", t, "end of the level loop (approximator.cpp:718)")
    t = t[:m.start()] + "#endif\n" + DP_CALL + t[m.start():]
    m = one(r'^#include "approximator\.h"[ \t]*\n', t, "first include of approximator.cpp")
    t = t[:m.end()] + '#include <deque>\n#include "dipgenie_hip.h"\ndg_ctx *dg_ref_ctx();\n' + t[m.end():]
    open(p, "w").write(t)


def main():
    if not os.path.isdir(REFSRC):
        print(f"[oracle] {REFSRC} not present; using the prebuilt oracle/_ref/DipGenie_ref_hip if any")
        return 0
    lib = os.path.join(ROOT, "dipgenie_amd", "csrc")
    assert os.path.exists(os.path.join(lib, "libdipgenie_hip.so")), "build dipgenie_amd/csrc first"
    newest = max(os.path.getmtime(os.path.join(REFSRC, f)) for f in os.listdir(REFSRC))
    if os.path.exists(OUT) and os.path.getmtime(OUT) > max(newest, os.path.getmtime(__file__), os.path.getmtime(os.path.join(ROOT, "include", "dipgenie_hip.h"))):
        return 0                                                              # up to date (the library is bound at run time)
    with tempfile.TemporaryDirectory(prefix="dg_ref_hip_") as td:             # outside the repository: reference sources never enter it
        src = os.path.join(td, "src")
        shutil.copytree(REFSRC, src)
        patch(src)
        objs = []
        procs = []
        for o in REFOBJS:
            obj = os.path.join(td, o + ".o")
            objs.append(obj)
            procs.append(subprocess.Popen(["g++", *REFFLAGS, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(src, o + ".cpp"), "-o", obj]))
        assert all(p.wait() == 0 for p in procs), "compilation of the patched reference failed"
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", *REFFLAGS, *objs, "-o", OUT, "-L" + lib, "-ldipgenie_hip", "-Wl,-rpath,$ORIGIN/../../dipgenie_amd/csrc", *REFLIBS])
    print(f"[oracle] built {os.path.relpath(OUT, ROOT)} (reference host code + INTEGRATION.md s2.1-2.3, device loops through libdipgenie_hip.so)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
