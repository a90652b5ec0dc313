#!/usr/bin/env python3
"""Pins the oracle to the REFERENCE per level ("per-level dp_cur values, sink edge chains").  Build container only.

What it does:
  1. copies /root/reference/src to a temp dir (never writes to /root/reference);
  2. inserts OUR OWN instrumentation into the copy of approximator.cpp -- no reference text is kept in this repository: the two
     insertion points are found by pattern (the roll `dp_cur.swap(dp_next);` at approximator.cpp:706, and the materialisation of
     the second edge list at :782), the inserted code is below:
       * after the roll, inside the `omp single`: the digest of dp_cur with the formula of oracle/oracle.h (sum over reachable
         cells t, r-major, of (value + 1)(t + 1) + 0x9E3779B97F4A7C15 ((pred_i << 15 | pred_j) + 1)(t + 1) mod 2^64), one line per
         level on the file named by DG_REF_DIGEST_OUT;
       * after the sink read-out: the two weighted-edge lists (:757-764, :781-782);
  3. builds it with oracle/Makefile's rules for the reference (same flags: REFFLAGS / REFLIBS), into the temp dir;
  4. runs the toy, bub_*, GFA-corner, c5s and MHC_4 diploid cases of tests/golden/e2e.json and writes tests/golden/level_digests.json
     (every digest verbatim for graphs up to 4,000 levels; for larger ones count + sha256 of the little-endian uint64 array + the
     first and last 8 digests).
tests/test_level_digests.py then requires the oracle's digests and edge lists, computed on the levelized graph OUR host pipeline
builds from the same files, to equal these: the GPU parity tests (HIP == oracle on every digest) then stand on the reference's own
cells, not only on its end results."""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFSRC = "/root/reference/src"

DIGEST_CODE = r'''
                /* --- dipgenie_amd instrumentation (tests/golden/make_level_digests.py): digest of dp_cur after the roll --- */
                if (const char *dg_out = getenv("DG_REF_DIGEST_OUT")) {
                    static FILE *dg_f = nullptr;
                    if (!dg_f) dg_f = fopen(dg_out, "w");
                    unsigned long long dg_d = 0;
                    for (std::size_t dg_t = 0; dg_t < dp_cur.size(); ++dg_t)
                        if (dp_cur[dg_t].value != NEG_INF)
                            dg_d += (unsigned long long)(unsigned int)(dp_cur[dg_t].value + 1) * (unsigned long long)(dg_t + 1) +
                                    0x9E3779B97F4A7C15ULL * ((((unsigned long long)dp_cur[dg_t].pred_i << 15) | (unsigned long long)dp_cur[dg_t].pred_j) + 1ULL) *
                                        (unsigned long long)(dg_t + 1);
                    fprintf(dg_f, "L %d %016llx\n", l + 1, dg_d);
                    fflush(dg_f);
                }
'''
EDGES_CODE = r'''
        /* --- dipgenie_amd instrumentation: the sink's two weighted-edge lists --- */
        if (const char *dg_out = getenv("DG_REF_DIGEST_OUT")) {
            FILE *dg_f = fopen((std::string(dg_out) + ".edges").c_str(), "w");
            fprintf(dg_f, "V %d %d\n", sink_dp_final.value, sink_dp_final.s_het);
            for (auto &e : weighted_p1_edges) fprintf(dg_f, "P1 %d %d\n", e.first, e.second);
            for (auto &e : weighted_p2_edges) fprintf(dg_f, "P2 %d %d\n", e.first, e.second);
            fclose(dg_f);
        }
'''


def instrumented_reference(td):
    src = os.path.join(td, "src")
    shutil.copytree(REFSRC, src)
    path = os.path.join(src, "approximator.cpp")
    txt = open(path).read()
    roll = list(re.finditer(r"^[ \t]*dp_cur\.swap\(dp_next\);[ \t]*\n", txt, flags=re.M))
    assert len(roll) == 1, "roll of the diploid level loop (approximator.cpp:706) not found exactly once"
    txt = txt[:roll[0].end()] + DIGEST_CODE + txt[roll[0].end():]
    mat = list(re.finditer(r"^[ \t]*std::vector<std::pair<int, int>> weighted_p2_edges = materialize_edges\(sink_dp_final\.p2_tail\);[ \t]*\n", txt, flags=re.M))
    assert len(mat) == 1, "materialisation of the second edge list (approximator.cpp:782) not found exactly once"
    txt = txt[:mat[0].end()] + EDGES_CODE + txt[mat[0].end():]
    txt = "#include <cstdio>\n#include <cstdlib>\n" + txt
    open(path, "w").write(txt)
    # oracle/Makefile's own rules for the reference, pointed at the instrumented copy (REF) and an output directory in the temp dir
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), f"REF={td}", f"OUT={td}/out", f"{td}/out/DipGenie_ref"])
    return os.path.join(td, "out", "DipGenie_ref")


def run_case(binary, gfa, reads, args, threads, td):
    out = os.path.join(td, "dig.txt")
    for f in (out, out + ".edges"):
        if os.path.exists(f):
            os.remove(f)
    env = dict(os.environ, DG_REF_DIGEST_OUT=out)
    subprocess.run([binary, f"-t{threads}", *args, "-g", gfa, "-r", reads, "-o", os.path.join(td, "o.fa")], env=env, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dig = {}
    for line in open(out):
        _, l, d = line.split()
        dig[int(l)] = int(d, 16)
    n = max(dig) + 1
    assert sorted(dig) == list(range(1, n)), "a level is missing"
    arr = [0] + [dig[l] for l in range(1, n)]
    value = s_het = None
    p1, p2 = [], []
    for line in open(out + ".edges"):
        f = line.split()
        if f[0] == "V":
            value, s_het = int(f[1]), int(f[2])
        else:
            (p1 if f[0] == "P1" else p2).append([int(f[1]), int(f[2])])
    rec = dict(n_levels=n, dp_value=value, s_het=s_het, p1=p1, p2=p2, fasta_md5=hashlib.md5(open(os.path.join(td, "o.fa"), "rb").read()).hexdigest())
    if n <= 4000:
        rec["digests"] = [f"{d:016x}" for d in arr[1:]]
    else:
        import struct
        rec["digests_sha256"] = hashlib.sha256(struct.pack(f"<{n - 1}Q", *arr[1:])).hexdigest()
        rec["first8"] = [f"{d:016x}" for d in arr[1:9]]
        rec["last8"] = [f"{d:016x}" for d in arr[-8:]]
    return rec


def main():
    e2e = json.load(open(os.path.join(HERE, "e2e.json")))
    names = [n for n, c in e2e.items() if "-p2" in c.get("args", []) and not c["gfa"].startswith("<") and not c["reads"].startswith("<")]
    out = {}
    with tempfile.TemporaryDirectory() as td:
        binary = instrumented_reference(td)
        for name in names:
            c = e2e[name]
            rec = run_case(binary, os.path.join(ROOT, c["gfa"]), os.path.join(ROOT, c["reads"]), c["args"], 8 if c.get("slow") or name == "c5s" else 2, td)
            assert rec["fasta_md5"] == c["fasta_md5"] and rec["dp_value"] == c["dp_value"], (name, "the instrumented build must give the unmodified reference's answer")
            out[name] = dict(gfa=c["gfa"], reads=c["reads"], args=c["args"], **rec)
            print(name, rec["n_levels"], rec["dp_value"], len(rec["p1"]), len(rec["p2"]), flush=True)
    json.dump(out, open(os.path.join(HERE, "level_digests.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
