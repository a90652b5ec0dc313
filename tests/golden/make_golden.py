#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the REAL reference (run in the build container only).

Needs /root/reference; builds oracle/_ref via `make -C oracle ref` (reference sources compiled where
they lie, outputs only under oracle/_ref/).  Commits only data: inputs (small GFA/FASTA files under
tests/golden/e2e/) and the reference's outputs for them.  Usage: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from dipgenie_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
BIN = os.path.join(REF, "DipGenie_ref")
HARNESS = os.path.join(REF, "ref_harness")


def sh(cmd, inp=None):
    return subprocess.run(cmd, input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)


def kat_sketch():
    rng = np.random.default_rng(2024)
    def rnd(n, alpha=b"ACGT"):
        return bytes(rng.choice(np.frombuffer(alpha, np.uint8), n).tobytes()).decode()
    seqs = [rnd(150) for _ in range(12)]
    seqs += [rnd(200, b"ACGTN"), rnd(180, b"acgtACGT"), rnd(160, b"ACGTNRY"), "A" * 120, "AC" * 80, "ACG" * 60,
             rnd(55), rnd(54), rnd(56), rnd(31), "ACGT" * 40, rnd(40) * 4, rnd(700), rnd(400, b"AACGT"),
             "ACGTCATGCAGTCGTAACGTAGTCGTCACAGTCAGTCGTAGCTATGTAGCGTCAGTCAGTCAGTCGTAGCGTAACGTCGTAGTCAGT"]
    out = []
    for (k, w) in [(31, 25), (5, 3), (15, 10), (21, 11), (32, 4)]:
        inp = ("\n".join(seqs) + "\n").encode()
        hs = sh([HARNESS, "hashes", str(k), str(w)], inp).stdout.decode().split("\n")
        ms = sh([HARNESS, "minimizers", str(k), str(w)], inp).stdout.decode().split("\n")
        for s, h, m in zip(seqs, hs, ms):
            out.append(dict(seq=s, k=k, w=w, hashes=h.split(), minimizers=m.split()))
    kmers = ["A" * 31, "ACGTCATGCAGTCGTAACGTAGTCGTCACAG", "ACGTACGTACGTACGTACGTACGTACGTACG", "ANGTCATGCAGTCGTAACGTAGTCGTCACAG",
             "ACGTA", "T" * 16, "G" * 17, "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACG"]
    hk = sh([HARNESS, "hash"] + kmers).stdout.decode().split()
    return dict(windows=out, kmers=[dict(kmer=a, hash=b) for a, b in zip(kmers, hk)])


def kat_fit():
    hists = {
        "mhc4_chm13_0.5x": {1: 116960, 2: 18754, 3: 2506, 4: 353, 5: 95, 6: 44, 7: 30, 8: 21, 9: 13, 10: 7, 11: 8, 12: 3, 13: 4, 14: 4,
                            15: 7, 16: 7, 17: 1, 18: 2, 19: 1, 20: 1, 21: 2, 22: 1, 24: 3, 25: 3, 28: 2, 33: 1, 38: 1},
        "toy_single_bin": {1: 7},
        "two_bins": {1: 3, 2: 4},
    }
    rng = np.random.default_rng(11)
    # ~8x diploid-like: errors at 1-2, het peak ~4, hom peak ~8
    x = np.concatenate([rng.geometric(0.7, 40000), rng.poisson(4.0, 9000) + 1, rng.poisson(8.0, 30000) + 1, rng.poisson(16, 800) + 1])
    u, c = np.unique(x, return_counts=True)
    hists["sim_8x"] = {int(a): int(b) for a, b in zip(u, c)}
    x = np.concatenate([rng.geometric(0.8, 20000), rng.poisson(2.0, 6000) + 1, rng.poisson(4.0, 20000) + 1])
    u, c = np.unique(x, return_counts=True)
    hists["sim_4x"] = {int(a): int(b) for a, b in zip(u, c)}
    out = {}
    for name, h in hists.items():
        inp = "".join(f"{m} {f}\n" for m, f in sorted(h.items())).encode()
        lines = sh([HARNESS, "fit"], inp).stdout.decode().split("\n")
        vals = [float(v) for v in lines[0].split()]
        out[name] = dict(hist={str(k): v for k, v in h.items()}, nll=vals[0], nll_repr=lines[0].split()[0],
                         params=dict(zip(["u_v", "sd_v", "var_w", "zp_copy", "zp_copy_het", "p_d", "p_e", "err_shape"], vals[1:])),
                         labels=lines[1].strip())
        print("fit", name, lines[0], lines[1][:40], flush=True)
    return out


def run_ref(gfa, reads, args, threads=4):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.fa")
        p = sh([BIN, f"-t{threads}"] + args + ["-g", gfa, "-r", reads, "-o", out])
        txt = (p.stdout.decode(errors="replace") + p.stderr.decode(errors="replace")).replace("\r", "\n")
        fa = open(out, "rb").read()
    d = dict(args=args, fasta_md5=hashlib.md5(fa).hexdigest())
    if len(fa) < 4000:
        d["fasta"] = fa.decode()
    m = re.search(r"DP value: (-?\d+)", txt)
    if m:
        d["dp_value"] = int(m.group(1))
    m = re.search(r"recombinations in P1: (-?\d+), recombinations in P2: (-?\d+), bp of P1: (\d+), bp of P2: (\d+)", txt)
    if m:
        d.update(r1=int(m.group(1)), r2=int(m.group(2)), len1=int(m.group(3)), len2=int(m.group(4)))
    m = re.search(r"r: (\d+) obj: (-?\d+)", txt)
    if m:
        d["obj"] = int(m.group(2))
    m = re.search(r"Recombination count: (\d+)", txt)
    if m:
        d["best_r_haploid"] = int(m.group(1))
    m = re.search(r"spectrum size: (\d+)", txt)
    if m:
        d["spectrum"] = int(m.group(1))
    return d


def gfa_variants(cases):
    """bub_a with its lines rearranged, for the GFA reader's corners: S lines in shuffled order (segment ids, assigned by first
    appearance, no longer follow the names), a walk step naming a segment that does not exist (warning, step dropped), an S line
    that comes after the first W line (that walk's steps on it are dropped, gfa-io.cpp)."""
    import random
    ed = os.path.join(HERE, "e2e")
    src = open(os.path.join(ed, "bub_a.gfa")).read().split("\n")
    S = [l for l in src if l.startswith("S\t")]; L = [l for l in src if l.startswith("L\t")]
    W = [l for l in src if l.startswith("W\t")]; H = [l for l in src if l.startswith("H")]
    rnd = random.Random(3)
    S2 = S[:]; rnd.shuffle(S2)
    w0 = W[0].split("\t"); w0[6] = w0[6].replace(">5", ">5>nosuch", 1) if ">5" in w0[6] else w0[6] + ">nosuch"
    open(os.path.join(ed, "gfa_shuffled.gfa"), "w").write("\n".join(H + S2 + L + ["\t".join(w0)] + W[1:]) + "\n")
    open(os.path.join(ed, "gfa_late_segment.gfa"), "w").write("\n".join(H + S[:-1] + L + [W[0]] + [S[-1]] + W[1:]) + "\n")
    for name, base, args in (("gfa_shuffled", "gfa_shuffled", ["-p2", "-R3"]), ("gfa_shuffled_p1", "gfa_shuffled", ["-p1", "-R3"]),
                             ("gfa_late_segment", "gfa_late_segment", ["-p2", "-R3"]), ("gfa_late_segment_p1", "gfa_late_segment", ["-p1", "-R3"])):
        cases[name] = dict(gfa=f"tests/golden/e2e/{base}.gfa", reads="tests/golden/e2e/bub_a.fa",
                           **run_ref(os.path.join(ed, base + ".gfa"), os.path.join(ed, "bub_a.fa"), args))
        print(name, cases[name].get("dp_value"), cases[name]["fasta_md5"], flush=True)


def e2e():
    cases = {}
    D = os.path.join(ROOT, "tests", "data")
    toy = [("toy2_p1", "test2.gfa", "read2.fa", ["-p1", "-R2"]), ("toy2_p2", "test2.gfa", "read2.fa", ["-p2", "-R2"]),
           ("toy1_p1", "test.gfa", "read.fa", ["-p1", "-R2", "-k5", "-w3"]), ("toy1_p2", "test.gfa", "read.fa", ["-p2", "-R2", "-k5", "-w3"]),
           ("toy1_p2_R0", "test.gfa", "read.fa", ["-p2", "-R0", "-k5", "-w3"]), ("toy1_p2_R5", "test.gfa", "read.fa", ["-p2", "-R5", "-k5", "-w3"])]
    for name, g, r, a in toy:
        cases[name] = dict(gfa=f"tests/data/{g}", reads=f"tests/data/{r}", **run_ref(os.path.join(D, g), os.path.join(D, r), a))
        print(name, cases[name].get("dp_value"), cases[name]["fasta_md5"], flush=True)
    specs = [
        ("bub_a", dict(seed=1, n_bubbles=10, n_haps=4), ["-p2", "-R4", "-k11", "-w5"]),
        ("bub_b", dict(seed=2, n_bubbles=25, n_haps=6, coverage=10.0), ["-p2", "-R6", "-k11", "-w5"]),
        ("bub_c", dict(seed=3, n_bubbles=40, n_haps=8, coverage=8.0, sub_rate=0.01), ["-p2", "-R8", "-k13", "-w7"]),
        ("bub_d", dict(seed=4, n_bubbles=30, n_haps=5, seg_len=(40, 120), coverage=12.0, read_len=100), ["-p2", "-R3", "-k15", "-w8"]),
        ("bub_e", dict(seed=5, n_bubbles=60, n_haps=10, coverage=6.0, alleles=(2, 4)), ["-p2", "-R10", "-k11", "-w4"]),
        ("bub_f", dict(seed=6, n_bubbles=20, n_haps=3, coverage=20.0, sub_rate=0.005), ["-p2", "-R2", "-k9", "-w4"]),
        ("bub_g", dict(seed=7, n_bubbles=35, n_haps=12, coverage=9.0, seg_len=(35, 90)), ["-p2", "-R18", "-k31", "-w25"]),
        ("bub_h", dict(seed=8, n_bubbles=15, n_haps=4, coverage=8.0), ["-p2", "-R1", "-k11", "-w5", "-T0.75"]),
        ("bub_a_p1", dict(seed=1, n_bubbles=10, n_haps=4), ["-p1", "-R4", "-k11", "-w5"]),
        ("bub_c_p1", dict(seed=3, n_bubbles=40, n_haps=8, coverage=8.0, sub_rate=0.01), ["-p1", "-R8", "-k13", "-w7"]),
    ]
    ed = os.path.join(HERE, "e2e")
    # chr22-style stand-in (BASELINE configs[4], scaled down): 100 haplotypes, R = 32
    gfa, reads = os.path.join(ed, "c5s.gfa"), os.path.join(ed, "c5s.fa")
    if not os.path.exists(gfa):
        segs, links, walks, rd = synth.linear_panel(22, backbone_bp=8000, n_haps=100)
        synth.write_gfa(gfa, segs, links, walks)
        synth.write_fasta(reads, rd)
    cases["c5s"] = dict(gfa="tests/golden/e2e/c5s.gfa", reads="tests/golden/e2e/c5s.fa", **run_ref(gfa, reads, ["-p2", "-R32"], threads=8))
    print("c5s", cases["c5s"].get("dp_value"), cases["c5s"].get("r1"), cases["c5s"].get("r2"), cases["c5s"]["fasta_md5"], flush=True)
    for name, kw, args in specs:
        base = name.replace("_p1", "")
        gfa, reads = os.path.join(ed, base + ".gfa"), os.path.join(ed, base + ".fa")
        if not os.path.exists(gfa):
            segs, links, walks, rd = synth.random_bubble_graph(**kw)
            synth.write_gfa(gfa, segs, links, walks)
            synth.write_fasta(reads, rd)
        cases[name] = dict(gfa=f"tests/golden/e2e/{base}.gfa", reads=f"tests/golden/e2e/{base}.fa", **run_ref(gfa, reads, args))
        print(name, cases[name].get("dp_value"), cases[name].get("r1"), cases[name].get("r2"), cases[name]["fasta_md5"], flush=True)
    gfa_variants(cases)
    # MHC_4 (reference's own test data), values from reference runs in this container (-t1 and -t8 identical)
    cases["mhc4_p2"] = dict(gfa="tests/data/MHC_4.gfa.gz", reads="tests/data/CHM13_reads.fq.gz", args=["-p2", "-R18"],
                            fasta_md5="46394489af8bc9026605ddf237aca4c7", dp_value=60729, r1=17, r2=1, len1=5005629, len2=4920284,
                            obj=5282, spectrum=138834, slow=True)
    cases["mhc24_p2"] = dict(gfa="<synth.ensure_mhc24>", reads="<synth.ensure_mhc24>", args=["-p2", "-R18"],
                             fasta_md5="cd13930ac90651b7e441506c1ecd4514", dp_value=331848, r1=10, r2=8, len1=5042783, len2=5032337, obj=64156,
                             slow=True, note="reference (oracle/_ref/DipGenie_ref -t8) on dipgenie_amd.synth.mosaic_panel(seed=24, "
                             "read_seed=4): 483.9 s wall, DP 354.6 s, RSS 9.8 GB (round 2; 892.6 / 731.8 s on a busy box in round 1)")
    # BASELINE configs[1] as written needs test/HG002.mhc.2x.fq.gz, which the reference tree does not ship
    # (.MISSING_LARGE_BLOBS): seeded 2x reads from the two HG002 walks instead; reference run in this container (-t8)
    cases["mhc4_hg002_2x"] = dict(gfa="tests/data/MHC_4.gfa.gz", reads="<synth.ensure_mhc4_hg002>", args=["-p2", "-R18"],
                                  fasta_md5="b56e7ea82ccd32ce24ec641a677c4c7e", reads_md5="fac39d90919e643f272fe6a7b69de56f",
                                  dp_value=181090, r1=9, r2=9, len1=5056406, len2=5026779, obj=45480, spectrum=387040, slow=True,
                                  note="BASELINE configs[1] with the missing HG002.mhc.2x.fq.gz replaced by seeded 2x reads from the two HG002 "
                                       "walks (dipgenie_amd.synth.ensure_mhc4_hg002, seed 1); reference (oracle/_ref/DipGenie_ref -t8): "
                                       "58 s wall, DP 12.3 s")
    cases["mhc4_p1"] = dict(gfa="tests/data/MHC_4.gfa.gz", reads="tests/data/CHM13_reads.fq.gz", args=["-p1"],
                            fasta_md5="0c4df87ded10634a36db0a2c90521ff0", best_r_haploid=0, spectrum=138834, slow=True)
    return cases


def anchors(cases):
    """Anchor_hits + homo_bv of the REAL reference (ref_harness anchors = Solver::read_gfa + read_ip_reads +
    compute_and_classify_anchors, solver.cpp:27-245, 449-887) for every e2e case with inputs on disk: the text dump
    ("id hap v0,v1,..." per occurrence in Anchor_hits order, then "homo id" lines), its sha256 and counts; dumps below
    64 KB are committed verbatim."""
    out = {}
    seen = {}
    for name, c in cases.items():
        if c["gfa"].startswith("<"):
            continue
        reads = c["reads"]
        if reads == "<synth.ensure_mhc4_hg002>":
            reads = os.path.relpath(synth.ensure_mhc4_hg002("/tmp/dg_golden_hg002")[1], ROOT)
        k = next((a[2:] for a in c["args"] if a.startswith("-k")), "31")
        w = next((a[2:] for a in c["args"] if a.startswith("-w")), "25")
        T = next((a[2:] for a in c["args"] if a.startswith("-T")), "1.0")
        key = (c["gfa"], reads, k, w, T)
        if key in seen:                                   # -p1 / -p2 / -R variants share the anchor stage
            out[name] = dict(same_as=seen[key])
            continue
        seen[key] = name
        with tempfile.TemporaryDirectory() as td:
            dump = os.path.join(td, "a.txt")
            sh([HARNESS, "anchors", k, w, T, "8", os.path.join(ROOT, c["gfa"]), os.path.join(ROOT, reads) if not os.path.isabs(reads) else reads, dump])
            txt = open(dump).read()
        lines = txt.splitlines()
        occ = [l for l in lines if not l.startswith("homo")]
        d = dict(k=int(k), w=int(w), T=float(T), n_occ=len(occ), n_ids=len({l.split()[0] for l in occ}), n_homo=len(lines) - len(occ),
                 sha256=hashlib.sha256(txt.encode()).hexdigest())
        if len(txt) < 65536:
            d["dump"] = lines
        out[name] = d
        print("anchors", name, d["n_occ"], d["n_ids"], d["n_homo"], d["sha256"][:12], flush=True)
    return out


def io_corners(cases):
    """GFA / read-file corners of the ingestion stage, goldens from the reference binary:
    * gfa_flipped_walk: bub_a with its SECOND walk written back to front, every step '<' -- gfa_walk_flip (gfa-io.cpp:64-93) turns it
      round (the segments' strands are fixed by the first walk that touches them), so the answer is bub_a's;
    * gfa_reverse_step: one step of one walk reversed ('<' in a forward walk): the walk keeps its majority strand, Solver::read_gfa
      meets a reverse-strand vertex and the program exits 1 without a FASTA (solver.cpp:116-119);
    * reads as multi-line FASTA (37 columns), as FASTQ with sequence and quality wrapped over several lines, with CRLF line ends, and
      with N bases (scattered, runs, one read of N only): kseq.h semantics, hashing of non-ACGT k-mers (solver.cpp:309-313)."""
    import random
    ed = os.path.join(HERE, "e2e")
    src = open(os.path.join(ed, "bub_a.gfa")).read().split("\n")
    S = [l for l in src if l.startswith("S\t")]; L = [l for l in src if l.startswith("L\t")]
    W = [l for l in src if l.startswith("W\t")]; H = [l for l in src if l.startswith("H")]
    w1 = W[1].split("\t")
    steps = re.findall(r"[<>][^<>]+", w1[6])
    assert all(st[0] == ">" for st in steps)
    w1[6] = "".join("<" + st[1:] for st in reversed(steps))
    open(os.path.join(ed, "gfa_flipped_walk.gfa"), "w").write("\n".join(H + S + L + [W[0], "\t".join(w1)] + W[2:]) + "\n")
    w2 = W[2].split("\t")
    st2 = re.findall(r"[<>][^<>]+", w2[6])
    st2[len(st2) // 2] = "<" + st2[len(st2) // 2][1:]
    w2[6] = "".join(st2)
    open(os.path.join(ed, "gfa_reverse_step.gfa"), "w").write("\n".join(H + S + L + W[:2] + ["\t".join(w2)] + W[3:]) + "\n")
    recs = [r.split("\n", 1) for r in open(os.path.join(ed, "bub_a.fa")).read().split(">")[1:]]
    recs = [(n.strip(), q.replace("\n", "")) for n, q in recs]
    wrap = lambda q, w: "\n".join(q[i:i + w] for i in range(0, len(q), w))
    open(os.path.join(ed, "reads_multiline.fa"), "w").write("".join(f">{n}\n{wrap(q, 37)}\n" for n, q in recs))
    open(os.path.join(ed, "reads_wrapped.fq"), "w").write("".join(f"@{n}\n{wrap(q, 41)}\n+\n{wrap('I' * len(q), 41)}\n" for n, q in recs))
    open(os.path.join(ed, "reads_crlf.fa"), "wb").write("".join(f">{n}\r\n{q}\r\n" for n, q in recs).encode())
    rnd = random.Random(11)
    withn = []
    for i, (n, q) in enumerate(recs):
        q = list(q)
        if i % 7 == 0:
            for _ in range(3):
                q[rnd.randrange(len(q))] = "N"
        if i % 31 == 5:
            a = rnd.randrange(len(q) - 12)
            q[a:a + 12] = "N" * 12
        if i == 3:
            q = ["N"] * len(q)
        withn.append((n, "".join(q)))
    open(os.path.join(ed, "reads_with_N.fa"), "w").write("".join(f">{n}\n{q}\n" for n, q in withn))
    base = dict(cases["bub_a"])
    args = base["args"]
    for name, gfa, reads in (("gfa_flipped_walk", "gfa_flipped_walk.gfa", "bub_a.fa"), ("reads_multiline", "bub_a.gfa", "reads_multiline.fa"),
                             ("reads_wrapped_fastq", "bub_a.gfa", "reads_wrapped.fq"), ("reads_crlf", "bub_a.gfa", "reads_crlf.fa"),
                             ("reads_with_N", "bub_a.gfa", "reads_with_N.fa"), ("reads_with_N_p1", "bub_a.gfa", "reads_with_N.fa")):
        a = [x if x != "-p2" else "-p1" for x in args] if name.endswith("_p1") else args
        cases[name] = dict(gfa=f"tests/golden/e2e/{gfa}", reads=f"tests/golden/e2e/{reads}", **run_ref(os.path.join(ed, gfa), os.path.join(ed, reads), a))
        print(name, cases[name].get("dp_value"), cases[name]["fasta_md5"], flush=True)
    with tempfile.TemporaryDirectory() as td:                      # the run that must fail
        out = os.path.join(td, "o.fa")
        p = subprocess.run([BIN, "-t2"] + args + ["-g", os.path.join(ed, "gfa_reverse_step.gfa"), "-r", os.path.join(ed, "bub_a.fa"), "-o", out],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        cases["gfa_reverse_step"] = dict(gfa="tests/golden/e2e/gfa_reverse_step.gfa", reads="tests/golden/e2e/bub_a.fa", args=args, exit_code=p.returncode,
                                         fasta_written=os.path.exists(out) and os.path.getsize(out) > 0)
        print("gfa_reverse_step exit", p.returncode, "fasta written:", cases["gfa_reverse_step"]["fasta_written"], flush=True)
    assert cases["gfa_flipped_walk"]["fasta_md5"] == cases["bub_a"]["fasta_md5"], "a flipped walk must give bub_a's answer"


def dpg():
    """toy levelized DP graphs (dg_dp_graph dumps, what dg_dp_load_graph receives) of the two toy runs: written by the host
    pipeline behind the oracle backend (tests/harness/dg_host_oracle -X -D); the DP values they must give (8 and 14) are
    the reference's (e2e.json: toy2_p2, toy1_p2)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "harness")])
    har = os.path.join(ROOT, "tests", "harness", "dg_host_oracle")
    D = os.path.join(ROOT, "tests", "data")
    for out, g, r, extra in (("toy2_R2", "test2.gfa", "read2.fa", []), ("toy1_k5w3_R2", "test.gfa", "read.fa", ["-k5", "-w3"])):
        subprocess.check_call([har, "-q", "-t2", "-p2", "-R2", *extra, "-X", "-D", os.path.join(HERE, out), "-g", os.path.join(D, g),
                               "-r", os.path.join(D, r), "-o", os.path.join("/tmp", out + ".fa")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    what = sys.argv[1:] or ["sketch", "fit", "e2e"]
    if "sketch" in what:
        json.dump(kat_sketch(), open(os.path.join(HERE, "kat_sketch.json"), "w"), indent=0)
    if "e2e" in what:
        json.dump(e2e(), open(os.path.join(HERE, "e2e.json"), "w"), indent=1)
    if "io" in what:                                              # adds the ingestion corner cases to the existing e2e.json
        cases = json.load(open(os.path.join(HERE, "e2e.json")))
        io_corners(cases)
        json.dump(cases, open(os.path.join(HERE, "e2e.json"), "w"), indent=1)
    if "anchors" in what:
        json.dump(anchors(json.load(open(os.path.join(HERE, "e2e.json")))), open(os.path.join(HERE, "anchors.json"), "w"), indent=0)
    if "dpg" in what:
        dpg()
    if "fit" in what:
        json.dump(kat_fit(), open(os.path.join(HERE, "kat_fit.json"), "w"), indent=1)
