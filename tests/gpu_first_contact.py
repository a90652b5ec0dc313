"""Ad-hoc first GPU contact script (superseded by the pytest -m gpu suite)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as orc
from dipgenie_amd import capi

ctx = capi.Context(0)
print("device:", ctx.device_info())
kat = [b"A" * 31, b"ACGTCATGCAGTCGTAACGTAGTCGTCACAG", b"ACGTACGTACGTACGTACGTACGTACGTACG", b"ANGTCATGCAGTCGTAACGTAGTCGTCACAG"]
h = ctx.hash_kmers(b"".join(kat), 31)
print("hash KAT:", [hex(int(x)) for x in h], all(int(h[i]) == orc.hash_kmer(kat[i]) for i in range(4)))

rng = np.random.default_rng(7)
def rnd(n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, np.uint8), n).tobytes())
reads = [rnd(150) for _ in range(2000)] + [rnd(40), rnd(55), rnd(54), b"", rnd(300, b"ACGTN"), rnd(200, b"acgtACGT"), rnd(1000)]
reads += [reads[0], reads[1][:100] + reads[2][:80]]
for (k, w) in [(31, 25), (5, 3), (15, 10)]:
    t0 = time.time(); hg, cg = ctx.sketch_reads(reads, k, w); t1 = time.time()
    ho, co = orc.sketch_reads(reads, k, w)
    print(f"sketch_reads k={k} w={w}: gpu {hg.size} oracle {ho.size} equal={np.array_equal(hg, ho) and np.array_equal(cg, co)} ({t1-t0:.3f}s)")
hap = rnd(300000) + rnd(50, b"ACGTN") + rnd(20000, b"acgt") + rnd(100000)
for (k, w) in [(31, 25), (5, 3)]:
    hg, pg = ctx.sketch_haplotype(hap, k, w)
    ho, po = orc.minimizers(hap, k, w)
    print(f"sketch_haplotype k={k} w={w}: gpu {hg.size} oracle {ho.size} equal={np.array_equal(hg, ho) and np.array_equal(pg, po)}")

for name in ["tests/golden/toy2_R2.dpg", "tests/golden/toy1_k5w3_R2.dpg", "tests/data/mhc4.dpg"]:
    g = capi.DpGraphArrays.load(os.path.join(ROOT, name))
    ctx.dp_set_option("digest", 1)
    t0 = time.time(); ctx.dp_load_graph(g); t1 = time.time(); out = ctx.dp_run(); t2 = time.time()
    tm = ctx.dp_timing()
    dg = ctx.dp_level_digest(g.n_levels)
    t3 = time.time(); ref = orc.dp_solve(g, want_digest=True); t4 = time.time()
    same = (out.value, out.s_het, out.p1, out.p2) == (ref["value"], ref["s_het"], ref["p1"], ref["p2"])
    print(f"{name}: value {out.value}/{ref['value']} s_het {out.s_het}/{ref['s_het']} paths_equal={out.p1 == ref['p1'] and out.p2 == ref['p2']} "
          f"ALL_EQUAL={same} digest_equal={np.array_equal(dg[1:], ref['digest'][1:])} cells={out.cells}=={ref['cells']} "
          f"load {t1-t0:.2f}s run {t2-t1:.2f}s (delta {tm.delta_ms:.1f} fwd {tm.forward_ms:.1f} tb {tm.traceback_ms:.1f} ms) oracle {t4-t3:.2f}s")
    ctx.dp_set_option("digest", 0)
    out2 = ctx.dp_run(); tm = ctx.dp_timing()
    print(f"   no-digest rerun: equal={out2.key() == out.key()} delta {tm.delta_ms:.1f} fwd {tm.forward_ms:.1f} tb {tm.traceback_ms:.1f} total {tm.total_ms:.1f} ms")
