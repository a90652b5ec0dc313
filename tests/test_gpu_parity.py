"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the committed goldens.
Bit-exact bar (integer DP, integer hashes): every value, s_het, edge list, level digest and hash must be
identical.  Nothing here reads /root/reference."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import graphgen
import oracle_py as orc
from dipgenie_amd import capi, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KAT = json.load(open(os.path.join(HERE, "golden", "kat_sketch.json")))
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))


def _rnd(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alpha, np.uint8), n).tobytes())


# ------------------------------------------------------------------------------------- sketch
def test_hash_kat(gpu_ctx):
    kmers = [e for e in KAT["kmers"]]
    for e in kmers:
        k = len(e["kmer"])
        assert int(gpu_ctx.hash_kmers(e["kmer"].encode(), k)[0]) == int(e["hash"], 16)
    rng = np.random.default_rng(1)
    for k in (1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 64, 100):
        blob = _rnd(rng, 50 * k, b"ACGTNacgt")
        got = gpu_ctx.hash_kmers(blob, k)
        assert [int(x) for x in got] == [orc.hash_kmer(blob[i * k:(i + 1) * k]) for i in range(50)], k


def test_sketch_golden_vectors(gpu_ctx):
    """reference compute_hashes / index_kmers outputs for ACGT, N, lower-case, repeats, short inputs"""
    for e in KAT["windows"]:
        s = e["seq"].encode()
        h, c = gpu_ctx.sketch_reads([s], e["k"], e["w"])
        assert [f"{int(x):016x}" for x in h] == e["hashes"], (e["k"], e["w"], e["seq"][:24])
        assert np.all(c == 1)
        hm, pm = gpu_ctx.sketch_haplotype(s, e["k"], e["w"])
        assert [f"{int(x):016x}" for x in hm] == e["minimizers"], (e["k"], e["w"], e["seq"][:24])


@pytest.mark.parametrize("k,w", [(31, 25), (5, 3), (15, 10), (32, 5), (40, 6), (11, 1), (3, 40)])
def test_sketch_reads_vs_oracle(gpu_ctx, k, w):
    rng = np.random.default_rng(100 + k)
    reads = [_rnd(rng, 150) for _ in range(1500)]
    reads += [_rnd(rng, n) for n in (0, 1, k - 1, k, k + w - 2, k + w - 1, k + w, 129 + k + w, 1000, 5000)]
    reads += [_rnd(rng, 300, b"ACGTN"), _rnd(rng, 300, b"acgtnACGTN"), _rnd(rng, 400, b"ACGTRYKM*"), b"N" * 200, b"A" * 300,
              b"AC" * 150, reads[0], reads[1][:90] + reads[2][:90]]
    hg, cg = gpu_ctx.sketch_reads(reads, k, w)
    ho, co = orc.sketch_reads(reads, k, w)
    assert np.array_equal(hg, ho) and np.array_equal(cg, co)
    assert np.all(np.diff(hg.astype(np.uint64)) > 0)          # globally sorted, distinct


SPECTRUM_MODES = {          # name: (options, spectrum_path expected: 0 tile kernel fills the buckets, 1 exact placement, 2 generic sort)
    "buckets": ({}, 0),
    "generic": ({"spectrum_mode": 1}, 2),
    "exact_placement": ({"spectrum_mode": 2}, 1),
    "two_buckets": ({"bucket_bits": 1, "bucket_stride": 1 << 20}, 0),
    "32k_buckets": ({"bucket_bits": 15, "bucket_stride": 4096}, 0),
    "stride_overrun": ({"bucket_bits": 9, "bucket_stride": 256}, 0),                 # full buckets: their further pairs go through the shared spill list
    "spill_overrun": ({"bucket_bits": 9, "bucket_stride": 256, "spill_cap": 1000}, 1),   # ... which runs over too: repeated with exact placement
    "no_spill_list": ({"bucket_bits": 9, "bucket_stride": 256, "spill_cap": -1}, 1),
    "results_over_stride": ({"bucket_bits": 4, "bucket_stride": 512}, 1),           # more distinct hashes in a bucket than it has slots: exact placement
    "host_segments": ({"bucket_bits": 3, "bucket_stride": 1 << 17, "residual_cap": 1}, 0),                       # residual lists overflow: those buckets are finished per segment with rocPRIM
    "host_segments_exact": ({"spectrum_mode": 2, "bucket_bits": 3, "residual_cap": 1}, 1),
    "too_many_host_segments": ({"bucket_bits": 3, "bucket_stride": 1 << 17, "residual_cap": 1, "host_buckets": 2}, 2),   # more buckets need the host than allowed: the generic sort after all
}


def _spectrum_reads(rng):
    reads = [_rnd(rng, 150) for _ in range(4000)]
    genome = _rnd(rng, 30000)
    reads += [genome[s:s + 150] for s in rng.integers(0, len(genome) - 150, 6000)]          # 30x: runs of ~20 reads per hash
    x, y = _rnd(rng, 500), _rnd(rng, 700)
    reads += [x + y + x, x + x + x + y, y + synth.revcomp(y)]                                 # the same hash in several tiles of one read
    reads += [x[:60] + x[:60] + x[:40], y[:75] + synth.revcomp(y[:75])]                       # ... and twice in one tile
    reads += [genome[s:s + 300] for s in rng.integers(0, len(genome) - 300, 300)]             # two-tile reads: the same (hash, read) may come from both tiles
    reads += [reads[5]] * 700 + [b"", b"ACGT", _rnd(rng, 300, b"ACGTN"), b"A" * 400, b"AC" * 200]
    return reads


def _reset_spectrum_options(ctx):
    for key in ("spectrum_mode", "bucket_bits", "bucket_stride", "residual_cap", "host_buckets", "spill_cap"):
        ctx.sketch_set_option(key, 0)


@pytest.mark.parametrize("mode", list(SPECTRUM_MODES))
def test_sketch_spectrum_paths_agree(gpu_ctx, mode):
    """Sp_R from hash-range buckets resolved in LDS tables == the stable radix sort of all pairs == the oracle, in every
    execution mode (who fills the buckets, bucket counts, residual lists finished by the host, the give-up routes)."""
    rng = np.random.default_rng(4242)
    reads = _spectrum_reads(rng)
    ho, co = orc.sketch_reads(reads, 21, 11)
    opts, want_path = SPECTRUM_MODES[mode]
    try:
        for key, v in opts.items():
            gpu_ctx.sketch_set_option(key, v)
        hg, cg = gpu_ctx.sketch_reads(reads, 21, 11)
        path, ovf, spilled = gpu_ctx.sketch_stat("spectrum_path"), gpu_ctx.sketch_stat("overflow_buckets"), gpu_ctx.sketch_stat("spilled_pairs")
    finally:
        _reset_spectrum_options(gpu_ctx)
    assert np.array_equal(hg, ho) and np.array_equal(cg, co)
    assert path == want_path
    if mode.startswith("host_segments"):
        assert ovf > 0
    if mode == "buckets":
        assert ovf == 0
    if mode == "stride_overrun":
        assert spilled > 0


def test_sketch_spectrum_heavy_hitters(gpu_ctx):
    """a hash held by more reads than a bucket has slots (13,000 copies of one read): the bucket's further pairs go through the shared
    spill list; with no room there the pass is repeated with exact placement, and that ctx keeps placing exactly afterwards"""
    rng = np.random.default_rng(77)
    one = _rnd(rng, 150)
    reads = [_rnd(rng, 150) for _ in range(3000)] + [one] * 13000
    ctx = capi.Context(0)
    hg, cg = ctx.sketch_reads(reads, 31, 25)
    assert ctx.sketch_stat("spectrum_path") == 0 and ctx.sketch_stat("spilled_pairs") > 0 and ctx.sketch_stat("overflow_buckets") == 0
    ctx.sketch_set_option("spill_cap", 100)
    hx, cx = ctx.sketch_reads(reads, 31, 25)
    assert ctx.sketch_stat("spectrum_path") == 1
    ctx.sketch_reads(reads[:5000], 31, 25)
    assert ctx.sketch_stat("spectrum_path") == 1                        # sticky
    ctx.sketch_set_option("spill_cap", 0)
    ctx.sketch_set_option("spectrum_mode", 1)
    h2, c2 = ctx.sketch_reads(reads, 31, 25)
    assert np.array_equal(hg, h2) and np.array_equal(cg, c2) and np.array_equal(hx, h2) and np.array_equal(cx, c2)
    ho, co = orc.sketch_reads(reads[:3000] + [one], 31, 25)
    h1, c1 = orc.sketch_reads([one], 31, 25)
    want = dict(zip(ho.tolist(), co.tolist()))
    for h in h1.tolist():
        want[h] += 12999
    assert dict(zip(hg.tolist(), cg.tolist())) == want


@pytest.mark.parametrize("mode", ["buckets", "exact_placement", "host_segments"])
def test_sketch_spectrum_multi_tile_reads(gpu_ctx, mode):
    """read sets made of reads longer than one tile (128 windows): the same (hash, read) can be emitted by several tiles; the
    bucket's hash set keeps one of each (repeats inside a read are frequent here: every read carries a duplicated segment)"""
    rng = np.random.default_rng(9)
    genome = _rnd(rng, 20000)
    reads = []
    for s0 in rng.integers(0, len(genome) - 400, 1500):
        r = genome[s0:s0 + 400]
        reads.append(r + r[40:240] + synth.revcomp(r[100:300]))
    reads += [_rnd(rng, 150) for _ in range(300)] + [reads[0]] * 50
    ho, co = orc.sketch_reads(reads, 21, 11)
    opts, want_path = SPECTRUM_MODES[mode]
    try:
        for key, v in opts.items():
            gpu_ctx.sketch_set_option(key, v)
        hg, cg = gpu_ctx.sketch_reads(reads, 21, 11)
        assert gpu_ctx.sketch_stat("spectrum_path") == want_path
    finally:
        _reset_spectrum_options(gpu_ctx)
    assert np.array_equal(hg, ho) and np.array_equal(cg, co)


def test_sketch_reads_empty_and_ragged(gpu_ctx):
    for reads in ([], [b""], [b"ACGT"], [b"", b"", b"ACGTACGT"]):
        h, c = gpu_ctx.sketch_reads(reads, 31, 25)
        assert h.size == 0 and c.size == 0


@pytest.mark.parametrize("k,w", [(31, 25), (5, 3), (21, 11)])
def test_sketch_haplotype_vs_oracle(gpu_ctx, k, w):
    rng = np.random.default_rng(7 + k)
    hap = _rnd(rng, 200000) + _rnd(rng, 80, b"ACGTN") + _rnd(rng, 30000, b"acgt") + b"A" * 500 + _rnd(rng, 100000) + b"ACG" * 300
    hg, pg = gpu_ctx.sketch_haplotype(hap, k, w)
    ho, po = orc.minimizers(hap, k, w)
    assert np.array_equal(hg, ho) and np.array_equal(pg, po)


def test_sketch_full_size_properties(gpu_ctx):
    """config-4 scale (1M x 150 bp) -- size-independent properties instead of the (slow) oracle:
    read order and strand do not matter, duplicating the read set doubles every count, counts sum to
    the number of (read, distinct hash) pairs of a sampled subset."""
    rng = np.random.default_rng(99)
    genome = _rnd(rng, 2_000_000)
    n = 200_000
    starts = rng.integers(0, len(genome) - 150, n)
    reads = [genome[s:s + 150] for s in starts]
    h1, c1 = gpu_ctx.sketch_reads(reads, 31, 25)
    perm = rng.permutation(n)
    reads2 = [synth.revcomp(reads[i]) if i & 1 else reads[i] for i in perm]
    h2, c2 = gpu_ctx.sketch_reads(reads2, 31, 25)
    assert np.array_equal(h1, h2) and np.array_equal(c1, c2)          # canonical k-mers: strand/order invariant
    h3, c3 = gpu_ctx.sketch_reads(reads + reads, 31, 25)
    assert np.array_equal(h1, h3) and np.array_equal(2 * c1, c3)
    sub = reads[:3000]
    hs, cs = gpu_ctx.sketch_reads(sub, 31, 25)
    ho, co = orc.sketch_reads(sub, 31, 25)
    assert np.array_equal(hs, ho) and np.array_equal(cs, co)
    assert int(c1.sum()) >= int(cs.sum())


# ------------------------------------------------------------------------------------- DP
def _dp_both(ctx, g, digest=True):
    ctx.dp_set_option("digest", 1 if digest else 0)
    out = ctx.dp_solve(g)
    ref = orc.dp_solve(g, want_digest=digest)
    assert (out.value, out.s_het) == (ref["value"], ref["s_het"])
    assert out.p1 == ref["p1"] and out.p2 == ref["p2"]
    assert (out.cells, out.relaxations) == (ref["cells"], ref["relaxations"])
    if digest:
        dg = ctx.dp_level_digest(g.n_levels)
        assert np.array_equal(dg[1:], ref["digest"][1:])
    ctx.dp_set_option("digest", 0)
    return out


def _fan_in_graph(k):
    """source -> k vertices -> ONE vertex (in-degree k, alternating weights) -> sink, with colours"""
    level_off = np.array([0, 1, 1 + k, 2 + k, 3 + k], np.int32)
    out, w = [], []
    out_off = [0]
    for j in range(k):                                        # source -> k vertices
        out.append(1 + j); w.append(0)
    out_off.append(len(out))
    for j in range(k):                                        # all k -> one vertex (in-degree 90), alternating weights per source
        out.append(1 + k); w.append(j & 1)
        out_off.append(len(out))
    out.append(2 + k); w.append(0); out_off.append(len(out))  # -> sink
    out_off.append(len(out))
    nV = 3 + k
    hom = [[] for _ in range(nV)]
    het = [[] for _ in range(nV)]
    for j in range(k):
        het[1 + j] = [j % 7]
        hom[1 + j] = [10 + j % 5]
    hom[1 + k] = [10, 12]
    het[1 + k] = [3]
    def csr(ls):
        off = np.zeros(nV + 1, np.int64); off[1:] = np.cumsum([len(x) for x in ls])
        return off, np.array([c for x in ls for c in x], np.int32)
    ho, hc = csr(hom); to, tc = csr(het)
    g = capi.DpGraphArrays(2, level_off=level_off, out_off=np.array(out_off, np.int64), out_dst=np.array(out, np.int32),
                           out_w=np.array(w, np.uint8), hom_off=ho, hom_col=hc, het_off=to, het_col=tc)
    return g


def test_dp_toy_goldens(gpu_ctx):
    out = _dp_both(gpu_ctx, capi.DpGraphArrays.load(os.path.join(HERE, "golden", "toy2_R2.dpg")))
    assert out.value == 8 and len(out.p1) - 1 == 1 and len(out.p2) - 1 == 0        # reference: DP value 8, r1=1 r2=0
    out = _dp_both(gpu_ctx, capi.DpGraphArrays.load(os.path.join(HERE, "golden", "toy1_k5w3_R2.dpg")))
    assert out.value == 14 and len(out.p1) - 1 == 1 and len(out.p2) - 1 == 1


@pytest.mark.parametrize("seed", range(24))
def test_dp_random_levelized(gpu_ctx, seed):
    kw = [dict(), dict(max_width=30, n_levels=40, R=6), dict(max_width=3, n_levels=200, R=2), dict(R=0), dict(p_w1=0.9, R=18),
          dict(p_colour=0.0), dict(p_colour=1.0, max_list=9, n_colours=10), dict(max_width=70, n_levels=10, R=4, extra_edges=3.0),
          dict(min_width=1, max_width=1, n_levels=30, R=3), dict(max_width=12, n_levels=1500, R=5, p_colour=0.1),
          dict(max_width=40, n_levels=25, R=33, p_w1=0.5), dict(n_levels=2, R=2)][seed % 12]
    g = graphgen.random_levelized(1000 + seed, **kw)
    _dp_both(gpu_ctx, g)


def test_dp_device_tables_equal_host_tables(gpu_ctx):
    """dg_dp_load_graph builds the sweep's tables with device kernels (dg_dp_build.hip); the host construction (option
    host_tables) is its twin: all twelve tables must come out byte-identical, on narrow / wide / fan-in / colourless / giant-column
    (in-degree > 64 and > 255) graphs, with and without row in-edge matrices"""
    cases = [dict(), dict(max_width=30, n_levels=40, R=6), dict(max_width=3, n_levels=200, R=2), dict(R=0), dict(p_colour=0.0),
             dict(p_colour=1.0, max_list=9, n_colours=10), dict(max_width=70, n_levels=10, R=4, extra_edges=3.0),
             dict(min_width=1, max_width=1, n_levels=30, R=3), dict(max_width=12, n_levels=1500, R=5, p_colour=0.1), dict(n_levels=2, R=2),
             dict(max_width=64, n_levels=30, R=3, p_w1=0.5, p_colour=0.8), dict(max_width=3, n_levels=8, R=2, extra_edges=100.0),
             dict(max_width=3, n_levels=6, R=1, extra_edges=300.0, dup_edges=False), dict(min_width=20, max_width=24, n_levels=8, R=2, extra_edges=70.0)]
    graphs = [graphgen.random_levelized(9100 + q, **kw) for q, kw in enumerate(cases)]
    graphs += [capi.DpGraphArrays.load(os.path.join(HERE, "golden", n)) for n in ("toy2_R2.dpg", "toy1_k5w3_R2.dpg")]
    try:
        for rowx in (1, 0):
            gpu_ctx.dp_set_option("rowx", rowx)
            for q, g in enumerate(graphs):
                gpu_ctx.dp_set_option("host_tables", 1)
                gpu_ctx.dp_load_graph(g)
                want = gpu_ctx.dp_table_digest()
                gpu_ctx.dp_set_option("host_tables", 0)
                gpu_ctx.dp_load_graph(g)
                got = gpu_ctx.dp_table_digest()
                assert got == want, (rowx, q, [t for t in got if got[t] != want[t]])
                out = gpu_ctx.dp_run()
                ref = orc.dp_solve(g)
                assert (out.value, out.s_het, out.p1, out.p2, out.cells, out.relaxations) == (ref["value"], ref["s_het"], ref["p1"], ref["p2"], ref["cells"], ref["relaxations"]), q
    finally:
        gpu_ctx.dp_set_option("rowx", 1)
        gpu_ctx.dp_set_option("host_tables", 0)


@pytest.mark.parametrize("mode", ["generic", "no_adaptive", "no_coop", "force_coop", "no_rowx", "general_chain", "no_l2_prefetch", "no_delta_overlap", "forced_delta_overlap", "no_far_prefetch", "host_tables"])
def test_dp_alternative_kernels(gpu_ctx, mode):
    """the generic fallback sweep, the fixed-RC launch, cooperative rows off / forced, the path without row in-edge
    matrices and the general chain walk (in place of the lean one) must all give the oracle's answer"""
    opts = {"generic": {"fast": 0}, "no_adaptive": {"adaptive_rc": 0}, "no_coop": {"coop": 0}, "force_coop": {"coop": 2}, "no_rowx": {"rowx": 0},
            "general_chain": {"lean_chain": 0}, "no_l2_prefetch": {"l2_prefetch": 0}, "no_delta_overlap": {"delta_overlap": 0}, "forced_delta_overlap": {"delta_overlap": 2}, "no_far_prefetch": {"pf_far": 0}, "host_tables": {"host_tables": 1}}[mode]
    try:
        for k, v in opts.items():
            gpu_ctx.dp_set_option(k, v)
        for seed, kw in [(1, dict(max_width=12, n_levels=300, R=5)), (2, dict(max_width=40, n_levels=60, R=18, p_w1=0.5)),
                         (3, dict(max_width=6, n_levels=2000, R=3, p_colour=0.2)), (4, dict(R=33, max_width=20, n_levels=50))]:
            g = graphgen.random_levelized(7000 + seed, **kw)
            out = gpu_ctx.dp_solve(g)
            ref = orc.dp_solve(g)
            assert (out.value, out.s_het, out.p1, out.p2) == (ref["value"], ref["s_het"], ref["p1"], ref["p2"]), (mode, seed)
        # fan-in rows (the cooperative tasks' and the row matrices' subject), with all level digests
        for seed, kw in [(5, dict(max_width=30, n_levels=120, R=18, p_w1=0.3, p_colour=0.5)), (6, dict(max_width=60, n_levels=40, R=32, p_w1=0.6)),
                         (7, dict(max_width=64, n_levels=30, R=3, p_w1=0.5, p_colour=0.8)), (8, dict(max_width=70, n_levels=12, R=4, extra_edges=3.0))]:
            _dp_both(gpu_ctx, graphgen.random_levelized(7100 + seed, **kw))
    finally:
        for k, v in {"fast": 1, "adaptive_rc": 1, "coop": 1, "rowx": 1, "lean_chain": 1, "l2_prefetch": 6, "delta_overlap": 1, "pf_far": 128, "host_tables": 0}.items():
            gpu_ctx.dp_set_option(k, v)


def test_dp_symmetric_form_measurement_build(built_hip):
    """The symmetric form of the sweep (dp_sweep_sym_kernel, dipgenie_amd/csrc/dg_dp_sweep_sym.hip: cells (i2, j2 >= i2) computed, value
    and own back-pointer stored for both (i2, j2) and (j2, i2)) is correct and slower than the plain form on the levels it was built for
    (DESIGN.md s3.3), so it lives in a measurement build only (make -C dipgenie_amd/csrc sym -> bin/libdipgenie_hip_sym.so; not built
    by __graft_entry__.build(); the record of this test on the MI355X: profiles/r04_sym_parity_and_ab.txt).  Forced onto EVERY level that
    can take it (sym = 2): value, s_het, edge lists and every level digest -- the value and the winning predecessor pair of every cell,
    mirror cells included, whose tie-break order (approximator.cpp:657-659) is NOT the transpose of their twin's -- must equal the
    oracle's, for 1 / 2 / 3 / 4 / 6 / 8 recombination counts per task, fan-in rows inline, plain launches, host-built tables, no row
    matrices.  Graphs: the 12 shapes of test_dp_random_levelized, fan-in rows (workgroups of their own), giant columns (in-degree > 64),
    vertices without in-edges (dead rows / columns), widths around the 16-row tile and the 64-lane block, R + 1 not a multiple of the
    chunk.  The product library must not know the option at all."""
    with pytest.raises(capi.DgError):
        c = capi.Context(0)
        try:
            c.dp_set_option("sym", 1)
        finally:
            c.close()
    lib = os.path.join(ROOT, "bin", "libdipgenie_hip_sym.so")
    if not os.path.exists(lib):
        pytest.skip("measurement build absent (make -C dipgenie_amd/csrc sym)")
    p = subprocess.run([sys.executable, os.path.join(HERE, "sym_parity_main.py")], env=dict(os.environ, DG_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0 and b"symmetric form parity ok" in p.stdout, p.stdout.decode()[-2000:] + p.stderr.decode()[-3000:]


@pytest.mark.parametrize("lean", [1, 0])
def test_dp_corrupt_lattice_is_an_error_not_a_fault(gpu_ctx, lean):
    """a damaged back-pointer lattice (one level overwritten between sweep and walk) must end in DG_ERR_STATE from both chain
    walks -- never in a wild colour-list read of the finish kernel: 0xFF = the "unreachable" word, 0x01 = rank 1 everywhere
    (a vertex with one in-edge then yields the all-ones guard word, whose ids lie outside every level), 0x30 = ranks beyond any list"""
    g = graphgen.random_levelized(9300, max_width=12, n_levels=120, R=4, p_colour=0.5)
    ref = orc.dp_solve(g)
    try:
        gpu_ctx.dp_set_option("lean_chain", lean)
        for level in (119, 60, 7, 1):
            for byte in (0xFF, 0x01, 0x30):
                gpu_ctx.dp_set_option("test_poison_level", level)
                gpu_ctx.dp_set_option("test_poison_byte", byte)
                gpu_ctx.dp_load_graph(g)
                try:
                    out = gpu_ctx.dp_run()
                except capi.DgError as e:
                    assert "corrupt" in str(e) or "disagree" in str(e), (level, byte, str(e))
                else:
                    # a poisoned level the answer path happens to cross with an in-range word may still decode: the run then ends
                    # normally only if the walked path is a path of the graph that re-scores to the DP value (dg_dp_run checks both),
                    # so the value is the sweep's; level 119 / byte 0xFF can never pass
                    assert byte != 0xFF, (level, byte)
                    assert out.value == ref["value"], (level, byte)
        gpu_ctx.dp_set_option("test_poison_level", 0)
        gpu_ctx.dp_load_graph(g)
        out = gpu_ctx.dp_run()
        assert (out.value, out.s_het, out.p1, out.p2) == (ref["value"], ref["s_het"], ref["p1"], ref["p2"])
    finally:
        gpu_ctx.dp_set_option("test_poison_level", 0)
        gpu_ctx.dp_set_option("test_poison_byte", 0xFF)
        gpu_ctx.dp_set_option("lean_chain", 1)


def test_dp_launch_profile_counts_every_level(gpu_ctx):
    """dg_dp_get_launch_profile (what bench.py matches its rocprof summary against): one launch per destination level,
    the same on a replayed pass (hipGraph batches) as on the capturing one, and with plain launches"""
    g = graphgen.random_levelized(7400, max_width=20, n_levels=2500, R=6, p_w1=0.4)
    gpu_ctx.dp_load_graph(g)
    profiles = []
    for _ in range(3):
        gpu_ctx.dp_run()
        profiles.append(gpu_ctx.dp_launch_profile())
    assert sum(profiles[0].values()) == g.n_levels - 1 and profiles[0] == profiles[1] == profiles[2], profiles
    try:
        gpu_ctx.dp_set_option("graph_batch", 0)
        gpu_ctx.dp_run()
        assert gpu_ctx.dp_launch_profile() == profiles[0]
    finally:
        gpu_ctx.dp_set_option("graph_batch", -1)


def test_dp_large_recombination_budget(gpu_ctx):
    """R >= 1024 (the level-0 initialisation used to be one over-sized block): values, edge lists and digests"""
    _dp_both(gpu_ctx, graphgen.random_levelized(7300, max_width=5, n_levels=12, R=1100, p_w1=0.5))
    _dp_both(gpu_ctx, graphgen.random_levelized(7301, max_width=3, n_levels=8, R=4096, p_w1=0.9))


@pytest.mark.parametrize("seg_cells", [1, 5000, 200000])
def test_dp_segmented_lattice(gpu_ctx, seg_cells):
    """checkpoint + recompute (lattices beyond HBM, BASELINE config 5): forced here with tiny segments; value, s_het,
    edge lists and every level digest must not change"""
    try:
        gpu_ctx.dp_set_option("segment_cells", seg_cells)
        for seed, kw in [(11, dict(max_width=14, n_levels=400, R=6)), (12, dict(max_width=45, n_levels=70, R=18, p_w1=0.5)),
                         (13, dict(n_levels=2, R=2)), (14, dict(max_width=8, n_levels=3000, R=3, p_colour=0.3)), (15, dict(R=33, max_width=25, n_levels=90))]:
            _dp_both(gpu_ctx, graphgen.random_levelized(8000 + seed, **kw))
        _dp_both(gpu_ctx, capi.DpGraphArrays.load(os.path.join(HERE, "golden", "toy1_k5w3_R2.dpg")))
    finally:
        gpu_ctx.dp_set_option("segment_cells", 0)


@pytest.mark.parametrize("cap,seg", [(1, 0), (400, 0), (20000, 0), (400, 5000), (1, 1)])
def test_dp_delta_windows(gpu_ctx, cap, seg):
    """score-delta matrices that outgrow their budget are recomputed window by window right before the levels that read
    them (chr22-scale panels: hundreds of GB otherwise) -- forced here with tiny budgets, alone and together with a
    segmented lattice (whose second pass re-enters windows half way)"""
    try:
        gpu_ctx.dp_set_option("delta_cap_entries", cap)
        gpu_ctx.dp_set_option("segment_cells", seg)
        for seed, kw in [(31, dict(max_width=14, n_levels=300, R=6, p_colour=0.6)), (32, dict(max_width=45, n_levels=60, R=18, p_w1=0.5, p_colour=0.9)),
                         (33, dict(n_levels=2, R=2)), (34, dict(max_width=8, n_levels=2000, R=3, p_colour=0.3)), (35, dict(R=33, max_width=25, n_levels=90, p_colour=0.1))]:
            _dp_both(gpu_ctx, graphgen.random_levelized(8200 + seed, **kw))
        _dp_both(gpu_ctx, capi.DpGraphArrays.load(os.path.join(HERE, "golden", "toy1_k5w3_R2.dpg")))
    finally:
        gpu_ctx.dp_set_option("delta_cap_entries", 0)
        gpu_ctx.dp_set_option("segment_cells", 0)


@pytest.mark.parametrize("ahead,cap,seg", [(0, 0, 0), (1, 0, 0), (7, 0, 0), (7, 400, 0), (3, 400, 5000), (1000, 0, 5000)])
def test_dp_sweep_lookahead(gpu_ctx, ahead, cap, seg):
    """the sweep streams the graph tables of the next batch of levels through the Infinity Cache (reads only): any batch
    size, with delta windows and lattice segments cutting the ranges, leaves every result and level digest unchanged"""
    try:
        gpu_ctx.dp_set_option("warm_ahead", ahead)
        gpu_ctx.dp_set_option("delta_cap_entries", cap)
        gpu_ctx.dp_set_option("segment_cells", seg)
        for seed, kw in [(41, dict(max_width=14, n_levels=300, R=6, p_colour=0.6)), (42, dict(max_width=45, n_levels=60, R=18, p_w1=0.5, p_colour=0.9)),
                         (43, dict(n_levels=2, R=2)), (44, dict(max_width=8, n_levels=2000, R=3, p_colour=0.3))]:
            _dp_both(gpu_ctx, graphgen.random_levelized(8300 + seed, **kw))
        _dp_both(gpu_ctx, capi.DpGraphArrays.load(os.path.join(HERE, "golden", "toy1_k5w3_R2.dpg")))
    finally:
        gpu_ctx.dp_set_option("warm_ahead", 128)
        gpu_ctx.dp_set_option("delta_cap_entries", 0)
        gpu_ctx.dp_set_option("segment_cells", 0)


@pytest.mark.parametrize("batch,seg,chunk", [(0, 0, 0), (1, 0, 0), (7, 0, 0), (1000, 0, 0), (7, 5000, 0), (5, 0, 3000)])
def test_dp_graph_batches(gpu_ctx, batch, seg, chunk):
    """level launches captured into hipGraphs and replayed: the capturing pass and the replaying passes give the oracle's
    answer and level digests, also with lattice segments / pool chunks cutting the batches, and after another graph was
    loaded into the same context (stale batches must not survive)"""
    try:
        gpu_ctx.dp_set_option("segment_cells", seg)
        if chunk:
            gpu_ctx.dp_set_option("lattice_chunk_cells", chunk)
        gpu_ctx.dp_set_option("graph_batch", batch)
        for seed, kw in [(51, dict(max_width=14, n_levels=300, R=6, p_colour=0.6)), (52, dict(max_width=45, n_levels=60, R=18, p_w1=0.5, p_colour=0.9)),
                         (53, dict(n_levels=2, R=2)), (54, dict(max_width=8, n_levels=2000, R=3, p_colour=0.3))]:
            g = graphgen.random_levelized(8400 + seed, **kw)
            _dp_both(gpu_ctx, g)
            first = gpu_ctx.dp_run()
            for _ in range(2):                                  # replays of the cached batches
                assert gpu_ctx.dp_run().key() == first.key()
    finally:
        gpu_ctx.dp_set_option("graph_batch", -1)
        gpu_ctx.dp_set_option("segment_cells", 0)
        if chunk:
            gpu_ctx.dp_set_option("lattice_chunk_cells", 1 << 31)


def test_dp_graph_batches_fall_back_on_uncapturable_stream(gpu_ctx):
    """a context that adopted the legacy null stream cannot capture: the narrow-graph batches must fall back to plain
    launches (same answer), not fail"""
    ctx = capi.Context(0)
    try:
        ctx.set_stream(0)                                       # hipStreamLegacy: stream capture is not allowed on it
        ctx.dp_set_option("graph_batch", 7)
        for seed in (61, 62):
            g = graphgen.random_levelized(8500 + seed, max_width=12, n_levels=120, R=5, p_colour=0.5)
            _dp_both(ctx, g)
            assert ctx.dp_run().key() == ctx.dp_run().key()
    finally:
        ctx.close()


@pytest.mark.timeout(240)
def test_dp_load_while_a_reservation_is_still_mapping():
    """Regression test of the round-3 hang (profiles/r03_load_hang_record.txt): dg_dp_prealloc starts a background thread that maps 8 GB
    lattice chunks; dg_dp_load_graph holds the pool's pause from its first allocation, the thread parks on it, and the lattice plan --
    still under the pause -- clears the pool (pool_trim joins the thread).  Before the fix that join never returned.  Deterministic: the
    loads follow the reservations at once, while chunks are still being mapped (6 chunks take ~0.6 s).  (a) a toy graph: the exact
    single-allocation plan clears the pool; (b) a graph whose plan wants another chunk size (segment_cells): the pool is cleared
    and re-requested; (c) back to the default plan with a reservation in flight.  Every load must return and every answer must
    equal the oracle's; run once, in a context of its own (fresh pool)."""
    ctx = capi.Context(0)
    try:
        toy = capi.DpGraphArrays.load(os.path.join(HERE, "golden", "toy1_k5w3_R2.dpg"))
        mid = graphgen.random_levelized(8777, max_width=40, n_levels=300, R=12, p_w1=0.3, p_colour=0.4)
        ref_toy, ref_mid = orc.dp_solve(toy), orc.dp_solve(mid)
        want = lambda r: (r["value"], r["s_het"], r["p1"], r["p2"])
        got = lambda o: (o.value, o.s_het, o.p1, o.p2)
        ctx.dp_prealloc(6 * (8 << 30))                          # the thread starts mapping ...
        assert got(ctx.dp_solve(toy)) == want(ref_toy)          # ... (a) and the toy's plan clears the pool under the load's pause
        ctx.dp_prealloc(6 * (8 << 30))
        ctx.dp_set_option("segment_cells", max(2, int(ref_mid["cells"]) // 5))
        assert got(ctx.dp_solve(mid)) == want(ref_mid)          # (b) other chunk size: clear + new request, checkpoint + recompute
        ctx.dp_set_option("segment_cells", 0)
        ctx.dp_prealloc(3 * (8 << 30))
        assert got(ctx.dp_solve(mid)) == want(ref_mid)          # (c)
        ctx.dp_prealloc(2 * (8 << 30))
        ctx.dp_load_graph(toy)                                  # load only, run later: the pool thread must be gone or parked harmlessly
        assert got(ctx.dp_run()) == want(ref_toy)
    finally:
        ctx.close()


@pytest.mark.parametrize("chunk_cells", [1, 3000, 150000])
def test_dp_chunked_lattice(gpu_ctx, chunk_cells):
    """the resident back-pointer lattice is a pool of chunks mapped by a background thread while the sweep runs;
    forced here with tiny chunks (one level per chunk at 1): results and level digests must not change, also when
    the same context is reused for graphs that need more / fewer chunks, and with a reservation made up front"""
    try:
        gpu_ctx.dp_set_option("lattice_chunk_cells", chunk_cells)
        gpu_ctx.dp_prealloc(chunk_cells * 4 * 3)
        for seed, kw in [(21, dict(max_width=14, n_levels=400, R=6)), (22, dict(max_width=45, n_levels=70, R=18, p_w1=0.5)),
                         (23, dict(n_levels=2, R=2)), (24, dict(max_width=8, n_levels=3000, R=3, p_colour=0.3)), (25, dict(R=33, max_width=25, n_levels=90)),
                         (26, dict(max_width=6, n_levels=40, R=4))]:
            _dp_both(gpu_ctx, graphgen.random_levelized(8100 + seed, **kw))
        _dp_both(gpu_ctx, capi.DpGraphArrays.load(os.path.join(HERE, "golden", "toy1_k5w3_R2.dpg")))
    finally:
        gpu_ctx.dp_set_option("lattice_chunk_cells", 1 << 31)


@pytest.mark.parametrize("k", [90, 255, 256, 300])
def test_dp_giant_indegree_uses_generic_path(gpu_ctx, k):
    """a vertex with in-degree > 64 (more than 64 haplotypes recombining into one vertex) takes the general task
    variant; beyond 255 the 8-bit in-edge ranks of the back-pointers overflow and the level keeps wide words on the
    generic kernel"""
    _dp_both(gpu_ctx, _fan_in_graph(k))


def test_dp_contexts_are_independent(gpu_ctx):
    """"a ctx is not thread-safe, different ctxs are independent" (include/dipgenie_hip.h): three contexts on one GPU,
    one host thread each, different graphs in flight at the same time -- the cohort mode bench.py measures"""
    import threading
    graphs = [graphgen.random_levelized(8300 + q, max_width=20 + 5 * q, n_levels=400 + 100 * q, R=6 + 6 * q, p_colour=0.4) for q in range(3)]
    refs = [orc.dp_solve(g) for g in graphs]
    ctxs = [capi.Context(0) for _ in graphs]
    got, errs = [None] * 3, []

    def work(q):
        try:
            for _ in range(4):
                got[q] = ctxs[q].dp_solve(graphs[q])
        except Exception as e:                                  # noqa: BLE001 - reported below
            errs.append((q, repr(e)))
    th = [threading.Thread(target=work, args=(q,)) for q in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in ctxs:
        c.close()
    assert not errs, errs
    for q in range(3):
        assert (got[q].value, got[q].s_het, got[q].p1, got[q].p2) == (refs[q]["value"], refs[q]["s_het"], refs[q]["p1"], refs[q]["p2"]), q


def test_dp_unreachable_sink(gpu_ctx):
    # every path to the sink needs 2 recombinations but R = 1: value stays NEG_INF, edge lists empty
    lo = np.array([0, 1, 2, 3], np.int32)
    g = capi.DpGraphArrays(1, level_off=lo, out_off=np.array([0, 1, 2, 2], np.int64), out_dst=np.array([1, 2], np.int32),
                           out_w=np.array([1, 1], np.uint8), hom_off=np.zeros(4, np.int64), hom_col=np.zeros(0, np.int32),
                           het_off=np.zeros(4, np.int64), het_col=np.zeros(0, np.int32))
    out = _dp_both(gpu_ctx, g)
    assert out.value == -(2 ** 31) // 4 and out.p1 == [] and out.p2 == []


def test_dp_rejects_bad_graphs(gpu_ctx):
    g = graphgen.random_levelized(5)
    bad = capi.DpGraphArrays(g.R, **{n: getattr(g, n).copy() for n in g.NAMES})
    bad.out_dst[0] = 0                                    # edge back to level 0
    with pytest.raises(capi.DgError, match="next level"):
        gpu_ctx.dp_solve(bad)
    bad = capi.DpGraphArrays(g.R, **{n: getattr(g, n).copy() for n in g.NAMES})
    bad.level_off[1] = 2                                  # two sources
    with pytest.raises(capi.DgError):
        gpu_ctx.dp_solve(bad)


def test_dp_pipeline_graphs_vs_oracle(gpu_ctx, built_cpu, tmp_path):
    """levelized graphs produced by the host pipeline for the committed e2e cases"""
    for name in ("bub_b", "bub_c", "bub_e", "bub_g"):
        c = CASES[name]
        pre = str(tmp_path / name)
        subprocess.run([built_cpu, "-q", "-t4"] + c["args"] + ["-g", os.path.join(ROOT, c["gfa"]), "-r", os.path.join(ROOT, c["reads"]),
                       "-o", pre + ".fa", "-D", pre], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = _dp_both(gpu_ctx, capi.DpGraphArrays.load(pre + ".dpg"))
        assert out.value == c["dp_value"] and len(out.p1) - 1 == c["r1"] and len(out.p2) - 1 == c["r2"]


# ------------------------------------------------------------------------------------- product CLI end to end
def _run_cli(cli, case, tmp_path, extra=()):
    out, js = tmp_path / "o.fa", tmp_path / "o.json"
    subprocess.run([cli, "-t8"] + case["args"] + ["-g", os.path.join(ROOT, case["gfa"]), "-r", os.path.join(ROOT, case["reads"]),
                   "-o", str(out), "-J", str(js), *extra], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out, "rb").read(), json.load(open(js))


def test_cli_reverse_strand_step_exits_1_without_fasta(built_hip, gpu_ctx, tmp_path):
    """a '<' step inside a forward walk: the reference leaves through exit(1) at solver.cpp:116-119 and writes no FASTA (golden: the
    reference binary's own behaviour, tests/golden/e2e.json gfa_reverse_step); so does the drop-in CLI"""
    c = CASES["gfa_reverse_step"]
    assert c["exit_code"] == 1 and not c["fasta_written"]
    out = tmp_path / "o.fa"
    p = subprocess.run([built_hip, "-t4"] + c["args"] + ["-g", os.path.join(ROOT, c["gfa"]), "-r", os.path.join(ROOT, c["reads"]), "-o", str(out)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert p.returncode == 1 and (not out.exists() or out.stat().st_size == 0)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if not c.get("slow") and not c["gfa"].startswith("<") and "exit_code" not in c])
def test_cli_e2e_small(built_hip, gpu_ctx, name, tmp_path):
    fa, summ = _run_cli(built_hip, CASES[name], tmp_path)
    assert hashlib.md5(fa).hexdigest() == CASES[name]["fasta_md5"]
    for key in ("dp_value", "r1", "r2", "spectrum"):
        if key in CASES[name]:
            assert summ[key] == CASES[name][key]


def test_cli_e2e_mhc4_diploid(built_hip, gpu_ctx, tmp_path):
    """BASELINE config 2 (with the available reads): byte-identical FASTA to the reference, and the DP
    lattice of the full-size graph re-checked level by level against the oracle."""
    c = CASES["mhc4_p2"]
    fa, summ = _run_cli(built_hip, c, tmp_path, extra=("-D", str(tmp_path / "mhc4")))
    assert hashlib.md5(fa).hexdigest() == c["fasta_md5"] == "46394489af8bc9026605ddf237aca4c7"
    assert (summ["dp_value"], summ["r1"], summ["r2"], summ["spectrum"]) == (60729, 17, 1, 138834)
    assert summ["cells"] == 421330909 and summ["relaxations"] == 659218148
    _dp_both(gpu_ctx, capi.DpGraphArrays.load(str(tmp_path / "mhc4.dpg")))


def test_cli_e2e_mhc24_synthetic(built_hip, gpu_ctx, tmp_path):
    """BASELINE config 3 (the bench workload): the seeded synthetic MHC-24; the reference needed 892.6 s for it"""
    c = CASES["mhc24_p2"]
    gfa, reads, _ = synth.ensure_mhc24(str(tmp_path / "mhc24"))
    case = dict(c, gfa=os.path.relpath(gfa, ROOT), reads=os.path.relpath(reads, ROOT))
    fa, summ = _run_cli(built_hip, case, tmp_path)
    assert hashlib.md5(fa).hexdigest() == c["fasta_md5"] == "cd13930ac90651b7e441506c1ecd4514"
    assert (summ["dp_value"], summ["r1"], summ["r2"]) == (331848, 10, 8)


def test_cli_e2e_mhc4_hg002_2x(built_hip, gpu_ctx, tmp_path):
    """BASELINE config 2 with diploid reads: seeded 2x reads from the two HG002 walks stand in for the missing
    test/HG002.mhc.2x.fq.gz; golden = the reference binary on the same files"""
    c = CASES["mhc4_hg002_2x"]
    gfa, reads = synth.ensure_mhc4_hg002(str(tmp_path / "hg002"))
    assert hashlib.md5(open(reads, "rb").read()).hexdigest() == c["reads_md5"]
    case = dict(c, gfa=os.path.relpath(gfa, ROOT), reads=os.path.relpath(reads, ROOT))
    fa, summ = _run_cli(built_hip, case, tmp_path)
    assert hashlib.md5(fa).hexdigest() == c["fasta_md5"] == "b56e7ea82ccd32ce24ec641a677c4c7e"
    assert (summ["dp_value"], summ["r1"], summ["r2"], summ["spectrum"]) == (181090, 9, 9, 387040)


def test_cli_e2e_mhc4_haploid(built_hip, gpu_ctx, tmp_path):
    c = CASES["mhc4_p1"]
    fa, summ = _run_cli(built_hip, c, tmp_path)
    assert hashlib.md5(fa).hexdigest() == c["fasta_md5"] == "0c4df87ded10634a36db0a2c90521ff0"


def test_cli_e2e_mhc4_haploid_device_tables(built_hip, gpu_ctx, tmp_path):
    """BASELINE configs[0] with the (vertex, r) tables forced onto the device (499 k vertices, 250 k dependent levels)"""
    c = CASES["mhc4_p1"]
    out = tmp_path / "o.fa"
    subprocess.run([built_hip, "-t8"] + c["args"] + ["-g", os.path.join(ROOT, c["gfa"]), "-r", os.path.join(ROOT, c["reads"]), "-o", str(out)],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, DG_HAPLOID="device"))
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == c["fasta_md5"] == "0c4df87ded10634a36db0a2c90521ff0"


def test_dp_full_size_idempotent(gpu_ctx, built_hip, tmp_path):
    """size-independent properties at full size: re-running on the resident graph is bit-identical and
    a wider recombination budget never lowers the optimum."""
    c = CASES["mhc4_p2"]
    _run_cli(built_hip, c, tmp_path, extra=("-D", str(tmp_path / "g")))
    g = capi.DpGraphArrays.load(str(tmp_path / "g.dpg"))
    gpu_ctx.dp_load_graph(g)
    a, b = gpu_ctx.dp_run(), gpu_ctx.dp_run()
    assert a.key() == b.key() and a.value == 60729
    g2 = capi.DpGraphArrays(g.R + 4, **{n: getattr(g, n) for n in g.NAMES})
    assert gpu_ctx.dp_solve(g2).value >= a.value
    g3 = capi.DpGraphArrays(4, **{n: getattr(g, n) for n in g.NAMES})
    assert gpu_ctx.dp_solve(g3).value <= a.value


# ------------------------------------------------------------------------------------- anchors on the device
ANCH = json.load(open(os.path.join(HERE, "golden", "anchors.json")))


@pytest.mark.parametrize("name", [n for n, a in ANCH.items() if "same_as" not in a and not CASES[n]["reads"].startswith("<")])
def test_device_anchor_hits_equal_reference(built_hip, gpu_ctx, name, tmp_path):
    """dg_anchor_* (vertex spans, dictionary join, shared-anchor filter with its decimal-string key order, occurrence
    sort) through the product CLI: the Anchor_hits + homo_bv dump must be the reference Solver object's, line by line
    (tests/golden/anchors.json); DG_HOST_ANCHORS=1 (the host join) must give the same file"""
    c, a = CASES[name], ANCH[name]
    for env_extra, tag in (({}, "dev"), ({"DG_HOST_ANCHORS": "1"}, "host")):
        dump = tmp_path / f"anchors_{tag}.txt"
        subprocess.run([built_hip, "-t8", "-p2", f"-k{a['k']}", f"-w{a['w']}", f"-T{a['T']}", "-g", os.path.join(ROOT, c["gfa"]), "-r", os.path.join(ROOT, c["reads"]),
                        "-o", str(tmp_path / "o.fa"), "-A", str(dump), "-X"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, **env_extra))
        txt = open(dump).read()
        lines = txt.splitlines()
        if "dump" in a:
            assert lines == a["dump"], tag
        assert hashlib.sha256(txt.encode()).hexdigest() == a["sha256"], tag


def test_device_anchor_hits_hg002_2x(built_hip, gpu_ctx, tmp_path):
    """the larger read set (seeded 2x HG002 reads on MHC_4: 225,856 occurrences after the filter)"""
    a = ANCH["mhc4_hg002_2x"]
    gfa, reads = synth.ensure_mhc4_hg002(str(tmp_path / "hg002"))
    dump = tmp_path / "anchors.txt"
    subprocess.run([built_hip, "-t8", "-p2", "-g", gfa, "-r", reads, "-o", str(tmp_path / "o.fa"), "-A", str(dump), "-X"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert hashlib.sha256(open(dump, "rb").read()).hexdigest() == a["sha256"]


# ------------------------------------------------------------------------------------- haploid DP on the device
@pytest.mark.parametrize("seed", range(12))
def test_haploid_dp_vs_oracle(gpu_ctx, seed):
    """dg_dp_solve_haploid against the literal scatter loop (approximator.cpp:44-72): every dp / back_vtx / back_r entry"""
    kw = [dict(), dict(n=400, R=18, span=20), dict(n=50, R=0), dict(n=300, R=6, p_zero_colour=1.0), dict(n=300, R=6, p_zero_colour=0.0, dup=0.5),
          dict(n=2000, R=3, span=2, avg_deg=1.2), dict(n=1, R=2), dict(n=2, R=1), dict(n=500, R=70, span=40, avg_deg=4.0),
          dict(n=800, R=18, p_w1=1.0), dict(n=800, R=18, p_w1=0.0), dict(n=5000, R=18, span=3)][seed]
    off, dst, w, ncol = graphgen.random_topological(9100 + seed, **kw)
    R = kw.get("R", 4)
    got = gpu_ctx.dp_solve_haploid(R, off, dst, w, ncol)
    want = orc.dp_haploid(R, off, dst, w, ncol)
    for name, a, b in zip(("dp", "back_vtx", "back_r"), got, want):
        assert np.array_equal(a, b), (name, seed)


def test_haploid_rejects_unordered_graph(gpu_ctx):
    off, dst, w, ncol = graphgen.random_topological(1, n=10)
    dst = dst.copy(); dst[0] = 0                                 # edge back to vertex 0
    with pytest.raises(capi.DgError, match="topological"):
        gpu_ctx.dp_solve_haploid(2, off, dst, w, ncol)


@pytest.mark.parametrize("name", ["toy2_p1", "toy1_p1", "bub_a_p1", "bub_c_p1"])
def test_cli_haploid_device_equals_host_tables(built_hip, gpu_ctx, name, tmp_path):
    """-p1 through the device (vertex, r) tables (DG_HAPLOID=device; by default the CLI picks by graph shape, and these
    narrow graphs go to the host gather loop) and through the host loop: same FASTA = the reference's (tests/golden/e2e.json)"""
    c = CASES[name]
    for env_extra in ({"DG_HAPLOID": "device"}, {"DG_HAPLOID": "host"}, {}):
        out = tmp_path / "o.fa"
        subprocess.run([built_hip, "-t4"] + c["args"] + ["-g", os.path.join(ROOT, c["gfa"]), "-r", os.path.join(ROOT, c["reads"]), "-o", str(out)],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **env_extra))
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == c["fasta_md5"]
