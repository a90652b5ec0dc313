"""Seeded synthetic workloads for tests and bench.py.

* ``random_bubble_graph`` -- small random bubble-chain pangenomes + reads (parity fixtures).
* ``mosaic_panel``        -- the MHC-24 stand-in: the 5 real walks of tests/data/MHC_4.gfa.gz plus
  19 mosaic walks that switch founder at shared vertices (crossover every ~250 kb) and carry
  private single-base variants (one per ~2 kb), then 4x 150-bp reads from two of the mosaics.

All generators are deterministic given their seed and emit GFA 1.1 with forward-strand W-lines,
single source, acyclic -- the only shape the reference accepts.
"""
import gzip
import os

import numpy as np

_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(s: bytes) -> bytes:
    return s.translate(_COMP)[::-1]


def _rand_seq(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes())


def write_gfa(path, seg_seqs, links, walks, names=None):
    """seg_seqs: list[bytes] (segment i is named str(i+1)); links: iterable of (a, b) 0-based;
    walks: list of (sample, hap_index, [seg ids])."""
    with open(path, "wb") as f:
        f.write(b"H\tVN:Z:1.1\n")
        for i, s in enumerate(seg_seqs):
            f.write(b"S\t%d\t%s\n" % (i + 1, s))
        for a, b in links:
            f.write(b"L\t%d\t+\t%d\t+\t0M\n" % (a + 1, b + 1))
        for sample, hap, w in walks:
            body = b"".join(b">%d" % (v + 1) for v in w)
            n = sum(len(seg_seqs[v]) for v in w)
            f.write(b"W\t%s\t%d\t%s\t0\t%d\t%s\n" % (sample.encode(), hap, sample.encode(), n, body))


def write_fasta(path, reads, prefix="r"):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">%s%d\n%s\n" % (prefix.encode(), i, r))


def simulate_reads(rng, haps, n_reads, read_len=150, sub_rate=0.002):
    """Uniform reads from the given haplotype strings (round-robin), both strands, substitutions."""
    out = []
    alpha = np.frombuffer(b"ACGT", np.uint8)
    for i in range(n_reads):
        h = haps[i % len(haps)]
        if len(h) <= read_len:
            r = bytearray(h)
        else:
            p = int(rng.integers(0, len(h) - read_len + 1))
            r = bytearray(h[p:p + read_len])
        if sub_rate > 0:
            nerr = rng.binomial(len(r), sub_rate)
            for q in rng.integers(0, max(len(r), 1), nerr):
                r[q] = int(alpha[(np.searchsorted(alpha, r[q]) + 1 + rng.integers(0, 3)) % 4])
        r = bytes(r)
        if rng.integers(0, 2):
            r = revcomp(r)
        out.append(r)
    return out


def simulate_reads_array(rng, haps, n_reads, read_len=150, sub_rate=0.002):
    """Vectorised simulator for large read sets (BASELINE configs[3]: 30x = 10^6 reads): uniform positions on the given
    haplotypes (round-robin), substitutions at sub_rate, random strand.  Returns a uint8 array [n_reads, read_len].
    Same recipe as simulate_reads but another random stream, so no golden depends on it."""
    comp = np.arange(256, dtype=np.uint8)
    for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
        comp[a] = b
    alpha = np.frombuffer(b"ACGT", np.uint8)
    code = np.zeros(256, np.uint8)
    code[alpha] = np.arange(4, dtype=np.uint8)
    out = np.empty((n_reads, read_len), np.uint8)
    cols = np.arange(read_len, dtype=np.int64)
    for hi, h in enumerate(haps):
        rows = np.arange(hi, n_reads, len(haps))
        ha = np.frombuffer(h, np.uint8)
        pos = rng.integers(0, len(h) - read_len + 1, rows.size)
        for b0 in range(0, rows.size, 1 << 16):                      # blocks keep the index matrix small
            sl = slice(b0, min(b0 + (1 << 16), rows.size))
            out[rows[sl]] = ha[pos[sl, None] + cols[None, :]]
    nerr = int(rng.binomial(out.size, sub_rate))
    flat = out.reshape(-1)
    where = rng.integers(0, flat.size, nerr)
    flat[where] = alpha[(code[flat[where]] + 1 + rng.integers(0, 3, nerr)) % 4]
    flip = rng.integers(0, 2, n_reads).astype(bool)
    out[flip] = comp[out[flip][:, ::-1]]
    return out


def ensure_mhc4_hg002(cache_dir, base_gfa=None, seed=1, coverage=2.0, read_len=150, sub_rate=0.002, walks=("HG002",)):
    """BASELINE configs[1] stand-in for the missing test/HG002.mhc.2x.fq.gz: 2x reads (1x from each of
    the two HG002 walks of MHC_4.gfa.gz), seeded. Returns (gfa_path, reads_path)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base_gfa = base_gfa or os.path.join(root, "tests", "data", "MHC_4.gfa.gz")
    os.makedirs(cache_dir, exist_ok=True)
    reads = os.path.join(cache_dir, "mhc4_hg002_2x_seed%d.fa" % seed)
    if not os.path.exists(reads):
        _, seqs, _, wl = parse_gfa(base_gfa)
        haps = [b"".join(seqs[v] for v in w).upper() for (name, _, w) in wl if name in walks]
        assert len(haps) == 2, [n for (n, _, _) in wl]
        n_reads = int(coverage * sum(len(h) for h in haps) / 2 / read_len)
        rd = simulate_reads(np.random.default_rng(seed), haps, n_reads, read_len, sub_rate)
        tmp = reads + ".tmp%d" % os.getpid()
        write_fasta(tmp, rd)
        os.replace(tmp, reads)
    return base_gfa, reads


def ensure_mhc24_reads(cache_dir, coverage=30.0, seed=30, read_len=150, sample=(5, 6)):
    """Config-4 read set for the panel of ensure_mhc24: `coverage`x reads from the same two mosaic walks. Returns the path
    of a .npy uint8 matrix [n_reads, read_len] (memory-mapped by the ranks)."""
    gfa, _, _ = ensure_mhc24(cache_dir)
    path = os.path.join(cache_dir, "mhc24_%dx_seed%d.npy" % (int(coverage), seed))
    if not os.path.exists(path):
        _, seqs, _, walks = parse_gfa(gfa)
        haps = [b"".join(seqs[v] for v in walks[h][2]).upper() for h in sample]
        n_reads = int(coverage * sum(len(h) for h in haps) / 2 / read_len)
        arr = simulate_reads_array(np.random.default_rng(seed), haps, n_reads, read_len)
        tmp = path + ".tmp%d.npy" % os.getpid()
        np.save(tmp, arr)
        os.replace(tmp, path)
    return path


def random_bubble_graph(seed, n_bubbles=12, n_haps=4, seg_len=(20, 60), alleles=(2, 3), coverage=6.0,
                        read_len=80, sub_rate=0.0, sample_haps=(0, 1)):
    """A chain of bubbles: backbone segment, then `a` alternative allele segments, repeated.
    Returns (seg_seqs, links, walks, reads)."""
    rng = np.random.default_rng(seed)
    segs, links = [], []
    layers = []  # list of lists of segment ids
    for b in range(n_bubbles):
        bid = len(segs)
        segs.append(_rand_seq(rng, int(rng.integers(seg_len[0], seg_len[1] + 1))))
        layers.append([bid])
        a = int(rng.integers(alleles[0], alleles[1] + 1))
        base = _rand_seq(rng, int(rng.integers(1, 12)))
        ids = []
        for q in range(a):
            s = bytearray(base)
            if q:
                s[int(rng.integers(0, len(s)))] = int(rng.choice(np.frombuffer(b"ACGT", np.uint8)))
                if rng.integers(0, 3) == 0:
                    s += _rand_seq(rng, int(rng.integers(1, 6)))
            ids.append(len(segs))
            segs.append(bytes(s))
        layers.append(ids)
    tail = len(segs)
    segs.append(_rand_seq(rng, int(rng.integers(seg_len[0], seg_len[1] + 1))))
    layers.append([tail])
    for x, y in zip(layers[:-1], layers[1:]):
        for a in x:
            for b in y:
                links.append((a, b))
    walks = []
    for h in range(n_haps):
        w = [int(rng.choice(layer)) for layer in layers]
        walks.append(("hap%d" % h, h, w))
    hap_strs = [b"".join(segs[v] for v in walks[h][2]) for h in sample_haps]
    total = sum(len(s) for s in hap_strs)
    n_reads = max(2, int(coverage * total / read_len / 1.0))
    reads = simulate_reads(rng, hap_strs, n_reads, read_len, sub_rate)
    return segs, links, walks, reads


def linear_panel(seed, backbone_bp=200_000, n_haps=100, bubble_every=64, switch_bp=50_000, coverage=4.0, read_len=150,
                 sub_rate=0.002, sample=(0, 1)):
    """chr22-style stand-in (BASELINE configs[4], scaled by backbone_bp): uniform random backbone, one biallelic
    bubble every ~bubble_every bp (80 % SNP, 20 % 1-50 bp indel), n_haps walks from a copying model that switches
    template every ~switch_bp. Returns (seg_seqs, links, walks, reads)."""
    rng = np.random.default_rng(seed)
    segs, links, layers = [], [], []
    pos = 0
    while pos < backbone_bp:
        ln = int(rng.geometric(1.0 / bubble_every)) + 1
        layers.append([len(segs)])
        segs.append(_rand_seq(rng, ln))
        pos += ln
        if pos >= backbone_bp:
            break
        ref = _rand_seq(rng, 1)
        if rng.random() < 0.8:
            alt = bytes([b"ACGT"[(b"ACGT".index(ref) + 1 + int(rng.integers(0, 3))) % 4]])
        else:
            alt = ref + _rand_seq(rng, int(rng.integers(1, 51)))
        layers.append([len(segs), len(segs) + 1])
        segs += [ref, alt]
        pos += 1
    if len(layers[-1]) != 1:
        layers.append([len(segs)])
        segs.append(_rand_seq(rng, 40))
    for x, y in zip(layers[:-1], layers[1:]):
        for a in x:
            for b in y:
                links.append((a, b))
    # founders = 8 random allele vectors; haplotypes copy a founder and switch every ~switch_bp
    n_bub = sum(1 for l in layers if len(l) == 2)
    founders = rng.integers(0, 2, (8, n_bub))
    walks = []
    for h in range(n_haps):
        f = int(rng.integers(0, 8))
        acc, nxt, b = 0, int(rng.exponential(switch_bp)) + 100, 0
        w = []
        for l in layers:
            if len(l) == 1:
                v = l[0]
            else:
                v = l[int(founders[f, b])]
                b += 1
            w.append(v)
            acc += len(segs[v])
            if acc >= nxt:
                f = int(rng.integers(0, 8))
                nxt = acc + int(rng.exponential(switch_bp)) + 100
        walks.append(("hap%03d" % h, h, w))
    hap_strs = [b"".join(segs[v] for v in walks[h][2]) for h in sample]
    n_reads = int(coverage * sum(len(x) for x in hap_strs) / 2 / read_len)
    reads = simulate_reads(rng, hap_strs, n_reads, read_len, sub_rate)
    return segs, links, walks, reads


# ------------------------------------------------------------------------------------------------
def parse_gfa(path):
    """Minimal S/L/W reader (forward-strand graphs only). Returns (names, seqs, links, walks)."""
    op = gzip.open if path.endswith(".gz") else open
    names, seqs, idx, links, walks = [], [], {}, [], []
    with op(path, "rb") as f:
        for line in f:
            if line[:2] == b"S\t":
                _, n, s = line.rstrip(b"\r\n").split(b"\t")[:3]
                idx[n] = len(names)
                names.append(n)
                seqs.append(s)
            elif line[:2] == b"L\t":
                p = line.split(b"\t")
                links.append((idx[p[1]], idx[p[3]]))
            elif line[:2] == b"W\t":
                p = line.rstrip(b"\r\n").split(b"\t")
                w = [idx[x] for x in p[6].split(b">")[1:]]
                walks.append((p[1].decode(), int(p[2]), w))
    return names, seqs, links, walks


def mosaic_panel(base_gfa, out_gfa, out_reads, n_total=24, seed=24, switch_bp=250_000, variant_bp=2_000,
                 read_seed=4, coverage=4.0, read_len=150, sub_rate=0.002, sample=(5, 6)):
    """Build the MHC-24 stand-in from the 5-walk MHC_4 graph. Returns a dict of sizes."""
    names, seqs, links, walks = parse_gfa(base_gfa)
    rng = np.random.default_rng(seed)
    founders = [w for (_, _, w) in walks]
    lens = np.array([len(s) for s in seqs], np.int64)
    posmaps = [dict((v, i) for i, v in enumerate(w)) for w in founders]   # vertex -> index on the founder
    seqs = list(seqs)
    links = list(links)
    link_set = set(links)
    new_walks = list(walks)
    alpha = b"ACGT"
    for m in range(n_total - len(founders)):
        cur = int(rng.integers(0, len(founders)))
        i = 0
        out = []
        acc = 0
        next_switch = int(rng.exponential(switch_bp)) + 1000
        next_var = int(rng.exponential(variant_bp)) + 50
        while i < len(founders[cur]):
            v = founders[cur][i]
            out.append(v)
            acc += int(lens[v])
            i += 1
            if acc >= next_switch and i < len(founders[cur]):
                # switch to another founder that also contains the vertex just emitted
                cand = [f for f in range(len(founders)) if f != cur and v in posmaps[f]]
                if cand:
                    cur = int(rng.choice(cand))
                    i = posmaps[cur][v] + 1
                    next_switch = acc + int(rng.exponential(switch_bp)) + 1000
        # private single-base variants: a parallel copy S' of an interior segment with one base changed
        w2 = list(out)
        acc = 0
        for t in range(1, len(w2) - 1):
            v = w2[t]
            acc += int(lens[v])
            if acc >= next_var and 1 <= lens[v] <= 200:
                s = bytearray(seqs[v])
                q = int(rng.integers(0, len(s)))
                s[q] = alpha[(alpha.index(bytes([s[q]]).upper()) + 1 + int(rng.integers(0, 3))) % 4] if bytes([s[q]]).upper() in alpha else ord("A")
                nid = len(seqs)
                seqs.append(bytes(s))
                for e in ((w2[t - 1], nid), (nid, w2[t + 1])):
                    if e not in link_set:
                        link_set.add(e)
                        links.append(e)
                w2[t] = nid
                next_var = acc + int(rng.exponential(variant_bp)) + 50
        # segments created for this mosaic are private; later mosaics never traverse them
        new_walks.append(("MOSAIC%02d" % m, 1, w2))
        # all links used by the mosaic must exist (switches follow founder edges, so they do)
    lens = None
    write_gfa(out_gfa, seqs, links, new_walks)
    rrng = np.random.default_rng(read_seed)
    haps = [b"".join(seqs[v] for v in new_walks[h][2]).upper() for h in sample]
    n_reads = int(coverage * sum(len(h) for h in haps) / 2 / read_len)
    reads = simulate_reads(rrng, haps, n_reads, read_len, sub_rate)
    write_fasta(out_reads, reads)
    return dict(n_segments=len(seqs), n_links=len(links), n_walks=len(new_walks), n_reads=len(reads),
                hap_bp=[len(h) for h in haps])


def prefix_panel(gfa_in, out_gfa, out_reads, frac, read_seed=4, coverage=4.0, read_len=150, sub_rate=0.002, sample=(5, 6)):
    """The first `frac` of a panel as a panel of its own: every walk is cut at the first segment, about `frac` of the way
    along walk 0, that ALL walks traverse (so the cut graph keeps a single sink); segments and links beyond it are
    dropped, reads are re-simulated (same recipe as the full panel) from the cut sample haplotypes.  Used as the bounded
    sample of the bench workload for the reference binary.  Returns a dict of sizes."""
    names, seqs, links, walks = parse_gfa(gfa_in)
    sets = [set(w) for (_, _, w) in walks]
    w0 = walks[0][2]
    cut = None
    for t in range(max(1, int(frac * len(w0))), len(w0)):
        if all(w0[t] in st for st in sets):
            cut = w0[t]
            break
    if cut is None:
        raise ValueError("no segment shared by all walks after the requested fraction")
    new_walks = [(n, h, w[: w.index(cut) + 1]) for (n, h, w) in walks]
    keep = sorted(set(v for (_, _, w) in new_walks for v in w))
    remap = {v: i for i, v in enumerate(keep)}
    seqs2 = [seqs[v] for v in keep]
    links2 = [(remap[a], remap[b]) for (a, b) in links if a in remap and b in remap and a != cut]
    walks2 = [(n, h, [remap[v] for v in w]) for (n, h, w) in new_walks]
    write_gfa(out_gfa, seqs2, links2, walks2)
    rrng = np.random.default_rng(read_seed)
    haps = [b"".join(seqs2[v] for v in walks2[h][2]).upper() for h in sample]
    n_reads = int(coverage * sum(len(h) for h in haps) / 2 / read_len)
    reads = simulate_reads(rrng, haps, n_reads, read_len, sub_rate)
    write_fasta(out_reads, reads)
    return dict(n_segments=len(seqs2), n_links=len(links2), n_walks=len(walks2), n_reads=len(reads), hap_bp=[len(h) for h in haps],
                frac=frac)


def ensure_mhc24(cache_dir, base_gfa=None):
    """Generate (once) the config-3 workload under cache_dir; returns (gfa_path, reads_path, info)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base_gfa = base_gfa or os.path.join(root, "tests", "data", "MHC_4.gfa.gz")
    os.makedirs(cache_dir, exist_ok=True)
    gfa, reads, meta = (os.path.join(cache_dir, n) for n in ("mhc24.gfa", "mhc24_4x.fa", "mhc24.json"))
    if not (os.path.exists(gfa) and os.path.exists(reads) and os.path.exists(meta)):
        info = mosaic_panel(base_gfa, gfa, reads)
        with open(meta, "w") as f:
            json.dump(info, f)
    with open(meta) as f:
        return gfa, reads, json.load(f)
