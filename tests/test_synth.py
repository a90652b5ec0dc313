"""CPU: the seeded workload generators are deterministic (goldens and bench inputs depend on them)."""
import hashlib
import json
import os

import numpy as np

from dipgenie_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))


def test_hg002_read_set_is_the_golden_one(tmp_path):
    """BASELINE configs[1] stand-in: the read set the reference golden (mhc4_hg002_2x) was produced on"""
    gfa, reads = synth.ensure_mhc4_hg002(str(tmp_path))
    assert os.path.basename(gfa) == "MHC_4.gfa.gz"
    data = open(reads, "rb").read()
    assert hashlib.md5(data).hexdigest() == CASES["mhc4_hg002_2x"]["reads_md5"]
    assert data.count(b">") == 66608 and all(len(x) == 150 for x in data.split(b"\n")[1:200:2])
    assert synth.ensure_mhc4_hg002(str(tmp_path))[1] == reads           # cached, not regenerated


def test_simulate_reads_array_is_seeded_and_plausible():
    """config-4 read simulator: same seed -> same matrix; reads are substrings of a haplotype (or of its reverse
    complement) up to the substitution rate"""
    rng = np.random.default_rng(7)
    haps = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 20000)) for _ in range(2)]
    a = synth.simulate_reads_array(np.random.default_rng(30), haps, 4000, 150, 0.002)
    b = synth.simulate_reads_array(np.random.default_rng(30), haps, 4000, 150, 0.002)
    assert a.shape == (4000, 150) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert set(np.unique(a).tolist()) <= set(b"ACGT")
    exact = flipped = 0
    for i in range(0, 4000, 10):
        r, h = a[i].tobytes(), haps[i % 2]
        if r in h:
            exact += 1
        elif synth.revcomp(r) in h:
            flipped += 1
    assert exact > 80 and flipped > 80 and exact + flipped > 250       # P(no substitution in 150 bp) = 0.74
    c = synth.simulate_reads_array(np.random.default_rng(31), haps, 4000, 150, 0.0)
    assert not np.array_equal(a, c)
    for i in range(0, 4000, 97):                                         # error-free: every read maps exactly
        r = c[i].tobytes()
        assert r in haps[i % 2] or synth.revcomp(r) in haps[i % 2]
