"""CPU: Anchor_hits + homo_bv of the product's HOST pipeline (oracle sketches behind it) against the dump of the REAL
reference's Solver object (tests/golden/anchors.json, written by tests/golden/make_golden.py through
oracle/_ref/ref_harness `anchors`): pins the dictionary join, the shared-anchor filter with its string-key order, the
(front, back) occurrence sort and the HOM/HET labels (solver.cpp:415-446, 560-663, 745-879) occurrence by occurrence."""
import hashlib
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))
ANCH = json.load(open(os.path.join(HERE, "golden", "anchors.json")))
NAMES = [n for n, a in ANCH.items() if "same_as" not in a and not CASES[n]["reads"].startswith("<")]


def anchor_args(a):
    return [f"-k{a['k']}", f"-w{a['w']}", f"-T{a['T']}"]


def check_dump(path, a):
    txt = open(path).read()
    lines = txt.splitlines()
    occ = [l for l in lines if not l.startswith("homo")]
    assert (len(occ), len(lines) - len(occ)) == (a["n_occ"], a["n_homo"])
    if "dump" in a:
        assert lines == a["dump"]
    assert hashlib.sha256(txt.encode()).hexdigest() == a["sha256"]


@pytest.mark.parametrize("name", NAMES)
def test_host_anchor_hits_equal_reference(name, built_cpu, tmp_path):
    c, a = CASES[name], ANCH[name]
    dump = tmp_path / "anchors.txt"
    subprocess.run([built_cpu, "-q", "-t4", "-p2", *anchor_args(a), "-g", os.path.join(ROOT, c["gfa"]), "-r", os.path.join(ROOT, c["reads"]),
                    "-o", str(tmp_path / "o.fa"), "-A", str(dump), "-X"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    check_dump(dump, a)
