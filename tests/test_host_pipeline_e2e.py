"""CPU: the product's HOST pipeline (dipgenie_amd/host) with the oracle standing in for the two device
loops must reproduce the reference's outputs byte for byte (tests/golden/e2e.json = outputs of the
unmodified reference binary on the same inputs)."""
import hashlib
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))
FAST = [n for n, c in CASES.items() if not c.get("slow") and not c["gfa"].startswith("<")]


def run_case(binary, case, tmp_path, extra=()):
    out = tmp_path / "o.fa"
    js = tmp_path / "o.json"
    cmd = [binary, "-q", "-t4"] if "dg_host_oracle" in binary else [binary, "-t4"]
    cmd += case["args"] + ["-g", os.path.join(ROOT, case["gfa"]), "-r", os.path.join(ROOT, case["reads"]), "-o", str(out), "-J", str(js), *extra]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    fa = open(out, "rb").read()
    return fa, json.load(open(js))


def check(case, fa, summ):
    assert hashlib.md5(fa).hexdigest() == case["fasta_md5"]
    if "fasta" in case:
        assert fa.decode() == case["fasta"]
    for key in ("dp_value", "r1", "r2", "len1", "len2", "obj", "spectrum", "best_r_haploid"):
        if key in case and not (key == "len1" and "-p1" in case["args"]):
            assert summ[key] == case[key], key


@pytest.mark.parametrize("name", FAST)
def test_e2e_fast(name, built_cpu, tmp_path):
    fa, summ = run_case(built_cpu, CASES[name], tmp_path)
    check(CASES[name], fa, summ)


@pytest.mark.slow
def test_e2e_mhc4_diploid(built_cpu, tmp_path):
    # reference: FASTA md5 46394489..., DP value 60729, P1 17 / P2 1 recombinations
    fa, summ = run_case(built_cpu, CASES["mhc4_p2"], tmp_path)
    check(CASES["mhc4_p2"], fa, summ)
    assert summ["cells"] == 421330909 and summ["relaxations"] == 659218148 and summ["n_levels"] == 120363


@pytest.mark.skipif(not os.environ.get("DG_TEST_ALL"), reason="set DG_TEST_ALL=1 (adds ~35 s)")
def test_e2e_mhc4_haploid(built_cpu, tmp_path):
    fa, summ = run_case(built_cpu, CASES["mhc4_p1"], tmp_path)
    check(CASES["mhc4_p1"], fa, summ)
