"""-m gpu: the drop-in boundary proven with the REFERENCE's own host code.  oracle/_ref/DipGenie_ref_hip is the reference's sources with
the patch of INTEGRATION.md s2.1-2.3 applied (oracle/ref_hip_patch.py, build container only: the patched sources never enter the
repository; the binary travels like oracle/_ref/DipGenie_ref) and linked against dipgenie_amd/csrc/libdipgenie_hip.so: gfa_read,
Solver::read_gfa, read_ip_reads, the anchor join / filter / fit, Approximator::solve's graph construction, topological reorder,
strict levelize, path -> sequence and the FASTA writer are the reference's own code; compute_hashes, index_kmers' window loop and
the diploid level loop run through the C ABI on the MI355X.  The FASTA must be the unmodified reference's (tests/golden/e2e.json).
Test infrastructure: nothing under dipgenie_amd/ or in bench.py's timed region touches this binary."""
import hashlib
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(ROOT, "oracle", "_ref", "DipGenie_ref_hip")
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))


@pytest.mark.parametrize("name", ["toy2_p2", "toy1_p2", "bub_c", "bub_g", "c5s", "mhc4_p2"])
def test_reference_host_code_with_hip_loops(name, built_hip, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/DipGenie_ref_hip absent (built by __graft_entry__.build() where /root/reference exists)")
    c = CASES[name]
    out = tmp_path / "o.fa"
    p = subprocess.run([EXE, "-t8", *c["args"], "-g", os.path.join(ROOT, c["gfa"]), "-r", os.path.join(ROOT, c["reads"]), "-o", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, HIP_FORCE_DEV_KERNARG="1"))
    txt = p.stdout.decode(errors="replace").replace("\r", "\n")
    assert p.returncode == 0, txt[-1500:] + p.stderr.decode(errors="replace")[-1500:]
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == c["fasta_md5"]
    assert int(re.search(r"DP value: (-?\d+)", txt).group(1)) == c["dp_value"]
    m = re.search(r"recombinations in P1: (-?\d+), recombinations in P2: (-?\d+)", txt)
    assert (int(m.group(1)), int(m.group(2))) == (c["r1"], c["r2"])
    ldd = subprocess.run(["ldd", EXE], stdout=subprocess.PIPE).stdout.decode()
    assert "libdipgenie_hip.so" in ldd and "liboracle" not in ldd
