"""GPU (-m gpu): BASELINE configs[3] as ONE run -- `python -m dipgenie_amd.run_sharded` with the product ops (HipOps, RCCL) and the
product run library (libdipgenie_run.so) in a world of one rank (one GPU per box: RCCL refuses two ranks on one device; the
multi-rank exchange is covered under gloo by tests/test_run_sharded_gloo.py).  The FASTA it writes must be byte-identical to
bin/DipGenie's on the same GFA + reads: the synthetic MHC-24 panel with the 30x read set (1,007,415 x 150 bp) and with the 4x set."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from dipgenie_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))


def _md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def _sharded(gfa, reads, out, js, extra=(), world=1):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    # one rank: RCCL with the collective path forced; more: every rank on device 0 with gloo (RCCL refuses two ranks on one device)
    mode = ["--gpus", "1", "--force-collectives", "--backend", "nccl"] if world == 1 else ["--gpus", str(world), "--backend", "gloo", "--device", "0"]
    cmd = [sys.executable, "-m", "dipgenie_amd.run_sharded", *mode, "-g", gfa, "-r", reads,
           "-o", out, "-J", js, "-t", "16", "-q", *extra]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return json.load(open(js))


def test_sharded_run_equals_cli_on_mhc24(built_hip, tmp_path_factory):
    cache = os.path.join(os.environ.get("DG_BENCH_CACHE", str(tmp_path_factory.mktemp("rs"))), "mhc24")
    gfa, reads4, _ = synth.ensure_mhc24(cache)
    tmp = tmp_path_factory.mktemp("rs_out")
    # 4x reads: the golden of the plain CLI run
    summ = _sharded(gfa, reads4, str(tmp / "s4.fa"), str(tmp / "s4.json"), ["-R", "18"])
    assert _md5(tmp / "s4.fa") == CASES["mhc24_p2"]["fasta_md5"] and summ["dp_value"] == CASES["mhc24_p2"]["dp_value"]
    assert summ["world"] == 1 and summ["spectrum"] == CASES["mhc24_p2"].get("spectrum", summ["spectrum"])
    # 30x reads (configs[3]): the plain CLI on the same FASTA is the reference point
    arr = np.load(synth.ensure_mhc24_reads(cache), mmap_mode="r")
    reads30 = str(tmp / "reads30.fa")
    with open(reads30, "wb") as f:
        n, rl = arr.shape
        head = np.frombuffer(b">r\n", np.uint8)
        block = np.empty((n, 3 + rl + 1), np.uint8)
        block[:, :3] = head; block[:, 3:3 + rl] = arr; block[:, -1] = ord("\n")
        f.write(block.tobytes())
    subprocess.run([built_hip, "-t", "16", "-p2", "-R18", "-g", gfa, "-r", reads30, "-o", str(tmp / "c30.fa"), "-J", str(tmp / "c30.json")], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cli = json.load(open(tmp / "c30.json"))
    summ = _sharded(gfa, reads30, str(tmp / "s30.fa"), str(tmp / "s30.json"), ["-R", "18"])
    assert _md5(tmp / "s30.fa") == _md5(tmp / "c30.fa")
    assert (summ["dp_value"], summ["spectrum"], summ["n_levels"], summ["cells"]) == (cli["dp_value"], cli["spectrum"], cli["n_levels"], cli["cells"])
    assert summ["n_reads"] == n and sum(summ["range_sizes"]) == summ["spectrum"]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_with_several_ranks_on_one_gpu(built_hip, tmp_path_factory, world):
    """the same job with the haplotype sketches and the reads really sharded over 2 / 3 processes, all on this box's one GPU (the HIP
    operations of every rank are the product's; gloo carries the exchange): the reference's FASTA"""
    cache = os.path.join(os.environ.get("DG_BENCH_CACHE", str(tmp_path_factory.mktemp("rs"))), "mhc24")
    gfa, reads4, _ = synth.ensure_mhc24(cache)
    tmp = tmp_path_factory.mktemp(f"rs_w{world}")
    summ = _sharded(gfa, reads4, str(tmp / "s.fa"), str(tmp / "s.json"), ["-R", "18"], world=world)
    assert _md5(tmp / "s.fa") == CASES["mhc24_p2"]["fasta_md5"] and summ["dp_value"] == CASES["mhc24_p2"]["dp_value"]
    assert summ["world"] == world and sum(summ["range_sizes"]) == summ["spectrum"]
