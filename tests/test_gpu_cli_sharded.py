"""-m gpu: BASELINE configs[3] as ONE C++ process -- bin/DipGenie --gpus N (dipgenie_amd/host/main.cpp run_sharded, dipgenie_amd/csrc/dg_shard.hip):
one host thread per rank, haplotype h sketched on rank h mod N, reads scored in contiguous blocks, dictionary hit vector all-reduced,
(hash, #reads) runs exchanged by hash range and merged by their owners, everything from the anchor join on (fit, graph, DP, FASTA) on
the first device.  This box has one GPU and RCCL refuses two ranks on one device, so:
  * RCCL (librccl, dlopen-ed by the library) is exercised in a world of ONE rank (--gpus 1 --shard-transport rccl: communicator,
    all-reduce of the hit vector, the grouped exchange degenerates to the own-range copy);
  * N = 2 and 3 run with the host-staged transport: N ranks = N threads with a context each on the one device, every device operation
    the product's, the collectives staged through host memory (the role gloo plays for the Python driver).
Every run must write the reference's FASTA (tests/golden/e2e.json); N > 1 under RCCL has never run (no multi-GPU lease)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from dipgenie_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))


def _run(cli, gfa, reads, args, out, mode):
    js = str(out) + ".json"
    p = subprocess.run([cli, "-t8", *args, *mode, "-g", gfa, "-r", reads, "-o", str(out), "-J", js], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), json.load(open(js)), p.stderr.decode(errors="replace")


@pytest.mark.parametrize("mode", [["--gpus", "1", "--shard-transport", "rccl"], ["--gpus", "2", "--shard-transport", "host"], ["--gpus", "3", "--shard-transport=host"]])
@pytest.mark.parametrize("name", ["toy1_p2", "bub_c", "bub_g", "c5s", "reads_with_N"])
def test_sharded_cli_small(built_hip, tmp_path, name, mode):
    c = CASES[name]
    md5, summ, _ = _run(built_hip, os.path.join(ROOT, c["gfa"]), os.path.join(ROOT, c["reads"]), c["args"], tmp_path / "o.fa", mode)
    assert md5 == c["fasta_md5"] and summ["dp_value"] == c["dp_value"] and (summ["r1"], summ["r2"]) == (c["r1"], c["r2"])
    assert summ["gpus"] == int(mode[1]) and summ["shard_transport"] == ("rccl" if "rccl" in mode[-1] else "host")
    if "spectrum" in c:
        assert summ["spectrum"] == c["spectrum"]


def test_sharded_cli_mhc24_and_config4(built_hip, tmp_path_factory):
    """the bench panel with its 4x reads (golden: the reference's FASTA) and with the 30x reads of BASELINE configs[3] (the plain CLI on the
    same file is the reference point), RCCL at one rank and host-staged at two and three"""
    cache = os.path.join(os.environ.get("DG_BENCH_CACHE", str(tmp_path_factory.mktemp("cs"))), "mhc24")
    gfa, reads4, _ = synth.ensure_mhc24(cache)
    tmp = tmp_path_factory.mktemp("cs_out")
    c = CASES["mhc24_p2"]
    hits = set()
    for q, mode in enumerate((["--gpus", "1", "--shard-transport", "rccl"], ["--gpus", "2", "--shard-transport", "host"], ["--gpus", "3", "--shard-transport", "host"])):
        md5, summ, _ = _run(built_hip, gfa, reads4, ["-p2", "-R18"], tmp / f"s4_{q}.fa", mode)
        assert md5 == c["fasta_md5"] and summ["dp_value"] == c["dp_value"]
        hits.add((summ["dictionary"], summ["dictionary_hits"], summ["spectrum"]))
    assert len(hits) == 1 and min(next(iter(hits))) > 0                  # the all-reduced hit vector does not depend on the sharding
    arr = np.load(synth.ensure_mhc24_reads(cache), mmap_mode="r")
    reads30 = str(tmp / "reads30.fa")
    n, rl = arr.shape
    block = np.empty((n, 3 + rl + 1), np.uint8)
    block[:, :3] = np.frombuffer(b">r\n", np.uint8); block[:, 3:3 + rl] = arr; block[:, -1] = ord("\n")
    open(reads30, "wb").write(block.tobytes())
    want, plain, _ = _run(built_hip, gfa, reads30, ["-p2", "-R18"], tmp / "c30.fa", [])
    for q, mode in enumerate((["--gpus", "1", "--shard-transport", "rccl"], ["--gpus", "3", "--shard-transport", "host"])):
        md5, summ, _ = _run(built_hip, gfa, reads30, ["-p2", "-R18"], tmp / f"s30_{q}.fa", mode)
        assert md5 == want and (summ["dp_value"], summ["spectrum"], summ["cells"]) == (plain["dp_value"], plain["spectrum"], plain["cells"])
