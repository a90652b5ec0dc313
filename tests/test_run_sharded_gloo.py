"""CPU, gloo, world 1 / 2 / 3: BASELINE configs[3] as ONE run -- dipgenie_amd.run_sharded (the product module: haplotype sketches
sharded over the ranks, reads scored by dist_sketch.ShardedSketch, spectrum + sketches injected into the host pipeline on rank 0)
-- with the oracle standing in for every device loop (CpuOps shim for the scoring ops, tests/harness/libdg_run_oracle.so for the run
library).  The FASTA rank 0 writes must be the reference's (tests/golden/e2e.json), i.e. what the single-process run gives."""
import hashlib
import json
import os
import socket
import types

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASES = json.load(open(os.path.join(HERE, "golden", "e2e.json")))
RUN_LIB = os.path.join(HERE, "harness", "libdg_run_oracle.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _args(case, out, js):
    a = dict(gfa=os.path.join(ROOT, case["gfa"]), reads=os.path.join(ROOT, case["reads"]), out=out, threads=2, R=18, k=31, w=25, T=1.0, json=js,
             quiet=True, force_collectives=False, device=None)
    for x in case["args"]:
        if x.startswith("-R"): a["R"] = int(x[2:])
        if x.startswith("-k"): a["k"] = int(x[2:])
        if x.startswith("-w"): a["w"] = int(x[2:])
        if x.startswith("-T"): a["T"] = float(x[2:])
    return types.SimpleNamespace(**a)


def _worker(rank, world, port, a, q):
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import oracle_py as orc
    from test_dist_gloo import CpuOps
    from dipgenie_amd import run_sharded as rs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ops = CpuOps(orc)
        ops.sketch_haplotype = lambda seq, k, w: orc.minimizers(seq, k, w)
        summ = rs.run_rank(a, ops=ops, lib_path=RUN_LIB, device="cpu")
        q.put((rank, summ))
    except Exception as e:                                      # noqa: BLE001 - reported to the parent
        q.put((rank, repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("bub_a", 1), ("bub_a", 2), ("bub_e", 3), ("bub_h", 2), ("toy1_p2", 3)])
def test_sharded_run_gives_the_reference_fasta(name, world, built_cpu, tmp_path):
    assert os.path.exists(RUN_LIB), "tests/harness/libdg_run_oracle.so missing (make -C tests/harness)"
    case = CASES[name]
    out, js = str(tmp_path / "o.fa"), str(tmp_path / "o.json")
    a = _args(case, out, js)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, a, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(not isinstance(v, str) for v in res.values()), res
    summ = res[0]
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == case["fasta_md5"]
    for key in ("dp_value", "r1", "r2", "len1", "len2", "obj", "spectrum"):
        if key in case:
            assert summ[key] == case[key], key
    assert summ["world"] == world and sum(summ["range_sizes"]) == summ["spectrum"]
    assert json.load(open(js))["dp_value"] == summ["dp_value"]
