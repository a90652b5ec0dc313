"""CPU, world_size 2, gloo: the read-sharding / all-gather / merge / all-reduce glue of
dipgenie_amd.dist_sketch with the oracle as the per-shard sketch.  The merged spectrum and the
dictionary counts must equal the single-process oracle result on the whole read set."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, reads, dict_hashes, want_h, want_c, want_d, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import oracle_py as orc
    from dipgenie_amd import dist_sketch as ds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ds.shard_bounds(len(reads), world, rank)
    h, c = orc.sketch_reads(reads[lo:hi], 15, 8)
    ht, ct = torch.from_numpy(h.view(np.int64).copy()), torch.from_numpy(c.copy())
    hh, cc = ds.allgather_runs(ht, ct, "cpu")
    uh, uc = ds.merge_runs_torch(hh, cc)
    ok = np.array_equal(uh, want_h) and np.array_equal(uc, want_c)
    # dictionary counts + all-reduce(sum)
    counts = torch.zeros(len(dict_hashes), dtype=torch.int32)
    pos = np.searchsorted(h, dict_hashes)
    hit = (pos < h.size) & (h[np.minimum(pos, max(h.size - 1, 0))] == dict_hashes) if h.size else np.zeros(len(dict_hashes), bool)
    counts[torch.from_numpy(np.nonzero(hit)[0])] = torch.from_numpy(c[pos[hit]])
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    ok = ok and np.array_equal(counts.numpy(), want_d)
    q.put((rank, bool(ok), hi - lo))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sketch_gloo(world):
    import oracle_py as orc
    rng = np.random.default_rng(3)
    genome = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 20000).tobytes())
    reads = [genome[s:s + 100] for s in rng.integers(0, len(genome) - 100, 401)] + [b"", b"ACGT"]
    want_h, want_c = orc.sketch_reads(reads, 15, 8)
    dh, _ = orc.minimizers(genome, 15, 8)
    dict_hashes = np.unique(dh)
    pos = np.searchsorted(want_h, dict_hashes)
    hit = (pos < want_h.size) & (want_h[np.minimum(pos, want_h.size - 1)] == dict_hashes)
    want_d = np.zeros(dict_hashes.size, np.int32)
    want_d[hit] = want_c[pos[hit]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, reads, dict_hashes, want_h, want_c, want_d, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert sum(n for _, _, n in res) == len(reads)


def test_shard_bounds_cover():
    from dipgenie_amd.dist_sketch import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
