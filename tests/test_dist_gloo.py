"""CPU, gloo, world 2 and 3: the PRODUCT class dipgenie_amd.dist_sketch.ShardedSketch -- hash-range partition, the
send-count matrix, the all-to-all of (hash, count) runs, the fused all-reduce, the id reconstruction -- driven through
torch.distributed with a CPU shim injected for the device operations (`CpuOps`: the oracle as the per-shard sketch,
numpy for the joins).  Everything must equal the single-process oracle result on the whole read set
(Sp_R keys / kmer_count / ids / Hist_kmer, solver.cpp:526-555, 711-755)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class CpuOps:
    """test shim for dist_sketch.HipOps: same methods, CPU tensors (uint64 hashes held as int64, like the device path)"""
    stream = None

    def __init__(self, orc):
        self.orc = orc

    @staticmethod
    def _u(t):
        return t.numpy().view(np.uint64)

    def sketch_reads(self, bases_t, off_t, k, w):
        b, off = bases_t.numpy().tobytes(), off_t.numpy()
        reads = [b[off[i]:off[i + 1]] for i in range(off.size - 1)]
        h, c = self.orc.sketch_reads(reads, k, w)
        return torch.from_numpy(h.view(np.int64).copy()), torch.from_numpy(c.copy())

    def count_dictionary(self, dict_t, h, c):
        d, hh = self._u(dict_t), self._u(h)
        out = np.zeros(d.size, np.int32)
        if hh.size:
            pos = np.searchsorted(hh, d)
            hit = (pos < hh.size) & (hh[np.minimum(pos, hh.size - 1)] == d)
            out[hit] = c.numpy()[pos[hit]]
        return torch.from_numpy(out)

    def partition(self, h, world):
        hh = self._u(h)
        owner = ((hh >> np.uint64(32)) * np.uint64(world)) >> np.uint64(32)
        return torch.from_numpy(np.searchsorted(owner, np.arange(world + 1), side="left").astype(np.int64))

    def merge_runs(self, h, c):
        hh, cc = self._u(h), c.numpy()
        order = np.argsort(hh, kind="stable")
        hh, cc = hh[order], cc[order]
        if hh.size == 0:
            return h[:0], c[:0]
        uh, idx = np.unique(hh, return_index=True)
        uc = np.add.reduceat(cc, idx).astype(np.int32)
        if uh[-1] == np.uint64(0xFFFFFFFFFFFFFFFF) and uc[-1] == 0:           # padding of the fixed-size exchange (as dg_sketch_merge_runs_dev)
            uh, uc = uh[:-1], uc[:-1]
        return torch.from_numpy(uh.view(np.int64).copy()), torch.from_numpy(uc.copy())

    def rank_dictionary(self, dict_t, h, rank1):
        d, hh, out = self._u(dict_t), self._u(h), rank1.numpy()
        if hh.size:
            pos = np.searchsorted(hh, d)
            hit = (pos < hh.size) & (hh[np.minimum(pos, hh.size - 1)] == d)
            out[hit] += pos[hit] + 1

    def histogram(self, c, hist):
        out = hist.numpy()
        out += np.bincount(np.minimum(c.numpy(), out.size - 1), minlength=out.size)


def reference(orc, reads, dict_hashes, k, w, n_bins):
    """single-process answer on the whole read set"""
    h, c = orc.sketch_reads(reads, k, w)
    pos = np.searchsorted(h, dict_hashes)
    hit = (pos < h.size) & (h[np.minimum(pos, max(h.size - 1, 0))] == dict_hashes) if h.size else np.zeros(dict_hashes.size, bool)
    counts = np.zeros(dict_hashes.size, np.int32)
    counts[hit] = c[pos[hit]]
    ids = np.full(dict_hashes.size, -1, np.int64)
    ids[hit] = pos[hit]
    hist = np.bincount(np.minimum(c, n_bins - 1), minlength=n_bins).astype(np.int64)
    return h, c, counts, ids, hist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def tensors_of(reads):
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    return torch.from_numpy(np.frombuffer(b"".join(reads) or b"\0", np.uint8).copy()), torch.from_numpy(off)


def _worker(rank, world, port, reads, dict_hashes, k, w, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import oracle_py as orc
    from dipgenie_amd import dist_sketch as ds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = ds.shard_bounds(len(reads), world, rank)
        bases_t, off_t = tensors_of(reads[lo:hi])
        dict_t = torch.from_numpy(dict_hashes.view(np.int64).copy())
        sk = ds.ShardedSketch(CpuOps(orc), "cpu")
        sc = sk.score(bases_t, off_t, dict_t, k, w)
        gh, gc = sk.gather_spectrum(sc)
        want_h, want_c, want_counts, want_ids, want_hist = reference(orc, reads, dict_hashes, k, w, ds.HIST_BINS)
        rh = sc.range_hash.numpy().view(np.uint64)
        ok = {
            "counts": np.array_equal(sc.counts.numpy(), want_counts),
            "ids": np.array_equal(sc.ids.numpy(), want_ids),
            "n_distinct": sc.n_distinct == want_h.size,
            "hist": np.array_equal(sc.hist.numpy(), want_hist),
            "spectrum": np.array_equal(gh.numpy().view(np.uint64), want_h) and np.array_equal(gc.numpy(), want_c),
            # the range really is this rank's slice of the global list: nothing replicated
            "range": np.array_equal(rh, want_h[sc.range_base: sc.range_base + rh.size]) and sum(sc.range_sizes) == want_h.size
                     and bool(np.all(ds.hash_owner(sc.range_hash, world).numpy() == rank)),
        }
        # second step of the same instance: the fixed-size exchange (cap fixed by the first step; no size exchange, no host read) --
        # every output identical; then a cap that is too small on purpose: the overflow flag reaches every rank, validated() repeats
        # the step with exact runs and re-calibrates
        assert sk.cap is not None and sk.cap >= 64
        sc2 = sk.score(bases_t, off_t, dict_t, k, w)

        def same(a, b):
            return (torch.equal(a.counts, b.counts) and torch.equal(a.ids, b.ids) and torch.equal(a.hist, b.hist) and torch.equal(a.range_hash, b.range_hash)
                    and torch.equal(a.range_count, b.range_count) and a.range_sizes == b.range_sizes and a.range_base == b.range_base)
        ok["fixed_size_step"] = same(sc, sc2) and not sc2.exchange_overflow and sk.validated(sc2, bases_t, off_t, dict_t, k, w) is sc2
        sk.cap = 4
        sc3 = sk.score(bases_t, off_t, dict_t, k, w)
        ok["overflow_seen_by_every_rank"] = sc3.exchange_overflow
        sc4 = sk.validated(sc3, bases_t, off_t, dict_t, k, w)
        ok["overflow_repaired"] = same(sc, sc4) and sk.cap >= 64
        q.put((rank, ok, hi - lo, int(rh.size)))
    finally:
        dist.destroy_process_group()


def _inputs(seed, n_reads):
    import oracle_py as orc
    rng = np.random.default_rng(seed)
    genome = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 20000).tobytes())
    reads = [genome[s:s + 100] for s in rng.integers(0, len(genome) - 100, n_reads)] + [b"", b"ACGT"]
    dh, _ = orc.minimizers(genome[:12000] + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 3000).tobytes()), 15, 8)
    return reads, np.unique(dh)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sketch_gloo(world):
    reads, dict_hashes = _inputs(3, 401)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, reads, dict_hashes, 15, 8, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, _, _ in res:
        assert all(ok.values()), (rank, ok)
    assert sum(n for _, _, n, _ in res) == len(reads)
    sizes = [n for _, _, _, n in res]
    assert min(sizes) > 0 and max(sizes) < 2 * (sum(sizes) / world)          # ranges are balanced (uniform hashes)


def test_sharded_sketch_single_rank():
    """world 1 (no process group): the same class, no collectives"""
    import oracle_py as orc
    from dipgenie_amd import dist_sketch as ds
    reads, dict_hashes = _inputs(5, 150)
    bases_t, off_t = tensors_of(reads)
    sk = ds.ShardedSketch(CpuOps(orc), "cpu")
    sc = sk.score(bases_t, off_t, torch.from_numpy(dict_hashes.view(np.int64).copy()), 15, 8)
    want_h, want_c, want_counts, want_ids, want_hist = reference(orc, reads, dict_hashes, 15, 8, ds.HIST_BINS)
    assert np.array_equal(sc.range_hash.numpy().view(np.uint64), want_h) and np.array_equal(sc.range_count.numpy(), want_c)
    assert np.array_equal(sc.counts.numpy(), want_counts) and np.array_equal(sc.ids.numpy(), want_ids)
    assert sc.n_distinct == want_h.size and np.array_equal(sc.hist.numpy(), want_hist) and sc.range_base == 0


def test_shard_bounds_cover():
    from dipgenie_amd.dist_sketch import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_hash_owner_is_unsigned():
    from dipgenie_amd.dist_sketch import hash_owner
    h = np.array([0, 1 << 31, (1 << 63) - 1, 1 << 63, (1 << 64) - 1], np.uint64)
    for w in (1, 2, 3, 8):
        got = hash_owner(torch.from_numpy(h.view(np.int64).copy()), w).numpy()
        want = ((h >> np.uint64(32)) * np.uint64(w)) >> np.uint64(32)
        assert np.array_equal(got, want.astype(np.int64)) and got.max() < w
