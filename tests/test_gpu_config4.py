"""GPU (-m gpu): BASELINE configs[3] at full size -- the seeded 30x read set (1,007,415 x 150 bp) on the synthetic MHC-24
panel with its haplotype-minimizer dictionary -- through the product's scoring class (dipgenie_amd.dist_sketch) and the
device entry points of its sharded path.  The oracle would need minutes for the whole set, so full size is pinned by
size-independent properties (order / strand invariance, count doubling, shard-and-merge == one shot) plus oracle equality
on a sample; the multi-rank collectives themselves are covered by tests/test_dist_gloo.py on the same class."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle_py as orc
from dipgenie_amd import capi, synth
from dipgenie_amd.dist_sketch import HIST_BINS, HipOps, ShardedSketch, shard_bounds

pytestmark = pytest.mark.gpu
K, W = 31, 25
_COMP = np.zeros(256, np.uint8)
_COMP[:] = np.arange(256)
for a, b in zip(b"ACGT", b"TGCA"):
    _COMP[a] = b


@pytest.fixture(scope="module")
def cfg4(tmp_path_factory, gpu_ctx):
    cache_root = os.environ.get("DG_BENCH_CACHE", str(tmp_path_factory.mktemp("cfg4")))
    cache = os.path.join(cache_root, "mhc24")
    gfa, _, _ = synth.ensure_mhc24(cache)
    arr = np.load(synth.ensure_mhc24_reads(cache), mmap_mode="r")
    assert arr.shape == (1007415, 150)
    _, seqs, _, walks = synth.parse_gfa(gfa)
    D = np.unique(np.concatenate([gpu_ctx.sketch_haplotype(b"".join(seqs[v] for v in wv), K, W)[0] for (_, _, wv) in walks]))
    ctx = capi.Context(0)
    dev = torch.device("cuda", 0)
    ops = HipOps(ctx, dev)
    yield dict(cache_root=cache_root, arr=np.array(arr), D=D, ops=ops, dev=dev, dict_t=torch.from_numpy(D.view(np.int64).copy()).to(dev))
    ctx.close()


def _dev(arr2d, dev):
    n, rl = arr2d.shape
    return torch.from_numpy(np.ascontiguousarray(arr2d).reshape(-1)).to(dev), (torch.arange(n + 1, dtype=torch.int64) * rl).to(dev)


def _join(h, c, D):
    pos = np.searchsorted(h, D)
    hit = (pos < h.size) & (h[np.minimum(pos, h.size - 1)] == D)
    counts = np.zeros(D.size, np.int32); counts[hit] = c[pos[hit]]
    ids = np.full(D.size, -1, np.int64); ids[hit] = pos[hit]
    return counts, ids


def test_config4_scoring_class_equals_single_call(cfg4, gpu_ctx):
    """ShardedSketch at world 1 (device-resident path: dg_sketch_reads_dev + dictionary join + ids + histogram) against
    dg_sketch_reads on host buffers and a numpy join, on the whole 30x set"""
    arr, D, dev = cfg4["arr"], cfg4["D"], cfg4["dev"]
    n, rl = arr.shape
    h1, c1 = gpu_ctx.sketch_reads_flat(arr.tobytes(), np.arange(n + 1, dtype=np.int64) * rl, K, W)
    assert np.all(np.diff(h1) > 0)
    sk = ShardedSketch(cfg4["ops"], dev)
    b, o = _dev(arr, dev)
    sc = sk.score(b, o, cfg4["dict_t"], K, W)
    torch.cuda.synchronize()
    assert np.array_equal(sc.range_hash.cpu().numpy().view(np.uint64), h1) and np.array_equal(sc.range_count.cpu().numpy(), c1)
    counts, ids = _join(h1, c1, D)
    assert np.array_equal(sc.counts.cpu().numpy(), counts) and np.array_equal(sc.ids.cpu().numpy(), ids)
    assert sc.n_distinct == h1.size and sc.range_base == 0
    assert np.array_equal(sc.hist.cpu().numpy(), np.bincount(np.minimum(c1, HIST_BINS - 1), minlength=HIST_BINS))
    assert int((counts > 0).sum()) > 100000                      # the reads do hit the panel's dictionary


def test_config4_full_size_properties(cfg4):
    """read order and strand do not matter; the doubled read set doubles every count; a sample equals the oracle"""
    with torch.cuda.stream(cfg4["ops"].stream):                  # direct HipOps calls: everything on the ops' stream
        _full_size_properties(cfg4)


def _full_size_properties(cfg4):
    arr, dev, ops = cfg4["arr"], cfg4["dev"], cfg4["ops"]
    rng = np.random.default_rng(4)
    h1, c1 = (t.cpu().numpy() for t in ops.sketch_reads(*_dev(arr, dev), K, W))
    perm = rng.permutation(arr.shape[0])
    arr2 = arr[perm]
    flip = (np.arange(arr2.shape[0]) & 1) == 1
    arr2[flip] = _COMP[arr2[flip][:, ::-1]]                      # every other read as its reverse complement
    h2, c2 = (t.cpu().numpy() for t in ops.sketch_reads(*_dev(arr2, dev), K, W))
    assert np.array_equal(h1, h2) and np.array_equal(c1, c2)
    h3, c3 = (t.cpu().numpy() for t in ops.sketch_reads(*_dev(np.concatenate([arr, arr2]), dev), K, W))
    assert np.array_equal(h1, h3) and np.array_equal(2 * c1, c3)
    sub = arr[rng.choice(arr.shape[0], 4000, replace=False)]
    hs, cs = (t.cpu().numpy() for t in ops.sketch_reads(*_dev(sub, dev), K, W))
    ho, co = orc.sketch_reads([bytes(r) for r in sub], K, W)
    assert np.array_equal(hs.view(np.uint64), ho) and np.array_equal(cs, co)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_config4_shard_exchange_merge_on_one_gpu(cfg4, world):
    """the device side of the sharded path at full size, ranks emulated one after the other on this GPU: per-shard
    sketch, hash-range partition (dg_sketch_partition_dev), per-owner merge (dg_sketch_merge_runs_dev), dictionary ids
    (dg_sketch_rank_dictionary_dev) and histogram (dg_sketch_histogram_dev) must reassemble the one-shot result"""
    with torch.cuda.stream(cfg4["ops"].stream):
        _shard_exchange_merge(cfg4, world)


def _shard_exchange_merge(cfg4, world):
    arr, D, dev, ops, dict_t = cfg4["arr"], cfg4["D"], cfg4["dev"], cfg4["ops"], cfg4["dict_t"]
    n = arr.shape[0]
    h1, c1 = (t.cpu().numpy() for t in ops.sketch_reads(*_dev(arr, dev), K, W))
    h1 = h1.view(np.uint64)
    runs = []
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        h, c = ops.sketch_reads(*_dev(arr[lo:hi], dev), K, W)
        split = ops.partition(h, world).cpu().numpy()
        assert split[0] == 0 and split[-1] == h.numel() and np.all(np.diff(split) >= 0)
        owner = (((h.cpu().numpy().view(np.uint64) >> np.uint64(32)) * np.uint64(world)) >> np.uint64(32)).astype(np.int64)
        assert np.array_equal(split, np.searchsorted(owner, np.arange(world + 1)))
        runs.append((h.clone(), c.clone(), split))
    base, rank1, hist = 0, torch.zeros(D.size, dtype=torch.int64, device=dev), torch.zeros(HIST_BINS, dtype=torch.int64, device=dev)
    bases = []
    for owner in range(world):                                   # what rank `owner` receives from the all-to-all
        hh = torch.cat([h[s[owner]:s[owner + 1]] for h, _, s in runs])
        cc = torch.cat([c[s[owner]:s[owner + 1]] for _, c, s in runs])
        rh, rc = ops.merge_runs(hh, cc)
        assert np.array_equal(rh.cpu().numpy().view(np.uint64), h1[base: base + rh.numel()])
        assert np.array_equal(rc.cpu().numpy(), c1[base: base + rh.numel()])
        ops.rank_dictionary(dict_t, rh, rank1)
        ops.histogram(rc, hist)
        bases.append(base)
        base += rh.numel()
    assert base == h1.size
    torch.cuda.synchronize()
    from dipgenie_amd.dist_sketch import hash_owner
    r1 = rank1.cpu()
    ids = torch.where(r1 > 0, r1 - 1 + torch.tensor(bases)[hash_owner(dict_t.cpu(), world)], torch.full_like(r1, -1)).numpy()
    assert np.array_equal(ids, _join(h1, c1, D)[1])
    assert np.array_equal(hist.cpu().numpy(), np.bincount(np.minimum(c1, HIST_BINS - 1), minlength=HIST_BINS))


def test_config4_collective_path_under_rccl_world_of_one(cfg4):
    """every RCCL call of the sharded class (async all-reduce, all-gather of the send counts, all-to-all of the runs, fused
    all-reduce) on this GPU: torch.distributed "nccl" backend in a world of one rank, the class forced through its
    collective path, in a child process (tools/rccl_rehearsal.py asserts equality with the collective-free pass)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0", DG_BENCH_CACHE=cfg4["cache_root"])
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_rehearsal.py"), "300000"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"RCCL world-1 rehearsal OK: 300000 reads" in p.stdout, p.stdout.decode()[-2000:]
