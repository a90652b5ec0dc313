"""Read-sharded minimizer scoring across the GPUs of one node (one process per GPU, torch.distributed).

north_star: "the per-read minimizer extraction + graph-hit counting shards naturally over reads across
the 8 GPUs of one node with RCCL all-reduce of per-haplotype hit vectors over xGMI".  Each rank sketches
its contiguous block of reads with the HIP kernels (C ABI, device-resident buffers), then

  1. dictionary path: every rank holds the sorted haplotype-minimizer dictionary D (M hashes); the local
     (hash, #reads) list is joined against D into count[M] (int32) and ALL-REDUCED (sum) -- the only
     collective the DP needs (anchor multiplicities);
  2. spectrum path: the local sorted distinct (hash, count) runs are ALL-GATHERED (padded to the max
     length) and merged (radix sort + reduce-by-key on device) so that every rank holds the exact global
     Sp_R / kmer_count (needed for anchor ranks and the multiplicity histogram).

torch is used for device memory, streams and the collectives only ("nccl" == RCCL on ROCm; "gloo" on
CPU for the tests, where `local_sketch` is injected by the test).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_reads, world, rank):
    """contiguous block of reads for `rank` (blocks differ by at most one read)"""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _u64_as_i64(t):
    return t.view(torch.int64)


def _coll_device(device):
    """gloo (CPU tests / one-GPU rehearsals) moves data through host tensors; RCCL works on the device tensors"""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def allgather_runs(hash_t, count_t, device):
    """all-gather variable-length (hash uint64-as-int64, count int32) runs; returns concatenated tensors."""
    out_device = device
    device = _coll_device(device)
    hash_t, count_t = hash_t.to(device), count_t.to(device)
    world = dist.get_world_size()
    n = torch.tensor([hash_t.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    hp = torch.zeros(m, dtype=torch.int64, device=device)
    cp = torch.zeros(m, dtype=torch.int32, device=device)
    hp[: hash_t.numel()] = hash_t
    cp[: count_t.numel()] = count_t
    hs = [torch.empty_like(hp) for _ in range(world)]
    cs = [torch.empty_like(cp) for _ in range(world)]
    dist.all_gather(hs, hp)
    dist.all_gather(cs, cp)
    return torch.cat([h[:s] for h, s in zip(hs, sizes)]).to(out_device), torch.cat([c[:s] for c, s in zip(cs, sizes)]).to(out_device)


def merge_runs_torch(hash_i64, count_i32):
    """CPU/gloo reference merge (tests): exact unsigned order via numpy."""
    h = hash_i64.cpu().numpy().view(np.uint64)
    c = count_i32.cpu().numpy()
    order = np.argsort(h, kind="stable")
    h, c = h[order], c[order]
    uh, idx = np.unique(h, return_index=True)
    return uh, np.add.reduceat(c, idx).astype(np.int32) if h.size else c


class ShardedSketch:
    """One instance per rank. `ctx` is a dipgenie_amd.capi.Context bound to this rank's GPU."""

    def __init__(self, ctx, device):
        self.ctx, self.device = ctx, device

    def local(self, bases_t, off_t, k, w):
        """bases_t uint8 [n_bases] and off_t int64 [n+1] device tensors of THIS rank's reads."""
        import ctypes as C
        from . import capi
        n_reads = off_t.numel() - 1
        cap = max(int(bases_t.numel()), 1)
        h = torch.empty(cap, dtype=torch.int64, device=self.device)
        c = torch.empty(cap, dtype=torch.int32, device=self.device)
        nd = C.c_int64()
        capi._check(capi.lib.dg_sketch_reads_dev(self.ctx.h, bases_t.data_ptr(), off_t.data_ptr(), n_reads, bases_t.numel(), k, w,
                                                 h.data_ptr(), c.data_ptr(), cap, C.byref(nd)), "dg_sketch_reads_dev")
        return h[: nd.value], c[: nd.value]

    def dictionary_counts(self, dict_t, h, c):
        """count[M] for the local shard, then all-reduce(sum) over ranks (RCCL)."""
        from . import capi
        counts = torch.zeros(dict_t.numel(), dtype=torch.int32, device=self.device)
        if counts.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()   # the zero fill is on torch's stream, the join on the ctx stream
        capi._check(capi.lib.dg_sketch_count_dictionary_dev(self.ctx.h, dict_t.data_ptr(), dict_t.numel(), h.data_ptr(), c.data_ptr(),
                                                            h.numel(), counts.data_ptr()), "dg_sketch_count_dictionary_dev")
        self.ctx_sync()
        if dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "gloo":
                host = counts.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                counts.copy_(host)
            else:
                dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        return counts

    def global_spectrum(self, h, c):
        """exact global (sorted distinct hash, #reads) on every rank"""
        import ctypes as C
        from . import capi
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return h, c
        hh, cc = allgather_runs(h, c, self.device)
        if hh.is_cuda:
            torch.cuda.synchronize(self.device)         # the collective ran on torch's stream, the merge runs on the ctx stream
        oh = torch.empty_like(hh)
        oc = torch.empty_like(cc)
        n = C.c_int64()
        capi._check(capi.lib.dg_sketch_merge_runs_dev(self.ctx.h, hh.data_ptr(), cc.data_ptr(), hh.numel(), oh.data_ptr(), oc.data_ptr(),
                                                      oh.numel(), C.byref(n)), "dg_sketch_merge_runs_dev")
        return oh[: n.value], oc[: n.value]

    def ctx_sync(self):
        from . import capi
        capi._check(capi.lib.dg_synchronize(self.ctx.h), "dg_synchronize")
