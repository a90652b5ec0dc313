"""Read-sharded minimizer scoring across the GPUs of one node (one process per GPU, torch.distributed).

north_star: "the per-read minimizer extraction + graph-hit counting shards naturally over reads across the 8 GPUs of one
node with RCCL all-reduce of per-haplotype hit vectors over xGMI".  Semantics restated: Solver::compute_hashes over all
reads, Sp_R / ids / kmer_count / Hist_kmer (/root/reference/src/solver.cpp:526-555, 711-755), sharded as SURVEY.md s8e:

  local     every rank sketches its contiguous block of reads on its GPU -> sorted distinct (hash, #reads) run;
  (1) dictionary path   the run is joined against the sorted haplotype-minimizer dictionary D (M hashes, resident on
      every rank) into count[M]; ALL-REDUCE(sum) -- issued asynchronously, it overlaps the exchange below;
  (2) spectrum path     the uint64 hash space is cut into `world` equal ranges (MurmurHash3 outputs are uniform), rank r
      owns range r: one ALL-GATHER of the world x world send-count matrix (the only size exchange, one host read), one
      ALL-TO-ALL of (hash, count) runs, local merge (radix sort + reduce-by-key) -> the rank's range of the exact global
      spectrum.  Nothing is replicated: the merge shrinks with the number of ranks;
  (3) one fused ALL-REDUCE(sum) of [distinct count per range | multiplicity histogram | id + 1 of every dictionary hash
      inside its owner's range]: every rank then knows count_sp_r, Hist_kmer and -- adding the exclusive scan of the
      per-range counts, the owner being a function of the hash -- the global Sp_R id of every dictionary hit (the only
      ids the DP stage consumes).

torch is plumbing here: device memory, one stream, the collectives ("nccl" == RCCL on ROCm; "gloo" in the CPU tests).
Every device operation goes through an `ops` object: `HipOps` (the C ABI of libdipgenie_hip.so on CUDA tensors -- the
product; there is no CPU implementation in this package) or a shim injected by the tests.
"""
import ctypes as C
from dataclasses import dataclass

import torch
import torch.distributed as dist

HIST_BINS = 4096            # multiplicities >= HIST_BINS - 1 share the last bin


def shard_bounds(n_reads, world, rank):
    """contiguous block of reads for `rank` (blocks differ by at most one read)"""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def hash_owner(hash_i64, world):
    """owner range of uint64 hashes held as int64 tensors: floor(hi32 * world / 2^32) (dg_sketch.hip: hash_owner)"""
    hi = (hash_i64 >> 32) & 0xFFFFFFFF
    return (hi * world) >> 32


class HipOps:
    """The device operations of the sharded path on CUDA tensors, through the C ABI.  The ctx adopts `stream` (a
    torch.cuda.Stream): library kernels, torch kernels and the collectives are then ordered by that one stream, no host
    synchronisation is needed for hand-offs (the entry points that return a size still wait for it)."""

    def __init__(self, ctx, device, stream=None):
        from . import capi
        self.capi, self.ctx, self.device = capi, ctx, device
        self.stream = stream if stream is not None else torch.cuda.Stream(device)
        ctx.set_stream(self.stream.cuda_stream)

    def sketch_reads(self, bases_t, off_t, k, w):
        """bases_t uint8 [n_bases] and off_t int64 [n+1] device tensors of THIS rank's reads -> (hash int64, count int32)"""
        n_reads = off_t.numel() - 1
        cap = max(int(bases_t.numel()), 1)
        h = torch.empty(cap, dtype=torch.int64, device=self.device)
        c = torch.empty(cap, dtype=torch.int32, device=self.device)
        nd = C.c_int64()
        self.capi._check(self.capi.lib.dg_sketch_reads_dev(self.ctx.h, bases_t.data_ptr(), off_t.data_ptr(), n_reads, bases_t.numel(), k, w,
                                                           h.data_ptr(), c.data_ptr(), cap, C.byref(nd)), "dg_sketch_reads_dev")
        return h[: nd.value], c[: nd.value]

    def count_dictionary(self, dict_t, h, c):
        counts = torch.zeros(dict_t.numel(), dtype=torch.int32, device=self.device)
        self.capi._check(self.capi.lib.dg_sketch_count_dictionary_dev(self.ctx.h, dict_t.data_ptr(), dict_t.numel(), h.data_ptr(), c.data_ptr(),
                                                                      h.numel(), counts.data_ptr()), "dg_sketch_count_dictionary_dev")
        return counts

    def partition(self, h, world):
        split = torch.empty(world + 1, dtype=torch.int64, device=self.device)
        self.capi._check(self.capi.lib.dg_sketch_partition_dev(self.ctx.h, h.data_ptr(), h.numel(), world, split.data_ptr()), "dg_sketch_partition_dev")
        return split

    def merge_runs(self, h, c):
        oh, oc = torch.empty_like(h), torch.empty_like(c)
        n = C.c_int64()
        self.capi._check(self.capi.lib.dg_sketch_merge_runs_dev(self.ctx.h, h.data_ptr(), c.data_ptr(), h.numel(), oh.data_ptr(), oc.data_ptr(),
                                                                oh.numel(), C.byref(n)), "dg_sketch_merge_runs_dev")
        return oh[: n.value], oc[: n.value]

    def rank_dictionary(self, dict_t, h, rank1):
        self.capi._check(self.capi.lib.dg_sketch_rank_dictionary_dev(self.ctx.h, dict_t.data_ptr(), dict_t.numel(), h.data_ptr(), h.numel(), 0,
                                                                     rank1.data_ptr()), "dg_sketch_rank_dictionary_dev")

    def count_rank_dictionary(self, dict_t, h, c, rank1):
        """count_dictionary + rank_dictionary against the same list (one rank): one binary search per dictionary hash"""
        counts = torch.zeros(dict_t.numel(), dtype=torch.int32, device=self.device)
        self.capi._check(self.capi.lib.dg_sketch_count_rank_dictionary_dev(self.ctx.h, dict_t.data_ptr(), dict_t.numel(), h.data_ptr(), c.data_ptr(),
                                                                           h.numel(), 0, counts.data_ptr(), rank1.data_ptr()), "dg_sketch_count_rank_dictionary_dev")
        return counts

    def histogram(self, c, hist):
        self.capi._check(self.capi.lib.dg_sketch_histogram_dev(self.ctx.h, c.data_ptr(), c.numel(), hist.numel(), hist.data_ptr()), "dg_sketch_histogram_dev")


class Score:
    """what the scoring stage hands to the DP stage, identical on every rank except the sharded range.  The sizes live on the device
    (sizes_t); n_distinct / range_base / range_sizes / exchange_overflow read them on first use -- outside the scoring step."""

    def __init__(self, counts, ids, hist, range_hash, range_count, rank, sizes_t=None, sizes=None, overflow_t=None):
        self.counts = counts            # int32 [M]  #reads containing dictionary hash i (kmer_count restricted to D)
        self.ids = ids                  # int64 [M]  global Sp_R id of dictionary hash i, -1 if no read has it
        self.hist = hist                # int64 [HIST_BINS]  Hist_kmer: multiplicity -> #distinct hashes
        self.range_hash = range_hash    # int64 [n_r]  this rank's hash range of the global spectrum, sorted
        self.range_count = range_count  # int32 [n_r]
        self._rank, self._sizes_t, self._sizes, self._overflow_t = rank, sizes_t, sizes, overflow_t

    @property
    def range_sizes(self):              # n_r of every rank
        if self._sizes is None:
            self._sizes = [int(x) for x in self._sizes_t.cpu().tolist()]
        return self._sizes

    @property
    def n_distinct(self):               # count_sp_r: distinct read-minimizer hashes over all reads
        return int(sum(self.range_sizes))

    @property
    def range_base(self):               # global id of range_hash[0]
        return int(sum(self.range_sizes[: self._rank]))

    @property
    def exchange_overflow(self):        # a rank had more pairs for one owner than the fixed-size exchange carries: the step must be repeated (ShardedSketch.validated)
        return bool(self._overflow_t is not None and int(self._overflow_t.item()) != 0)


class ShardedSketch:
    """One instance per rank.  `ops`: HipOps (product) or a test shim with the same methods."""

    def __init__(self, ops, device, force_exchange=False):
        """force_exchange: take the collective path even in a world of one rank (tools/rccl_rehearsal.py: every RCCL call
        of the sharded path exercised on a single GPU)"""
        self.ops, self.device, self.force_exchange = ops, device, force_exchange
        self.laps = None                                 # tools/rccl_rehearsal.py: dict of per-stage seconds (synchronising, so not for timed runs)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.gloo = dist.is_initialized() and dist.get_backend() == "gloo"
        self.stream = getattr(ops, "stream", None)
        # Fixed-size exchange: the first step of a read set exchanges exact runs (all-gather of the world x world send counts + host read
        # + all-to-all with uneven splits) and fixes `cap` = pairs per (sender, owner) slot, the same on every rank because every rank
        # sees the same matrix; later steps send [world, 1 + cap, 2] blocks with the run length in row 0 -- no size exchange, no host read.
        self.cap = None
        self.pad_slack = 1.25

    # gloo (CPU tests, one-GPU rehearsals) moves data through host tensors; RCCL works on the device tensors
    def _c(self, t):
        return t.cpu() if self.gloo and t.is_cuda else t

    def _back(self, t):
        return t.to(self.device) if t.device != torch.device(self.device) else t

    def score(self, bases_t, off_t, dict_t, k, w):
        """bases_t uint8 [n_bases], off_t int64 [n+1]: THIS rank's reads; dict_t int64 [M]: the sorted dictionary.
        Runs on the ops' stream, ordered after whatever the caller has enqueued on its current stream and before
        whatever it enqueues next."""
        if self.stream is None:
            return self._score(bases_t, off_t, dict_t, k, w)
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            out = self._score(bases_t, off_t, dict_t, k, w)
        caller.wait_stream(self.stream)
        return out

    def _dictionary_ranges(self, dict_t):
        """[lo, hi) = the slice of the sorted dictionary inside this rank's hash range, and the owner of every dictionary
        hash; a function of (dictionary, world) only: computed once and kept"""
        # keyed on the tensor OBJECT (kept alive here, so its address cannot be handed to another dictionary) and its version
        # counter (in-place edits): an address + length key would silently serve a stale slice to an equally long new tensor
        if getattr(self, "_dict_ref", None) is not dict_t or self._dict_ver != (dict_t._version, self.world):
            split = self.ops.partition(dict_t, self.world).cpu().tolist()
            self._dict_cache = (int(split[self.rank]), int(split[self.rank + 1]), hash_owner(dict_t, self.world))
            self._dict_ref, self._dict_ver = dict_t, (dict_t._version, self.world)
        return self._dict_cache

    def _lap(self, name):
        if self.laps is not None:
            import time
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.laps[name] = self.laps.get(name, 0.0) + now - self._t
            self._t = now

    def _score(self, bases_t, off_t, dict_t, k, w):
        ops, W, me = self.ops, self.world, self.rank
        if self.laps is not None:
            import time
            torch.cuda.synchronize(); self._t = time.perf_counter()
        h, c = ops.sketch_reads(bases_t, off_t, k, w)
        self._lap("sketch")
        multi = W > 1 or (self.force_exchange and dist.is_initialized())
        if not multi and hasattr(ops, "count_rank_dictionary"):
            # one rank: the local spectrum is the global one -- dictionary counts and ids from one pass, no size exchange
            M = dict_t.numel()
            tail = torch.zeros(HIST_BINS + M, dtype=torch.int64, device=self.device)
            counts = ops.count_rank_dictionary(dict_t, h, c, tail[HIST_BINS:])
            ops.histogram(c, tail[:HIST_BINS])
            self._lap("count + rank dictionary, histogram")
            return Score(counts, tail[HIST_BINS:] - 1, tail[:HIST_BINS], h, c, 0, sizes=[int(h.numel())])
        counts = ops.count_dictionary(dict_t, h, c)
        self._lap("count_dictionary")
        work = None
        if multi:                                        # (1) hit vector: asynchronous, overlaps the exchange
            counts_c = self._c(counts)
            work = dist.all_reduce(counts_c, op=dist.ReduceOp.SUM, async_op=True)
            self._lap("all_reduce counts (issue)")
        overflow = None
        if not multi:
            rh, rc = h, c
        elif self.cap is not None:                       # (2) hash-range exchange, fixed-size blocks: nothing read by the host
            split = ops.partition(h, W)
            self._lap("partition")
            cap, n = self.cap, h.numel()
            seg = split[1:] - split[:-1]                                             # pairs for every owner
            col = torch.arange(cap, device=self.device)
            valid = col[None, :] < seg[:, None]
            src = (split[:-1, None] + col[None, :]).clamp_(max=max(n - 1, 0))
            block = torch.empty((W, cap + 1, 2), dtype=torch.int64, device=self.device)
            block[:, 0, 0] = torch.where(seg <= cap, seg, torch.full_like(seg, -1))   # -1: this run does not fit (see Score.exchange_overflow)
            block[:, 0, 1] = 0
            if n > 0:
                block[:, 1:, 0] = torch.where(valid, h[src], torch.full((1, 1), -1, dtype=torch.int64, device=self.device))   # padding: hash 0xFFFF...F,
                block[:, 1:, 1] = torch.where(valid, c[src].to(torch.int64), torch.zeros((1, 1), dtype=torch.int64, device=self.device))   # count 0
            else:
                block[:, 1:, 0] = -1; block[:, 1:, 1] = 0
            block = self._c(block)
            recv = torch.empty_like(block)
            self._lap("pack payload")
            dist.all_to_all_single(recv, block)                                      # equal splits: [1 + cap, 2] from every rank
            recv = self._back(recv)
            self._lap("all_to_all")
            overflow = (recv[:, 0, 0] < 0).any().to(torch.int64)
            body = recv[:, 1:, :].reshape(-1, 2)
            rh, rc = ops.merge_runs(body[:, 0].contiguous(), body[:, 1].to(torch.int32).contiguous())   # the padding sorts last and is dropped there
            self._lap("unpack + merge_runs")
        else:                                            # (2) hash-range exchange, exact runs (first step of a read set: fixes `cap`)
            split = ops.partition(h, W)
            self._lap("partition")
            send = self._c((split[1:] - split[:-1]).contiguous())
            mat = torch.empty(W * W, dtype=torch.int64, device=send.device)
            dist.all_gather_into_tensor(mat, send)       # world x world send-count matrix: the one size exchange
            mat = mat.view(W, W).cpu()                   # the one host read of the exchange
            self.cap = max(64, -(-int(self.pad_slack * max(int(mat.max()), 1)) // 4096) * 4096)   # 1.25 x the largest run, rounded up to 4,096 pairs; the same on every rank (same matrix)
            in_splits, out_splits = mat[me].tolist(), mat[:, me].tolist()
            self._lap("all_gather counts + host read")
            payload = self._c(torch.stack([h, c.to(torch.int64)], dim=1).contiguous())     # [n, 2]: one collective for both columns
            recv = torch.empty((sum(out_splits), 2), dtype=torch.int64, device=payload.device)
            self._lap("pack payload")
            dist.all_to_all_single(recv, payload, output_split_sizes=out_splits, input_split_sizes=in_splits)
            recv = self._back(recv)
            self._lap("all_to_all")
            rh, rc = ops.merge_runs(recv[:, 0].contiguous(), recv[:, 1].to(torch.int32).contiguous())
            self._lap("unpack + merge_runs")
        # (3) per-range distinct counts | multiplicity histogram | dictionary ids, one fused all-reduce
        M = dict_t.numel()
        tail = torch.zeros(W + HIST_BINS + M + 1, dtype=torch.int64, device=self.device)   # (last word: overflow flag of the fixed-size exchange)
        tail[me] = rh.numel()
        if overflow is not None:
            tail[-1] = overflow
        ops.histogram(rc, tail[W: W + HIST_BINS])
        lo, hi, owner_t = self._dictionary_ranges(dict_t)  # only the dictionary hashes inside this rank's range can be in rh
        ops.rank_dictionary(dict_t[lo:hi], rh, tail[W + HIST_BINS + lo: W + HIST_BINS + hi])
        self._lap("histogram + rank_dictionary")
        if multi:
            tail_c = self._c(tail)
            dist.all_reduce(tail_c, op=dist.ReduceOp.SUM)
            tail = self._back(tail_c)
            work.wait()
            counts = self._back(counts_c)
            self._lap("fused all_reduce + wait")
        rank1 = tail[W + HIST_BINS: W + HIST_BINS + M]
        if multi:
            sizes = tail[:W]
            base = torch.cumsum(sizes, 0) - sizes        # exclusive scan: first global id of every range -- consumed on the device
            ids = torch.where(rank1 > 0, rank1 - 1 + base[owner_t], torch.full_like(rank1, -1))
            self._lap("ids")
            return Score(counts, ids, tail[W: W + HIST_BINS], rh, rc, me, sizes_t=sizes, overflow_t=tail[-1])
        self._lap("ids")                                 # one range: its size is known here, ids are the local ranks
        return Score(counts, rank1 - 1, tail[W: W + HIST_BINS], rh, rc, me, sizes=[int(rh.numel())])

    def validated(self, sc, bases_t, off_t, dict_t, k, w):
        """the one host check of a fixed-size step, OUTSIDE the step: if some rank had more pairs for an owner than `cap` carries (the
        read set changed under a calibrated instance), every rank sees the flag (it travelled in the fused all-reduce), drops `cap`
        and scores again with exact runs, which also re-calibrates"""
        if not sc.exchange_overflow:
            return sc
        self.cap = None
        return self.score(bases_t, off_t, dict_t, k, w)

    def gather_spectrum(self, sc):
        """the whole global spectrum on every rank (sorted distinct hashes, #reads): ranges concatenated in rank order.
        Not needed by the DP stage; for callers that want Sp_R replicated (and for the tests)."""
        if self.world == 1:
            return sc.range_hash, sc.range_count
        m = max(max(sc.range_sizes), 1)
        pad = torch.zeros((m, 2), dtype=torch.int64, device=self.device)
        pad[: sc.range_hash.numel(), 0] = sc.range_hash
        pad[: sc.range_hash.numel(), 1] = sc.range_count.to(torch.int64)
        pad = self._c(pad)
        out = torch.empty((self.world * m, 2), dtype=torch.int64, device=pad.device)
        dist.all_gather_into_tensor(out, pad)
        out = self._back(out).view(self.world, m, 2)
        parts = [out[r, : sc.range_sizes[r]] for r in range(self.world)]
        cat = torch.cat(parts)
        return cat[:, 0].contiguous(), cat[:, 1].to(torch.int32).contiguous()
