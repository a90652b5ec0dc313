"""BASELINE configs[3] as ONE run: "minimizer index+scoring sharded over 8 x MI355X with RCCL all-reduce, DP on GPU0".

    python -m dipgenie_amd.run_sharded --gpus N -g panel.gfa -r reads.fa -o out.fa [-t 16 -R 18 -k 31 -w 25 -T 1.0 -J summary.json]

The reference does all of this in one process (src/solver.cpp:449-887 then src/approximator.cpp:1014-1331).  Here one process per GPU
(torch.distributed; "nccl" == RCCL, "gloo" in the CPU tests), launched BEFORE anything touches a GPU:

  every rank   opens the GFA (host; dgr_open), sketches the haplotypes h = rank (mod world) on its GPU (index_kmers runs once per
               haplotype, independently: src/solver.cpp:470-473) and exchanges the minimizer lists: hashes to everybody (their
               sorted union is the dictionary D of the scoring stage), positions to rank 0;
  every rank   scores its contiguous block of the reads (dist_sketch.ShardedSketch: local HIP sketch, RCCL all-reduce of the
               dictionary hit vector, hash-range all-to-all, sharded merge, fused all-reduce) and the ranges of the exact global
               spectrum are gathered (Sp_R keys + kmer_count, src/solver.cpp:526-555, 711-732);
  rank 0       injects the haplotype sketches and the spectrum into the host pipeline (include/dipgenie_run.h) and runs the rest on
               ITS GPU: vertex spans, anchor join / filter / sort, fit + classify, graph, diploid DP, FASTA -- the output must be
               byte-identical to bin/DipGenie on the same inputs.

No CPU fallback: the scoring ops are dist_sketch.HipOps and the run library is libdipgenie_run.so (tests inject a shim and the
oracle-backed library).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
RUN_LIB = os.path.join(_HERE, "host", "libdipgenie_run.so")


class RunOptions(C.Structure):
    _fields_ = [("gfa_file", C.c_char_p), ("reads_file", C.c_char_p), ("out_file", C.c_char_p),
                ("threads", C.c_int32), ("ploidy", C.c_int32), ("R", C.c_int32), ("k", C.c_int32), ("w", C.c_int32),
                ("threshold", C.c_float), ("device", C.c_int32), ("quiet", C.c_int32)]


class RunSummary(C.Structure):
    _fields_ = [("dp_value", C.c_int32), ("s_het", C.c_int32), ("r1", C.c_int32), ("r2", C.c_int32), ("obj", C.c_int32),
                ("len1", C.c_int64), ("len2", C.c_int64), ("spectrum", C.c_int64), ("n_levels", C.c_int64), ("n_vertices", C.c_int64),
                ("cells", C.c_uint64), ("relaxations", C.c_uint64), ("seconds", C.c_double)]


class RunLib:
    """ctypes over include/dipgenie_run.h"""

    def __init__(self, path=RUN_LIB):
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `make -C dipgenie_amd/host` (no CPU fallback)")
        L = self.lib = C.CDLL(path)
        L.dgr_open.restype = C.c_void_p
        L.dgr_open.argtypes = [C.POINTER(RunOptions)]
        L.dgr_close.argtypes = [C.c_void_p]
        L.dgr_close.restype = None
        L.dgr_last_error.restype = C.c_char_p
        L.dgr_n_haplotypes.argtypes = [C.c_void_p]
        L.dgr_haplotype_sequence.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.dgr_load_reads.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.dgr_inject_haplotype_sketch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
        L.dgr_inject_spectrum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.dgr_solve.argtypes = [C.c_void_p, C.POINTER(RunSummary)]

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.dgr_last_error().decode()}")

    def open(self, **kw):
        o = RunOptions()
        for key, v in kw.items():
            setattr(o, key, v.encode() if isinstance(v, str) else v)
        self._opt = o                                    # keeps the strings alive
        h = self.lib.dgr_open(C.byref(o))
        if not h:
            raise RuntimeError(f"dgr_open failed: {self.lib.dgr_last_error().decode()}")
        return h

    def haplotype_sequence(self, h, idx):
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.dgr_haplotype_sequence(h, idx, C.byref(p), C.byref(n)), "dgr_haplotype_sequence")
        return C.string_at(p, n.value)

    def load_reads(self, h):
        """(bases uint8 [total], offsets int64 [n + 1]) as numpy views into the handle"""
        n, b, o = C.c_int64(), C.c_void_p(), C.c_void_p()
        self._check(self.lib.dgr_load_reads(h, C.byref(n), C.byref(b), C.byref(o)), "dgr_load_reads")
        off = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_int64)), (n.value + 1,))
        total = int(off[-1])
        bases = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), (max(total, 1),))[:total]
        return bases, off


def _find_free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pad_to(t, n, torch):
    out = torch.zeros(n, dtype=t.dtype, device=t.device)
    out[: t.numel()] = t
    return out


def run_rank(a, ops=None, lib_path=None, device=None):
    """one rank of the run; the process group is initialised (or absent: world 1).  ops: dist_sketch.HipOps (default, made here)
    or a test shim with the same methods plus sketch_haplotype(seq, k, w) -> (uint64 hashes, int64 positions) numpy arrays."""
    import torch
    import torch.distributed as dist
    from . import dist_sketch as ds
    t_start = time.perf_counter()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    gloo = dist.is_initialized() and dist.get_backend() == "gloo"
    if ops is None:
        from . import capi
        device = torch.device("cuda", a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
        ctx = capi.Context(device.index)
        ops = ds.HipOps(ctx, device)
        ops.sketch_haplotype = lambda seq, k, w: ctx.sketch_haplotype(seq, k, w)
    dev = device if device is not None else "cpu"
    dev_ordinal = device.index if isinstance(device, torch.device) and device.index is not None else 0
    coll = (lambda t: t.cpu()) if gloo else (lambda t: t)              # gloo moves host tensors, RCCL device tensors
    lib = RunLib(lib_path or RUN_LIB)
    H = lib.open(gfa_file=a.gfa, reads_file=a.reads, out_file=a.out, threads=a.threads, ploidy=2, R=a.R, k=a.k, w=a.w,
                 threshold=a.T, device=dev_ordinal, quiet=1)
    err = None
    summary = None
    try:
        summary = _run_rank_body(a, ops, lib, H, dev, world, rank, gloo, coll, t_start)
    except BaseException as e:                                          # noqa: BLE001 - reported to the peers below, then re-raised
        err = e
    finally:
        # The peers of a failed rank must not sit in a barrier until the backend's time-out (minutes for RCCL, 30 for gloo): every rank
        # contributes an ok flag.  This covers the common case -- rank 0 fails in its solve while the others are already waiting here;
        # a rank that fails INSIDE the exchange phase leaves through its launcher, which ends the group.
        all_ok = err is None
        if dist.is_initialized() and not isinstance(err, KeyboardInterrupt):
            try:
                flag = torch.tensor([1 if err is None else 0], dtype=torch.int32, device="cpu" if gloo else dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                all_ok = bool(int(flag.item()))
            except Exception:                                           # noqa: BLE001 - the group is already broken
                all_ok = False
        lib.lib.dgr_close(H)
    if err is not None:
        raise err
    if not all_ok:
        raise RuntimeError(f"rank {rank}: another rank of the sharded run failed")
    return summary


def _run_rank_body(a, ops, lib, H, dev, world, rank, gloo, coll, t_start):
    import torch
    import torch.distributed as dist
    from . import dist_sketch as ds
    laps = {"open_gfa": time.perf_counter() - t_start}
    t0 = time.perf_counter()
    # ---- haplotype side, sharded over the ranks (h mod world); lists exchanged as one int64 blob per rank ----
    nh = lib.lib.dgr_n_haplotypes(H)
    mine = {}
    for h in range(rank, nh, world):
        hh, pp = ops.sketch_haplotype(lib.haplotype_sequence(H, h), a.k, a.w)
        mine[h] = (np.ascontiguousarray(hh, np.uint64), np.ascontiguousarray(pp, np.int64))
    sizes = torch.zeros(nh, dtype=torch.int64)
    for h, (hh, _) in mine.items():
        sizes[h] = hh.size
    if world > 1:
        sizes_c = sizes.to(dev) if not gloo else sizes
        dist.all_reduce(sizes_c, op=dist.ReduceOp.SUM)
        sizes = sizes_c.cpu()
    sizes_l = sizes.tolist()
    per_rank = [sum(sizes_l[h] for h in range(r, nh, world)) for r in range(world)]
    cat = lambda q: (np.concatenate([mine[h][q] for h in sorted(mine)]) if mine else np.zeros(0, np.uint64 if q == 0 else np.int64))
    my_hash = torch.from_numpy(cat(0).view(np.int64).copy())
    my_pos = torch.from_numpy(cat(1).copy())
    if world > 1:
        m = max(max(per_rank), 1)
        send_h = coll(_pad_to(my_hash.to(dev), m, torch))
        all_h = torch.empty(world * m, dtype=torch.int64, device=send_h.device)
        dist.all_gather_into_tensor(all_h, send_h)                      # hashes: everybody needs the dictionary
        all_h = all_h.view(world, m)
        send_p = coll(_pad_to(my_pos.to(dev), m, torch))
        got_p = [torch.empty(m, dtype=torch.int64, device=send_p.device) for _ in range(world)] if rank == 0 else None
        dist.gather(send_p, got_p, dst=0)                               # positions: only the anchor stage (rank 0) needs them
        hashes_by_rank = [all_h[r, : per_rank[r]] for r in range(world)]
        pos_by_rank = [got_p[r][: per_rank[r]].cpu().numpy() for r in range(world)] if rank == 0 else None
    else:
        hashes_by_rank, pos_by_rank = [my_hash.to(dev)], [my_pos.numpy()]
    flat = torch.cat([x.to(dev) for x in hashes_by_rank]) if world > 1 else hashes_by_rank[0]
    dict_t, _ = ops.merge_runs(flat.contiguous(), torch.ones(flat.numel(), dtype=torch.int32, device=flat.device))   # sorted distinct (unsigned order)
    laps["haplotype_sketches+dictionary"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    # ---- reads, sharded: contiguous blocks ----
    bases, off = lib.load_reads(H)
    n_reads = off.size - 1
    lo, hi = ds.shard_bounds(n_reads, world, rank)
    b0, b1 = int(off[lo]), int(off[hi])
    bases_t = torch.from_numpy(np.array(bases[b0:b1]) if b1 > b0 else np.zeros(1, np.uint8)).to(dev)
    off_t = torch.from_numpy((off[lo: hi + 1] - off[lo]).astype(np.int64)).to(dev)
    laps["load_reads"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    sk = ds.ShardedSketch(ops, dev, force_exchange=bool(a.force_collectives))
    sc = sk.validated(sk.score(bases_t, off_t, dict_t, a.k, a.w), bases_t, off_t, dict_t, a.k, a.w)
    gh, gc = sk.gather_spectrum(sc)
    if hasattr(torch, "cuda") and str(dev) != "cpu":
        torch.cuda.synchronize()
    laps["score+gather_spectrum"] = time.perf_counter() - t0
    summary = None
    if rank == 0:
        t0 = time.perf_counter()
        # minimizer lists back in haplotype order: rank r holds h = r, r + world, ... concatenated
        for r in range(world):
            hr = hashes_by_rank[r].cpu().numpy().view(np.uint64)
            pr = pos_by_rank[r]
            o = 0
            for h in range(r, nh, world):
                n = sizes_l[h]
                hh, pp = np.ascontiguousarray(hr[o: o + n]), np.ascontiguousarray(pr[o: o + n])
                lib._check(lib.lib.dgr_inject_haplotype_sketch(H, h, hh.ctypes.data, pp.ctypes.data, n), "dgr_inject_haplotype_sketch")
                o += n
        sp_h = np.ascontiguousarray(gh.cpu().numpy().view(np.uint64))
        sp_c = np.ascontiguousarray(gc.cpu().numpy().astype(np.int32))
        hist = np.ascontiguousarray(sc.hist.cpu().numpy().astype(np.int64))
        assert sp_h.size == sc.n_distinct, "gathered spectrum does not have count_sp_r entries"
        lib._check(lib.lib.dgr_inject_spectrum(H, sp_h.ctypes.data, sp_c.ctypes.data, sp_h.size, hist.ctypes.data, hist.size), "dgr_inject_spectrum")
        laps["inject"] = time.perf_counter() - t0
        s = RunSummary()
        lib._check(lib.lib.dgr_solve(H, C.byref(s)), "dgr_solve")
        summary = {f: getattr(s, f) for f, _ in RunSummary._fields_}
        laps["solve(rank 0: anchors, fit, graph, DP, FASTA)"] = s.seconds
        summary.update(world=world, n_reads=int(n_reads), dictionary=int(dict_t.numel()), dictionary_hits=int((sc.counts > 0).sum().item()),
                       range_sizes=sc.range_sizes, stages_s=laps, wall_s=time.perf_counter() - t_start)
        if a.json:
            with open(a.json, "w") as f:
                json.dump(summary, f)
        if not a.quiet:
            print(json.dumps(summary), flush=True)
    return summary


def parse_args(argv=None):
    ap = argparse.ArgumentParser(prog="python -m dipgenie_amd.run_sharded", description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--force-collectives", action="store_true", help="world 1: still create the process group and take the collective path")
    ap.add_argument("--device", type=int, default=None, help="HIP device of this rank (default: LOCAL_RANK)")
    ap.add_argument("-g", dest="gfa", required=True)
    ap.add_argument("-r", dest="reads", required=True)
    ap.add_argument("-o", dest="out", required=True)
    ap.add_argument("-t", dest="threads", type=int, default=16)
    ap.add_argument("-R", dest="R", type=int, default=18)
    ap.add_argument("-k", dest="k", type=int, default=31)
    ap.add_argument("-w", dest="w", type=int, default=25)
    ap.add_argument("-T", dest="T", type=float, default=1.0)
    ap.add_argument("-J", dest="json", default=None)
    ap.add_argument("-q", dest="quiet", action="store_true")
    return ap.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    in_group = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not in_group and a.gpus > 1:
        # launcher: one process per GPU, started before anything here touches a GPU (children, not exec)
        # (--standalone: the launcher's own rendezvous picks and holds its port; a pre-probed free port could be taken by another job
        # between the probe and the bind)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "-m", "dipgenie_amd.run_sharded"] + (argv if argv is not None else sys.argv[1:])
        return subprocess.call(cmd)
    import torch.distributed as dist
    if in_group or a.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_find_free_port()))
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if a.backend == "nccl":
            import torch
            dev = torch.device("cuda", a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", "0")))
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run_rank(a)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
