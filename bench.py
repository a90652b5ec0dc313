#!/usr/bin/env python3
"""bench.py -- DipGenie diploid hot path on MI355X (see DESIGN.md s6 for what every field means).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

N = 1 (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic MHC-24 -- 24 walks over the MHC_4
segment set (5 real + 19 seeded mosaics with private variants), diploid -p2 -R18, 4x synthetic 150-bp reads.  A *step* is
one pass of the hot path over that input, every input already resident in HBM: (a) minimizer scoring of the read set
(HIP sketch, dictionary join, ids, histogram), then (b) the pair-of-paths DP (delta precompute, level sweep, traceback).
`value` = DP state cells / s.  The same line carries `sketch_config4`: the 30x read set (configs[3]) scored on this one
rank -- the N = 1 point of the scaling curve below -- and, measured by child processes after this process has released its device
memory: `hip_runtime.native_runtime_check` (the DP passes again from bin/dg_dp_bench, plain C++ over the C ABI on the HIP runtime the
library was built against), `config4_one_run` (configs[3] as one job: the plain CLI, bin/DipGenie --gpus 1 over RCCL and
python -m dipgenie_amd.run_sharded on the 30x reads as a FASTA), `config5` (configs[4] at a tenth of its size through the CLI: value-pass
cells/s, recompute factor, s8d fraction and counter bytes) and `cpu_baseline` (the reference binary on a prefix panel).

N > 1 (BASELINE.json configs[3]): what shards is the minimizer scoring -- north_star: "throughput ... reported at 1 GPU
(DP) and 1/2/4/8 GPUs (minimizer scoring)"; the DP is a chain of 1.4 x 10^5 dependent levels and does not shard
(replicas only).  A *step* is one scoring pass over the 1,007,415 x 150-bp 30x read set, sharded by read
over the ranks (dipgenie_amd/dist_sketch.py: local HIP sketch, RCCL all-reduce of the dictionary hit vector, hash-range
all-to-all of (hash, count) runs, sharded merge, one fused all-reduce of range sizes / histogram / ids).  `metric` =
sketch_reads_per_s, `value` = reads x steps / max-over-ranks time, "scaling": "strong" (the read set is fixed); rank 0
re-scores the whole set alone afterwards and every replicated output must be identical.  One DP instance per rank is
timed beside it as `dp_replicated` (informational).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # the stack's default; host-resident kernel arguments cost the sweep 1.9 us per level

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def truncated_graph(g, target_cells):
    """levels [0, P) of g as a standalone graph for the CPU baseline sample"""
    from dipgenie_amd.capi import DpGraphArrays
    k = np.diff(g.level_off).astype(np.int64)
    cum = np.cumsum(k[1:] ** 2 * (g.R + 1))
    P = int(min(max(np.searchsorted(cum, target_cells) + 2, 3), g.n_levels))
    nv = int(g.level_off[P])
    last0 = int(g.level_off[P - 1])
    out_off = g.out_off[: nv + 1].copy()
    out_off[last0:] = out_off[last0]                     # last kept level has no out-edges
    ne = int(out_off[-1])
    return DpGraphArrays(g.R, level_off=g.level_off[: P + 1], out_off=out_off, out_dst=g.out_dst[:ne], out_w=g.out_w[:ne],
                         hom_off=g.hom_off[: nv + 1], hom_col=g.hom_col[: int(g.hom_off[nv])],
                         het_off=g.het_off[: nv + 1], het_col=g.het_col[: int(g.het_off[nv])]), P


def usable_cores(cap=32):
    """CPU threads this process may really use: affinity mask, clipped by the cgroup CPU quota (a GPU box hands one GPU
    a 16-CPU share of a larger host; oversubscribing it makes OpenMP spin loops fight the HIP runtime's threads)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(cap, n))


def _run_reference(exe, cores, gfa, reads, out_fa):
    """runs the reference binary; returns (DP seconds from its own timer approximator.cpp:1006-1009, wall seconds, md5) or None"""
    import hashlib, re, subprocess
    t0 = time.perf_counter()
    try:
        p = subprocess.run([exe, f"-t{cores}", "-p2", "-R18", "-g", gfa, "-r", reads, "-o", out_fa], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900)
    except (OSError, subprocess.TimeoutExpired):
        return None
    wall = time.perf_counter() - t0
    m = re.search(r"diploid_dp_approximation_solver took (\d+) ms", p.stdout.decode(errors="replace"))
    if p.returncode != 0 or not m or not os.path.exists(out_fa):
        return None
    return max(int(m.group(1)), 1) / 1e3, wall, hashlib.md5(open(out_fa, "rb").read()).hexdigest()


def reference_baseline(cache, workload, bench_gfa, device, frac=0.05):
    """The reference's own OpenMP solver (oracle/_ref/DipGenie_ref, built by __graft_entry__.build() from /root/reference
    where that exists) timed on this node's host cores on a bounded sample of the bench workload: the first ~5 % (--ref-sample-frac; 10 % measured in profiles/r04_bench_mhc24.json) of the
    24-walk panel cut out as a panel of its own (dipgenie_amd.synth.prefix_panel; reads re-simulated with the same
    recipe).  Our CLI solves the same sample on the GPU: the two FASTA files must be identical, and its summary gives
    the cell count.  Falls back to the MHC_4 instance (BASELINE configs[1], golden md5) and then to None (oracle port)."""
    import hashlib, subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "DipGenie_ref")
    if not os.path.exists(exe):
        return None
    cores = usable_cores()
    os.makedirs(cache, exist_ok=True)
    if workload == "mhc24":
        try:
            from dipgenie_amd import synth
            pre_gfa, pre_fa = os.path.join(cache, "mhc24_prefix.gfa"), os.path.join(cache, "mhc24_prefix.fa")
            info = synth.prefix_panel(bench_gfa, pre_gfa, pre_fa, frac)
            r = _run_reference(exe, cores, pre_gfa, pre_fa, os.path.join(cache, "mhc24_prefix_ref.fa"))
            ours_fa, ours_js = os.path.join(cache, "mhc24_prefix_ours.fa"), os.path.join(cache, "mhc24_prefix_ours.json")
            subprocess.run([os.path.join(ROOT, "bin", "DipGenie"), "-t", str(cores), "-p2", "-R18", "-g", pre_gfa, "-r", pre_fa, "-o", ours_fa,
                            "-J", ours_js, "-G", str(device)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            ours = json.load(open(ours_js))
            if r and hashlib.md5(open(ours_fa, "rb").read()).hexdigest() == r[2]:
                dp_s, wall, md5 = r
                return {"value": ours["cells"] / dp_s, "unit": "cells/s", "cores": cores, "kind": "reference",
                        "threads_requested": 32, "threads_available": cores,
                        "sample": f"reference binary -t{cores} -p2 -R18 on the first {100 * frac:g} % of the bench panel cut out as its own panel (24 walks, "
                                  f"{info['hap_bp'][0]} bp, {ours['n_levels']} levels, {ours['cells']} cells, {info['n_reads']} reads): its DP function took "
                                  f"{dp_s:.2f} s of {wall:.1f} s end to end; FASTA identical to the GPU run on the same sample (md5 {md5[:8]}). "
                                  f"Threads: {cores} = this box's CPU share (north_star names a 32-thread baseline; no 32-core host is available to this run). "
                                  "The WHOLE bench workload takes the same binary 354.6 s of DP (105 M cells/s) and 483.9 s end to end at 8 threads "
                                  "in the build container (round-2 run, FASTA md5 cd13930a = the GPU run's; DESIGN.md section 6)"}
        except Exception as e:                              # noqa: BLE001 - any failure: next fallback
            log(f"prefix-panel reference baseline failed ({e!r}); falling back to MHC_4")
    gfa, reads = (os.path.join(ROOT, "tests", "data", n) for n in ("MHC_4.gfa.gz", "CHM13_reads.fq.gz"))
    if not (os.path.exists(gfa) and os.path.exists(reads)):
        return None
    r = _run_reference(exe, cores, gfa, reads, os.path.join(cache, "ref_mhc4_p2.fa"))
    if not r:
        return None
    dp_s, wall, md5 = r
    with open(os.path.join(ROOT, "tests", "golden", "e2e.json")) as f:
        if md5 != json.load(f)["mhc4_p2"]["fasta_md5"]:
            return None
    cells = 421330928
    return {"value": cells / dp_s, "unit": "cells/s", "cores": cores, "kind": "reference", "threads_requested": 32, "threads_available": cores,
            "sample": f"reference binary -t{cores} -p2 -R18 on MHC_4.gfa.gz + CHM13_reads.fq.gz (BASELINE configs[1] graph, 5 walks, "
                      f"{cells} cells): its DP function took {dp_s:.2f} s of {wall:.1f} s end to end; FASTA md5 matches the golden"}


def load_config4_shard(cache, device, world, rank):
    """BASELINE configs[3]: 30x reads (1,007,415 x 150 bp, seed 30) on the bench panel; this rank's contiguous block, resident"""
    import torch
    from dipgenie_amd import synth
    from dipgenie_amd.dist_sketch import shard_bounds
    path = synth.ensure_mhc24_reads(os.path.join(cache, "mhc24"))
    arr = np.load(path, mmap_mode="r")
    n, rl = arr.shape

    def resident(lo, hi):
        b = torch.from_numpy(np.array(arr[lo:hi]).reshape(-1)).to(device)
        o = (torch.arange(hi - lo + 1, dtype=torch.int64) * rl).to(device)
        return b, o
    lo, hi = shard_bounds(n, world, rank)
    return resident(lo, hi), resident, int(n), int(rl)


def same_score(a, b):
    import torch
    return bool(a.n_distinct == b.n_distinct and torch.equal(a.counts, b.counts) and torch.equal(a.ids, b.ids) and torch.equal(a.hist, b.hist))


def attach_profile(roof, launch_profile, workload):
    """HBM traffic (PMC) and the rocprofv3 kernel-trace average of the same kernels on the same workload come from the
    committed profile summary (rocprofv3 cannot run inside this process).  It is used only if it was taken on exactly
    the launches of THIS run: same kernel variants, same launch count per variant; otherwise the fields stay null."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_roofline.json")))
    roof["launch_profile"] = launch_profile
    if workload != "mhc24" or not found:
        return
    path = found[-1]                                        # the newest round's
    name = os.path.relpath(path, ROOT)
    with open(path) as f:
        prof = json.load(f)
    if prof.get("launch_profile") != launch_profile:
        roof["traffic_note"] = f"{name} was taken on a different set of sweep launches: stale, ignored (re-run tools/roofline_profile.sh + tools/roofline_from_profiles.py)"
        return
    roof["traffic"] = prof["hbm_bytes_per_launch"]["high"]
    roof["traffic_unit"] = "bytes/launch"
    roof["traffic_range"] = [prof["hbm_bytes_per_launch"]["low"], prof["hbm_bytes_per_launch"]["high"]]
    roof["traffic_source"] = (f"{name} (committed, not measured in this run; launch profile identical to this run's): WRITE_SIZE + "
                              "calibrated FETCH_SIZE of all sweep launches of one DP pass, separate --pmc passes, plain launches (sync_every)")
    roof["avg_launch_ms_rocprof"] = prof["kernel_trace"]["avg_launch_ns"] / 1e6
    roof["frac_rocprof"] = roof["algorithmic_bytes_per_launch"] / prof["kernel_trace"]["avg_launch_ns"] / roof["peak"]
    roof["frac_rocprof_note"] = ("an UNDER-TRACER figure, not a second estimate of the untraced rate: every traced dispatch carries its own completion signal and "
                                 "timestamps, so the traced durations sum to more than this run's whole step; `frac` (HIP events over the untraced launches) is the rate")
    # what the chip really moves per second during the sweep (counter bytes over THIS run's HIP-event launch time) against the HBM peak
    lo, hi = roof["traffic_range"]
    roof["hbm_frac_measured"] = [lo / (roof["avg_launch_ms"] * 1e-3) / 1e9 / roof["peak"], hi / (roof["avg_launch_ms"] * 1e-3) / 1e9 / roof["peak"]]


def native_runtime_check(dpg, device, warm, passes):
    """the DP part of the timed steps again, from plain C++ over the C ABI (bin/dg_dp_bench): a process that runs libdipgenie_hip.so on the
    HIP runtime it was built against (this process imported torch first and runs it on torch's bundled runtime)"""
    exe = os.path.join(ROOT, "bin", "dg_dp_bench")
    try:
        p = subprocess.run([exe, dpg, str(warm), str(passes), str(device)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        return json.loads(p.stdout.decode().strip().split("\n")[-1]) if p.returncode == 0 else {"error": p.stderr.decode()[-300:]}
    except (OSError, ValueError, subprocess.TimeoutExpired) as e:
        return {"error": repr(e)}


def config4_one_run_block(cache, gfa, device, cores):
    """BASELINE configs[3] as ONE job: python -m dipgenie_amd.run_sharded (one process per GPU under torch.distributed; here one rank with
    RCCL and the collective path forced) on the 30x read set written out as a FASTA, beside bin/DipGenie on the same file: what the
    Python / torch launcher costs on top of the C++ CLI.  Both FASTA files must be identical."""
    import hashlib
    from dipgenie_amd import synth
    arr = np.load(synth.ensure_mhc24_reads(os.path.join(cache, "mhc24")), mmap_mode="r")
    d = os.path.join(cache, "c4run")
    os.makedirs(d, exist_ok=True)
    reads30 = os.path.join(d, "reads30.fa")
    if not os.path.exists(reads30):
        n, rl = arr.shape
        block = np.empty((n, 3 + rl + 1), np.uint8)
        block[:, :3] = np.frombuffer(b">r\n", np.uint8); block[:, 3:3 + rl] = arr; block[:, -1] = ord("\n")
        with open(reads30 + ".tmp", "wb") as f:
            f.write(block.tobytes())
        os.replace(reads30 + ".tmp", reads30)
    out = {"reads": int(arr.shape[0]), "read_file": "the 30x read set (seed 30) as a FASTA"}
    t0 = time.time()
    p = subprocess.run([os.path.join(ROOT, "bin", "DipGenie"), "-t", str(cores), "-p2", "-R18", "-g", gfa, "-r", reads30, "-o", os.path.join(d, "cli.fa"), "-G", str(device)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    out["cli_wall_s"] = time.time() - t0
    if p.returncode != 0:
        return {"error": "CLI: " + p.stderr.decode()[-300:]}
    time.sleep(5.0)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.time()
    p = subprocess.run([sys.executable, "-m", "dipgenie_amd.run_sharded", "--gpus", "1", "--force-collectives", "--backend", "nccl", "--device", str(device), "-g", gfa, "-r", reads30,
                        "-o", os.path.join(d, "sharded.fa"), "-J", os.path.join(d, "sharded.json"), "-t", str(cores), "-R", "18", "-q"], env=env, cwd=ROOT,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    out["run_sharded_wall_s"] = time.time() - t0
    if p.returncode != 0:
        return {"error": "run_sharded: " + p.stderr.decode()[-300:]}
    time.sleep(5.0)
    t0 = time.time()                                                     # the same job as ONE C++ process: bin/DipGenie --gpus 1 (RCCL through librccl, one rank)
    p2 = subprocess.run([os.path.join(ROOT, "bin", "DipGenie"), "-t", str(cores), "-p2", "-R18", "--gpus", "1", "--shard-transport", "rccl", "--shard-devices", str(device),
                         "-g", gfa, "-r", reads30, "-o", os.path.join(d, "cli_sharded.fa"), "-J", os.path.join(d, "cli_sharded.json")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    out["cli_gpus1_rccl_wall_s"] = time.time() - t0
    if p2.returncode != 0:
        return {"error": "DipGenie --gpus 1: " + p2.stderr.decode()[-300:]}
    cs = json.load(open(os.path.join(d, "cli_sharded.json")))
    out["cli_gpus1_rccl"] = {k: cs[k] for k in ("dictionary", "dictionary_hits", "shard_sketch_ms", "shard_exchange_ms", "spectrum")}
    out["cli_gpus1_rccl"]["stages_s"] = {k: v for k, v in cs["stages"].items() if k.startswith("sharded")}
    sm = json.load(open(os.path.join(d, "sharded.json")))
    md5 = lambda f: hashlib.md5(open(f, "rb").read()).hexdigest()
    out.update(run_sharded_inside_s=sm["wall_s"], run_sharded_stages_s=sm["stages_s"], fasta_identical=md5(os.path.join(d, "cli.fa")) == md5(os.path.join(d, "sharded.fa")) == md5(os.path.join(d, "cli_sharded.fa")),
               note="run_sharded_wall_s - run_sharded_inside_s = interpreter + torch import + process-group start-up of the launcher; one rank, RCCL, collective path forced "
                    "(N > 1 under RCCL has never run: one GPU per lease)")
    return out


def config5_block(cache, device, cores):
    """BASELINE configs[4] at a tenth of its size (5 Mbp backbone x 100 walks, seed 22, -p2 -R32, 4x reads): the tier where the
    back-pointer lattice (1.4 TB) does not fit HBM, so the drop-in CLI runs checkpoint + recompute exactly as at full size.  One run
    of bin/DipGenie as a child process (about 11 s); the value pass (levels swept once, values only) gives cells/s, the second pass
    (every segment re-swept with back-pointers and walked) the recompute factor.  Counter bytes come from the newest committed
    profiles/rNN_c5_roofline.json (tools/c5_profile.sh + tools/c5_roofline.py), used only if it was taken on the same launches."""
    import glob
    from dipgenie_amd import synth
    d = os.path.join(cache, "c5_5m")
    os.makedirs(d, exist_ok=True)
    gfa, fa = os.path.join(d, "c5.gfa"), os.path.join(d, "c5.fa")
    if not (os.path.exists(gfa) and os.path.exists(fa)):
        segs, links, walks, reads = synth.linear_panel(22, backbone_bp=5_000_000, n_haps=100)
        synth.write_gfa(gfa + ".tmp", segs, links, walks); synth.write_fasta(fa + ".tmp", reads)
        os.replace(gfa + ".tmp", gfa); os.replace(fa + ".tmp", fa)
        del segs, links, walks, reads
    js = os.path.join(d, "o.json")
    t0 = time.time()
    p = subprocess.run([os.path.join(ROOT, "bin", "DipGenie"), "-t", str(cores), "-p2", "-R32", "-g", gfa, "-r", fa, "-o", os.path.join(d, "o.fa"), "-J", js, "-G", str(device)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    wall = time.time() - t0
    if p.returncode != 0:
        return {"error": p.stderr.decode()[-300:]}
    sm = json.load(open(js))
    cells, fwd_s, tb_s = sm["cells"], sm["dp_forward_ms"] / 1e3, sm["dp_traceback_ms"] / 1e3
    alg = 32.0 * cells + 16.0 * sm["dp_edge_pairs"] + 4.0 * sm["dp_colour_entries"]          # SURVEY.md s8d, one pass
    achieved = alg / fwd_s / 1e9
    blk = {"workload": "chr22-style panel, 5 Mbp backbone x 100 walks (seed 22), -p2 -R32, 4x reads: BASELINE configs[4] at a tenth of its size; lattice beyond HBM "
                       "(checkpoint + recompute, nothing forced)", "cells": cells, "levels": sm["n_levels"], "vertices": sm["n_vertices"], "dp_value": sm["dp_value"],
           "lattice_segments": sm["dp_segments"], "lattice_chunks": sm["dp_chunks"],
           "value_pass_cells_per_s": cells / fwd_s, "value_pass_s": fwd_s, "recompute_and_walk_s": tb_s, "recompute_factor": (fwd_s + tb_s) / fwd_s,
           "dp_cells_per_s_both_passes": cells / (fwd_s + tb_s), "end_to_end_s": wall, "stages_s": sm.get("stages"),
           "roofline": {"bound": "hbm", "kernel": "dp_sweep_fast_kernel / dp_sweep_coop_kernel (value pass)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_pass": alg, "launches_both_passes": sm["dp_forward_launches"],
                        "avg_launch_ms_value_pass": 1e3 * fwd_s / max(sm["n_levels"] - 1, 1), "traffic": None,
                        "note": "frac prices the s8d bytes (16-byte cells) against the value pass; the chip moves 4-byte values (+ 2-byte back-pointers in the second pass): see traffic"}}
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_c5_roofline.json")))
    if found:
        prof = json.load(open(found[-1]))
        name = os.path.relpath(found[-1], ROOT)
        if prof.get("cells") == cells and prof["kernel_trace"]["sweep_launches"] == sm["dp_forward_launches"]:
            lo, hi = prof["traffic_bytes_one_run"]
            r = blk["roofline"]
            r["traffic"] = hi
            r["traffic_unit"] = "bytes per CLI run (both passes, all sweep launches)"
            r["traffic_range"] = [lo, hi]
            r["traffic_source"] = (f"{name} (committed, not measured in this run; same cells and sweep launch count): WRITE_SIZE + FETCH_SIZE x 1.5-2.0, separate --pmc passes; "
                                   "these counters sit on the L2s' fabric side and include Infinity-Cache hits: an upper bound of HBM traffic")
            r["fabric_frac_measured"] = [lo / (fwd_s + tb_s) / 1e9 / HBM_PEAK_GBS, hi / (fwd_s + tb_s) / 1e9 / HBM_PEAK_GBS]
            r["avg_launch_us_under_tracer"] = prof["kernel_trace"]["avg_launch_us_under_tracer"]
        else:
            blk["roofline"]["traffic_note"] = f"{name} was taken on another set of launches: ignored"
    return blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mhc24", choices=["mhc24", "mhc4"])
    ap.add_argument("--cache", default=os.environ.get("DG_BENCH_CACHE", "/tmp/dg_bench_cache"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=float, default=6e8)
    ap.add_argument("--no-concurrent", action="store_true", help="skip the multi-instance-per-GPU measurement")
    ap.add_argument("--no-reference-baseline", action="store_true", help="skip the oracle/_ref run (about 30 s)")
    ap.add_argument("--e2e-runs", type=int, default=3, help="timed end-to-end CLI runs (the median is reported)")
    ap.add_argument("--e2e-gap-s", type=float, default=5.0, help="pause between CLI runs: the driver releases an exited run's HBM in the background")
    ap.add_argument("--no-config4", action="store_true", help="N = 1 only: skip the 30x read-set scoring measurement (BASELINE configs[3])")
    ap.add_argument("--no-config5", action="store_true", help="N = 1 only: skip the 5 Mbp x 100-walk run of the CLI (BASELINE configs[4] at a tenth of its size, about 25 s)")
    ap.add_argument("--ref-sample-frac", type=float, default=0.05, help="prefix of the bench panel the reference binary is timed on (cpu_baseline); 0.1 takes 74 s (161 M cells/s at 16 threads, profiles/r04_bench_mhc24.json)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (one-GPU box): DG_BENCH_DEVICE pins every rank to one card, DG_BENCH_BACKEND=gloo replaces RCCL
    if os.environ.get("DG_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["DG_BENCH_DEVICE"])
    backend = os.environ.get("DG_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(*vals):
        if world == 1:
            return vals
        t = torch.tensor(list(vals), dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(x) for x in t.tolist())

    # ---------------------------------------------------------------- build + inputs (untimed)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build(verbose=False)
    barrier()
    from dipgenie_amd import capi, synth
    from dipgenie_amd.dist_sketch import HipOps, ShardedSketch, shard_bounds

    os.makedirs(args.cache, exist_ok=True)
    if args.workload == "mhc24":
        name = "synthetic MHC-24 (5 real + 19 mosaic walks, seed 24), -p2 -R18, 4x 150-bp reads (seed 4)"
        if rank == 0:
            gfa, reads_path, info = synth.ensure_mhc24(os.path.join(args.cache, "mhc24"))
        barrier()
        gfa, reads_path, info = synth.ensure_mhc24(os.path.join(args.cache, "mhc24"))
    else:
        name = "MHC_4 (5 walks) + CHM13 0.5x reads, -p2 -R18"
        gfa = os.path.join(ROOT, "tests", "data", "MHC_4.gfa.gz")
        reads_path = os.path.join(ROOT, "tests", "data", "CHM13_reads.fq.gz")
    R, K, W = 18, 31, 25
    pre = os.path.join(args.cache, args.workload)
    cli = os.path.join(ROOT, "bin", "DipGenie")
    e2e = None
    if rank == 0:
        # one end-to-end run of the drop-in CLI (HIP sketch + HIP DP): produces the levelized DP graph
        # (.dpg) that the timed steps re-solve, and the end-to-end seconds with per-stage breakdown.
        base_cmd = [cli, "-t", str(usable_cores()), "-p2", f"-R{R}", "-g", gfa, "-r", reads_path, "-o", pre + ".fa", "-G", str(local_rank)]
        # Three timed runs, the median reported.  A run that has just exited leaves ~100 GB of HBM that the driver releases in the
        # background for 3-4 s, and the next process's first allocations wait for it (tools/back_to_back.sh): runs are spaced by
        # E2E_GAP_S so that each one sees the device a lone run sees.
        runs = []
        for rep in range(args.e2e_runs):
            if rep:
                time.sleep(args.e2e_gap_s)
            t0 = time.time()
            subprocess.run(base_cmd + ["-J", pre + ".json"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            wall = time.time() - t0                         # the timed run writes nothing but the FASTA (+ the summary)
            runs.append((wall, json.load(open(pre + ".json"))))
        walls = [w for w, _ in runs]
        wall_med, e2e = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
        e2e["wall_s"] = wall_med
        e2e["wall_runs_s"] = walls
        time.sleep(args.e2e_gap_s)
        subprocess.run(base_cmd + ["-D", pre], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)   # same run again, dumping the .dpg
        time.sleep(args.e2e_gap_s)
        log(f"end-to-end CLI runs: {', '.join(f'{w:.2f}' for w in walls)} s (median {wall_med:.2f}), DP value {e2e['dp_value']}")
    barrier()
    g = capi.DpGraphArrays.load(pre + ".dpg")

    ctx = capi.Context(local_rank)                         # DP instance of this rank
    ctx.dp_load_graph(g)                                   # graph resident in HBM from here on
    ctx_sk = capi.Context(local_rank)                      # scoring: its own context, on a torch stream shared with the collectives
    sk = ShardedSketch(HipOps(ctx_sk, device), device)

    # reads: this rank's contiguous shard, resident on the GPU; haplotype-minimizer dictionary D
    if reads_path.endswith(".gz"):
        import gzip
        lines = gzip.open(reads_path, "rb").read().split(b"\n")
        all_reads = lines[1::4]
    else:
        all_reads = open(reads_path, "rb").read().split(b"\n")[1::2]
    all_reads = [r for r in all_reads if r]
    lo, hi = shard_bounds(len(all_reads), world, rank)
    mine = all_reads[lo:hi]
    off = np.zeros(len(mine) + 1, np.int64)
    np.cumsum([len(r) for r in mine], out=off[1:])
    bases_t = torch.frombuffer(bytearray(b"".join(mine)) or bytearray(1), dtype=torch.uint8).to(device)
    off_t = torch.from_numpy(off).to(device)
    _, seqs, _, walks = synth.parse_gfa(gfa)
    dict_parts = []
    for (_, _, wv) in walks:
        h, _p = ctx.sketch_haplotype(b"".join(seqs[v] for v in wv), K, W)
        dict_parts.append(h)
    D = np.unique(np.concatenate(dict_parts))
    dict_t = torch.from_numpy(D.view(np.int64).copy()).to(device)
    del seqs, walks, dict_parts
    cfg4 = None
    if args.workload == "mhc24" and (world > 1 or not args.no_config4):
        if rank == 0:
            synth.ensure_mhc24_reads(os.path.join(args.cache, "mhc24"))
        barrier()
        cfg4 = load_config4_shard(args.cache, device, world, rank)

    def timed(step, n_warm, n_steps):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks"""
        for _ in range(n_warm):
            step()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)[0]

    last = {}
    if world == 1:
        # ------------------------------------------------------------ N = 1: the DP headline (BASELINE configs[2])
        acc = {"sk": 0.0, "dp": 0.0, "fwd": 0.0, "dl": 0.0, "tb": 0.0, "on": False}

        def step():
            t0 = time.perf_counter()
            last["score"] = sk.score(bases_t, off_t, dict_t, K, W)      # minimizer scoring of the 4x read set
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            last["out"] = ctx.dp_run()                                   # delta + sweep + traceback; synchronises
            t2 = time.perf_counter()
            if acc["on"]:
                tm = ctx.dp_timing()
                acc["sk"] += t1 - t0; acc["dp"] += t2 - t1
                acc["fwd"] += tm.forward_ms; acc["dl"] += tm.delta_ms; acc["tb"] += tm.traceback_ms
        for _ in range(args.warmup):
            step()
        acc["on"] = True
        elapsed = timed(step, 0, args.steps)
        out, score = last["out"], last["score"]
        sk4 = None
        if cfg4 is not None:                                             # BASELINE configs[3] at one rank: the N = 1 point of the scaling curve
            (b4, o4), _resident, n4, rl4 = cfg4
            el4 = timed(lambda: last.__setitem__("s4", sk.score(b4, o4, dict_t, K, W)), 1, 5)
            s4 = last["s4"]
            sk4 = {"workload": f"{n4} x {rl4}-bp reads (30x, seed 30) on the bench panel, 1 rank", "reads": n4, "passes": 5,
                   "reads_per_s": n4 * 5 / el4, "Gbp_per_s": n4 * rl4 * 5 / el4 / 1e9, "ms_per_pass": 1e3 * el4 / 5,
                   "local_sketch_ms": ctx_sk.sketch_timing().total_ms, "distinct_hashes": s4.n_distinct,
                   "dictionary_hits": int((s4.counts > 0).sum().item()), "collectives": "none (1 rank)"}
    else:
        # ------------------------------------------------------------ N > 1: read-sharded scoring (BASELINE configs[3])
        if cfg4 is None:
            raise SystemExit("--gpus N > 1 measures the sharded scoring of the mhc24 30x read set")
        (b4, o4), resident, n4, rl4 = cfg4

        def step():
            last["s4"] = sk.score(b4, o4, dict_t, K, W)
        elapsed = timed(step, args.warmup, args.steps)
        s4 = last["s4"]
        local_ms = max_over_ranks(ctx_sk.sketch_timing().total_ms)[0]
        # untimed cross-check: rank 0 scores the whole read set alone; every replicated output must be identical
        ok = 1.0
        if rank == 0:
            solo = ShardedSketch(sk.ops, device)
            solo.world, solo.rank = 1, 0
            fb, fo = resident(0, n4)
            ok = 1.0 if same_score(solo.score(fb, fo, dict_t, K, W), s4) else 0.0
            del fb, fo
        if max_over_ranks(1.0 - ok)[0] > 0:
            raise SystemExit("sharded config-4 scoring differs from the single-rank scoring")
        # informational: one DP instance per rank (the DP does not shard), 2 passes each
        dp_el = timed(lambda: last.__setitem__("out", ctx.dp_run()), 1, 2)
        out = last["out"]

    if rank == 0:
        if e2e is not None and out.value != e2e["dp_value"]:
            raise SystemExit(f"bench DP value {out.value} != CLI DP value {e2e['dp_value']}")
        steps = args.steps
        cells = int(out.cells)
        tm = ctx.dp_timing()
        config = {"workload": name, "R": R, "k": K, "w": W, "levels": g.n_levels, "vertices": g.n_vertices,
                  "cells_per_pass": cells, "relaxations_per_pass": int(out.relaxations), "reads": len(all_reads)}
        if world == 1:
            if e2e is not None and score.n_distinct != int(e2e["spectrum"]):     # scoring class == single-process sketch of the CLI
                raise SystemExit(f"spectrum size {score.n_distinct} != CLI spectrum size {e2e['spectrum']}")
            # algorithmic bytes of one DP pass, SURVEY.md s8d: 32 B/cell + 16 B/edge-pair + 4 B/colour entry read
            alg_bytes = 32.0 * cells + 16.0 * tm.edge_pairs + 4.0 * tm.colour_entries
            fwd_s = acc["fwd"] / 1e3 / steps
            n_launch = max(int(tm.n_forward_launches), 1)
            achieved = alg_bytes / fwd_s / 1e9
            config["parallelism"] = "1 GPU: minimizer scoring + DP on the same device"
            line = {
                "metric": "dp_state_cells_per_s", "value": cells * steps / elapsed, "unit": "cells/s",
                "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic", "config": config,
                "dp_ms": {"delta": acc["dl"] / steps, "forward": acc["fwd"] / steps, "traceback": acc["tb"] / steps, "wall": 1e3 * acc["dp"] / steps},
                "reads_per_s": len(all_reads) * steps / acc["sk"] if acc["sk"] > 0 else None,
                "sketch_ms_per_step": 1e3 * acc["sk"] / steps,
                "roofline": {"bound": "hbm", "kernel": "dp_sweep_fast_kernel (level sweep, forward)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                             "algorithmic_bytes_per_launch": alg_bytes / n_launch, "launches": n_launch, "avg_launch_ms": 1e3 * fwd_s / n_launch,
                             "dependency_chain_levels": g.n_levels},
            }
            attach_profile(line["roofline"], ctx.dp_launch_profile(), args.workload)
            if sk4 is not None:
                line["sketch_config4"] = sk4
        else:
            config = {"workload": f"{n4} x {rl4}-bp reads (30x, seed 30) on the synthetic MHC-24 panel (BASELINE configs[3]), k={K} w={W}, "
                                  f"dictionary of {int(dict_t.numel())} haplotype minimizers", "reads": n4, "read_length": rl4,
                      "parallelism": f"reads sharded over {world} ranks; {backend}: async all-reduce int32[{int(dict_t.numel())}] (hit vector) + one all-to-all of fixed-size "
                                     f"[{world}, 1 + {sk.cap}, 2] blocks of (hash, count) runs by hash range (run length in the payload; the calibrating warm-up step exchanged exact runs "
                                     "behind an all-gather of the send counts) + one fused all-reduce (range sizes, histogram, ids, overflow flag); no host read inside a step"}
            line = {
                "metric": "sketch_reads_per_s", "value": n4 * steps / elapsed, "unit": "reads/s",
                "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": config,
                "Gbp_per_s": n4 * rl4 * steps / elapsed / 1e9, "local_sketch_ms_max_rank": local_ms, "distinct_hashes": s4.n_distinct,
                "range_sizes": s4.range_sizes, "dictionary_hits": int((s4.counts > 0).sum().item()), "matches_single_rank": True,
                "n1_reference": "the N = 1 run reports the same measurement as sketch_config4.reads_per_s",
                "dp_replicated": {"value": world * cells * 2 / dp_el, "unit": "cells/s", "passes_each": 2,
                                  "note": "one independent DP instance per rank (the DP does not shard); informational, not `value`"},
            }
        if e2e is not None:
            line["end_to_end_s"] = e2e["wall_s"]
            line["end_to_end_runs_s"] = e2e.get("wall_runs_s")
            line["end_to_end_note"] = (f"median of {len(e2e.get('wall_runs_s') or [1])} runs of bin/DipGenie -t{usable_cores()} -p2 -R18 as a child process (exec to exit), "
                                       f"{args.e2e_gap_s:g} s apart; stages are the median run's")
            line["end_to_end_stages_s"] = e2e.get("stages")
            built, bound = capi.hip_versions()
            line["hip_runtime"] = {"library_built_against": built, "bound_in_this_process": bound,
                                   "note": "bench.py imports torch first, so the library runs on torch's bundled HIP runtime here; the CLI (end_to_end_s) runs on the system's"}
        if world == 1 and not args.no_concurrent:
            # Not the headline: several INDEPENDENT instances (samples) of the same workload in flight on this one GPU,
            # one dg_ctx + stream + host thread each.  A single instance is a chain of 140 k dependent level launches
            # that leaves most of the chip idle; this is what that idle capacity is worth to a cohort run.
            import threading
            free_b, _total_b = torch.cuda.mem_get_info()
            need = int(tm.bp_bytes + tm.delta_bytes + tm.state_bytes) + (6 << 30)
            n_inst = 1 + max(0, min(2, int(free_b // max(need, 1))))
            if n_inst > 1:
                extra = [capi.Context(local_rank) for _ in range(n_inst - 1)]
                for c2 in extra:
                    c2.dp_load_graph(g)
                    c2.dp_run()                               # untimed: maps this instance's lattice chunks (seconds right after the CLI child freed its HBM)
                every = [ctx] + extra
                res = [None] * n_inst

                def run_inst(q):
                    for _ in range(3):
                        res[q] = every[q].dp_run()
                ths = [threading.Thread(target=run_inst, args=(q,)) for q in range(n_inst)]
                t0 = time.perf_counter()
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                dt = time.perf_counter() - t0
                if any(r.key() != out.key() for r in res):
                    raise SystemExit("concurrent instances disagree with the single-instance result")
                line["concurrent_instances"] = {"instances": n_inst, "passes_each": 3, "value": n_inst * 3 * cells / dt, "unit": "cells/s",
                                                "vs_single_instance": (n_inst * 3 * cells / dt) / (cells * steps / acc["dp"]),
                                                "note": "independent samples on one GPU (one dg_ctx per sample); informational, not `value`"}
                for c2 in extra:
                    c2.close()
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_py as orc                           # the checker, timed as the CPU baseline ("port")
            gs, P = truncated_graph(g, args.cpu_sample_cells)
            t0 = time.perf_counter()
            ref = orc.dp_solve(gs)
            dt = time.perf_counter() - t0
            gout = ctx.dp_solve(gs)                           # same sample on the GPU: must agree
            if (gout.value, gout.s_het, gout.p1, gout.p2) != (ref["value"], ref["s_het"], ref["p1"], ref["p2"]):
                raise SystemExit("GPU and CPU baseline disagree on the sample")
            port = {"value": ref["cells"] / dt, "unit": "cells/s", "cores": 1, "kind": "port",
                    "sample": f"first {P} of {g.n_levels} levels of the same graph ({ref['cells']} cells, {dt:.1f} s, "
                              "oracle/oracle_dp.cpp single thread; result cross-checked against the GPU)"}
            refb = None if args.no_reference_baseline else reference_baseline(args.cache, args.workload, gfa, local_rank, args.ref_sample_frac)
            line["cpu_baseline"] = refb or port
            if refb:
                line["cpu_baseline_port"] = port
    ctx.close()
    ctx_sk.close()
    if rank == 0:
        if world == 1:
            # the remaining two blocks run as child processes on the system's HIP runtime, after this process has released its device memory
            del ctx, ctx_sk
            torch.cuda.empty_cache()
            time.sleep(args.e2e_gap_s)
            nat = native_runtime_check(pre + ".dpg", local_rank, args.warmup, args.steps)
            if "cells_per_s" in nat:
                nat["vs_this_process_dp_only"] = nat["cells_per_s"] / (cells * steps / acc["dp"])
                nat["note"] = ("bin/dg_dp_bench: the DP passes of the timed steps again from plain C++ over the C ABI, in a process bound to the HIP runtime the library "
                               "was built against; compare cells_per_s with cells x steps / dp wall of this process")
            if "hip_runtime" in line:
                line["hip_runtime"]["native_runtime_check"] = nat
            else:
                line["native_runtime_check"] = nat
            if args.workload == "mhc24" and not args.no_config4:
                time.sleep(args.e2e_gap_s)
                line["config4_one_run"] = config4_one_run_block(args.cache, gfa, local_rank, usable_cores())
            if not args.no_config5:
                time.sleep(args.e2e_gap_s)
                line["config5"] = config5_block(args.cache, local_rank, usable_cores())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
