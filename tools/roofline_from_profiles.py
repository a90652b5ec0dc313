#!/usr/bin/env python3
"""Recomputes the roofline numbers of the bench line from rocprofv3 evidence alone and writes the summary bench.py reads:

    python3 tools/roofline_from_profiles.py <bench.json> <dir with the files of tools/roofline_profile.sh> profiles/r02_roofline.json

  sweep kernel (dp_sweep_fast_kernel<*>, HBM bound):
     frac_rocprof = algorithmic bytes per launch / average launch duration (kernel-trace stats) / 8 TB/s
     algorithmic bytes per launch = (32 B x cells + 16 B x edge pairs + 4 B x colour entries) / launches  (from the bench line)
     traffic      = (WRITE_SIZE + FETCH_SIZE x correction) / launches, correction 1.5-2.0 (gfx950 tallies 128-B requests at 64 B,
                    profiles/README.md), including the look-ahead kernel's table reads
  sketch kernel (sketch_tile_kernel, ALU bound): VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8)
The launch profile (kernel variant -> launches of one pass) is stored with it; bench.py uses the file only when its own
run launched exactly that profile."""
import csv, json, re, sys

bench = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
d = sys.argv[2].rstrip("/")
PEAK = 8000.0


def variant(name):
    m = re.search(r"dp_sweep_(fast|coop)_kernel<(\d+), (false|true), (false|true)>", name)
    if m:
        return f"dp_sweep_{m.group(1)}_kernel<{m.group(2)},{'general' if m.group(4) == 'true' else 'lean'}>"
    return "dp_sweep_kernel" if "dp_sweep_kernel<" in name else None


stats = list(csv.DictReader(open(f"{d}/dp_kernel_stats.csv")))
passes = 3
sweep = [(variant(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"])) for r in stats if variant(r["Name"])]
calls = sum(c for _, c, _ in sweep)
total_ns = sum(t for _, _, t in sweep)
prof = {}
for v, c, _ in sweep:
    assert c % passes == 0, (v, c)
    prof[v] = prof.get(v, 0) + c // passes
other = {r["Name"].split("(")[0].replace("void dgi::", "").replace("dgi::", ""): (int(r["Calls"]) // passes, float(r["TotalDurationNs"]) / passes / 1e6)
         for r in stats if any(k in r["Name"] for k in ("dp_trace", "dp_delta", "dp_warm", "dp_edge", "dp_l2_prefetch", "dp_pf_ctl"))}
launches = calls // passes
roof = bench["roofline"]
assert launches == roof["launches"], (launches, roof["launches"])
alg = roof["algorithmic_bytes_per_launch"]
avg_ns = total_ns / calls


def pmc(counter):
    out = {}
    for r in csv.DictReader(open(f"{d}/dp_pmc_{counter}.csv")):
        out[r["kernel"]] = (int(r["dispatches"]), float(r["sum"]))
    return out


fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
sw = lambda t: sum(v[1] for k, v in t.items() if "dp_sweep" in k)      # KiB
nl = sum(v[0] for k, v in write.items() if "dp_sweep" in k)
assert nl == launches, (nl, launches)
warm_f = sum(v[1] for k, v in fetch.items() if "dp_warm_tables" in k)
w_kib, f_kib = sw(write), sw(fetch)
cells = bench["config"]["cells_per_pass"]
lo = (w_kib + 1.5 * f_kib + 2.0 * warm_f) * 1024
hi = (w_kib + 2.0 * f_kib + 2.0 * warm_f) * 1024
out = {
    "workload": bench["config"]["workload"], "launch_profile": dict(sorted(prof.items())), "launches": launches, "cells": cells,
    "kernel_trace": {"passes": passes, "sweep_launches": calls, "sweep_total_ms": total_ns / 1e6, "avg_launch_ns": avg_ns,
                     "per_variant_avg_us": {v: round(t / c / 1e3, 3) for v, c, t in sorted(sweep)}, "other_kernels_per_pass": other,
                     "source": "dp_kernel_stats.csv: rocprofv3 --kernel-trace --stats -- python3 tools/dp_once.py mhc24.dpg 1 3"},
    "algorithmic_bytes_per_launch": alg, "frac_rocprof": alg / avg_ns / PEAK,
    "frac_hip_events_bench": roof["frac"], "avg_launch_ns_hip_events_bench": roof["avg_launch_ms"] * 1e6,
    "WRITE_SIZE_KiB": w_kib, "FETCH_SIZE_KiB_raw": f_kib, "lookahead_FETCH_SIZE_KiB_raw": warm_f,
    "write_bytes_per_cell": w_kib * 1024 / cells, "fetch_bytes_per_cell_raw": f_kib * 1024 / cells,
    "fetch_correction": {"low": 1.5, "high": 2.0, "why": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B: 2.00x for 2, 4 and 16 B/lane coalesced reads, 1.50x for "
                         "84-byte row segments (profiles/r01_pmc_calib_*.csv); WRITE_SIZE is exact for full-wave stores and counts whole 64-B requests for sparse segment heads"},
    "hbm_bytes_per_launch": {"low": lo / launches, "high": hi / launches}, "hbm_bytes_per_pass": {"low": lo, "high": hi},
    "hbm_frac_of_peak_measured_traffic": {"low": lo / launches / avg_ns / PEAK, "high": hi / launches / avg_ns / PEAK},
    "pmc_note": "FETCH_SIZE / WRITE_SIZE: separate --pmc passes (no trace domains) with DG_SYNC_EVERY=512: the stream is drained every 512 launches "
                "(the profiler dies with ~1e5 queued dispatches), launches are plain -- same kernels and launch counts as the bench's",
}
# ---- sketch (config-4 scoring pass on one rank)
try:
    sk = {r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void dgi::", "").replace("dgi::", ""): (int(r["Calls"]), float(r["AverageNs"]))
          for r in csv.DictReader(open(f"{d}/sketch_kernel_stats.csv"))}
    tile_name = "sketch_tile_kernel<3, false>" if any("sketch_tile_kernel<3, false>" in k for k in sk) else "sketch_tile_kernel<2, false>"
    tile = next(v for k, v in sk.items() if tile_name in k)
    sq = {}
    for r in csv.DictReader(open(f"{d}/sketch_pmc_sq.csv")):
        if tile_name in r["kernel"]:
            sq[r["counter"]] = float(r["sum"]) / int(r["dispatches"])
    valu_busy = sq["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * sq["GRBM_GUI_ACTIVE"] / 8)
    out["sketch"] = {"kernel": tile_name, "avg_ns": tile[1], "calls": tile[0], "bound": "valu",
                     "counters_per_dispatch": sq, "valu_busy_frac": valu_busy,
                     "valu_wave_insts_per_read": sq["SQ_INSTS_VALU"] / bench["sketch_config4"]["reads"] if "sketch_config4" in bench else None,   # one wave per 150-bp read (one tile)
                     "wave_cycles_split": {k: sq[k] / sq["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if k in sq},
                     "algorithmic_GBps": (1.0 + 8 * 2 / 26) * bench["sketch_config4"]["reads"] * 150 / tile[1] if "sketch_config4" in bench else None,
                     "source": "sketch_kernel_stats.csv / sketch_pmc_sq.csv: rocprofv3 over tools/score_profile.py (1,007,415 x 150-bp reads, one rank)"}
    # the spectrum stage behind the tile kernel (round 2: rocPRIM sort + reduce; round 3: one LDS table per hash-range bucket)
    tab = next((v for k, v in sk.items() if "bk_table_kernel" in k), None)
    if tab and "sketch_config4" in bench:
        pairs = sq.get("pairs") or 8.0e6                          # emitted (hash, read) pairs of the set (dg_sketch_timing.n_emitted: 7,996,5xx)
        nd = bench["sketch_config4"]["distinct_hashes"]
        alg = 20.0 * pairs + 12.0 * nd                            # pass 1 reads the hash, pass 2 hash + read; 12 B per distinct hash out
        rest = {k: v for k, v in sk.items() if k.strip() in ("bk_dscan_kernel", "bk_gather_kernel")}
        out["sketch_spectrum"] = {"kernel": "bk_table_kernel", "avg_ns": tab[1], "calls": tab[0], "bound": "hbm (nominal); LDS atomic latency in fact",
                                  "algorithmic_bytes": alg, "achieved_GBps": alg / tab[1], "peak_GBps": PEAK, "frac": alg / tab[1] / PEAK,
                                  "other_kernels_avg_ns": {k.strip(): v[1] for k, v in rest.items()},
                                  "note": "one 1,024-lane workgroup per bucket (4,096 buckets of ~2 k pairs), 114 KB of LDS tables each: one workgroup per CU; "
                                          "the second pass over the bucket hits L2; 5 % of the scoring pass"}
except (OSError, KeyError, StopIteration, ZeroDivisionError) as e:
    out["sketch"] = {"error": repr(e)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("launches", "frac_rocprof", "frac_hip_events_bench", "hbm_bytes_per_launch", "write_bytes_per_cell")}, indent=1))
print(json.dumps(out["sketch"], indent=1)[:1500])
