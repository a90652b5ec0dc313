#!/usr/bin/env python3
"""Turns the output directory of tools/c5_profile.sh into profiles/rNN_c5_roofline.json -- everything bench.py's `config5` block quotes
from a profiler, recomputable from the committed CSVs:   python3 tools/c5_roofline.py <dir of tools/c5_profile.sh> <out.json>
  * kernel trace: sweep launches, traced total and average duration (an UNDER-TRACER figure: every traced dispatch carries its own
    completion signal, so it is an upper bound of the untraced pitch);
  * PMC: WRITE_SIZE and FETCH_SIZE (KiB) summed over all sweep launches of one CLI run (value pass + recompute pass), separate --pmc
    passes; FETCH_SIZE x 1.5-2.0 on gfx950 (profiles/r01_pmc_calib_*: 0.500 of wide coalesced reads, 0.667 of 84-byte row segments);
  * SQ: waves, wave cycles, VALU activity of the sweep kernels."""
import csv, json, sys
d, out = sys.argv[1], sys.argv[2]
plain = json.load(open(f"{d}/plain.json"))
cells, fwd_ms, tb_ms = plain["cells"], plain["dp_forward_ms"], plain["dp_traceback_ms"]
launches = plain.get("dp_forward_launches")
stats = {}
tot_ns = n_calls = 0
for r in csv.DictReader(open(f"{d}/c5_kernel_stats.csv")):
    if "dp_sweep" in r["Name"]:
        nm = r["Name"].split("(")[0].replace("void dgi::", "")
        stats[nm] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_us": float(r["AverageNs"]) / 1e3}
        tot_ns += float(r["TotalDurationNs"]); n_calls += int(r["Calls"])
def pmc(name):
    s = {}; n = 0
    for r in csv.DictReader(open(f"{d}/c5_pmc_{name}.csv")):
        if "dp_sweep" in r["kernel"]:
            s[r["counter"]] = s.get(r["counter"], 0.0) + float(r["sum"])
            if r["counter"] in (name, "SQ_WAVES"): n += int(r["dispatches"])
    return s, n
wr, n_wr = pmc("WRITE_SIZE"); fe, n_fe = pmc("FETCH_SIZE"); sq, n_sq = pmc("SQ")
visits = 2.0 * cells                                    # nominal: every level is swept twice beyond HBM (value pass + recompute pass; the second pass is plane-limited, below)
import os, re
swept = None
if os.path.exists(f"{d}/plain.err"):
    m = re.search(r"second pass swept ([0-9.]+) % of the \(level, plane\) pairs", open(f"{d}/plain.err").read())
    swept = float(m.group(1)) if m else None
w_b = wr["WRITE_SIZE"] * 1024.0; f_raw = fe["FETCH_SIZE"] * 1024.0
alg = 32.0 * cells + 16.0 * plain.get("dp_edge_pairs", plain["relaxations"] // 33) + 4.0 * plain.get("dp_colour_entries", 0)
res = {
    "workload": "chr22-style panel (BASELINE configs[4] at a tenth of its size): 5 Mbp backbone x 100 walks, seed 22, -p2 -R32, 4x reads; drop-in CLI",
    "cells": cells, "n_levels": plain["n_levels"], "dp_segments": plain["dp_segments"], "dp_chunks": plain["dp_chunks"],
    "untraced_run": {"forward_ms": fwd_ms, "recompute_and_walk_ms": tb_ms, "value_pass_cells_per_s": cells / (fwd_ms / 1e3),
                     "recompute_factor": (fwd_ms + tb_ms) / fwd_ms},
    "second_pass_planes_swept_pct_before_last_segment": swept,
    "kernel_trace": {"sweep_launches": n_calls, "total_ms": tot_ns / 1e6, "avg_launch_us_under_tracer": tot_ns / n_calls / 1e3, "per_variant": stats},
    "pmc": {"dispatches": {"WRITE_SIZE": n_wr, "FETCH_SIZE": n_fe, "SQ": n_sq},
            "WRITE_SIZE_bytes": w_b, "FETCH_SIZE_raw_bytes": f_raw, "FETCH_SIZE_calibrated_bytes": [1.5 * f_raw, 2.0 * f_raw],
            "bytes_per_cell_visit": {"write": w_b / visits, "fetch_raw": f_raw / visits, "fetch_calibrated": [1.5 * f_raw / visits, 2.0 * f_raw / visits]},
            "note": "FETCH_SIZE / WRITE_SIZE count the L2s' fabric-side requests: Infinity-Cache hits are included (MI355X_MICROARCH.md), so this is an upper bound of HBM traffic; "
                    "the two state buffers (2 x 10 MB on a 300-wide level) live in the Infinity Cache"},
    "traffic_bytes_one_run": [w_b + 1.5 * f_raw, w_b + 2.0 * f_raw],
    "algorithmic_bytes_one_pass_s8d": alg,
    "sq": sq,
}
if sq.get("SQ_WAVES"):
    res["sq_derived"] = {"waves_per_launch": sq["SQ_WAVES"] / max(n_sq, 1), "wave_lifetime_cycles": 4.0 * sq["SQ_WAVE_CYCLES"] / sq["SQ_WAVES"],
                         "valu_busy_frac": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024.0 * sq["GRBM_GUI_ACTIVE"] / 8.0),
                         "wave_cycles_waiting_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], "wave_cycles_issuing_frac": sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("untraced_run", "traffic_bytes_one_run", "algorithmic_bytes_one_pass_s8d")}, indent=1))
print(json.dumps(res["pmc"]["bytes_per_cell_visit"]), json.dumps(res.get("sq_derived")))
