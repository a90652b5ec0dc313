#!/usr/bin/env python3
"""Generate a chr22-style panel (BASELINE configs[4], scaled) and run the drop-in CLI on it.
usage: python tools/run_c5.py <backbone_bp> [n_haps] [R]"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import synth
bp = int(sys.argv[1]); H = int(sys.argv[2]) if len(sys.argv) > 2 else 100; R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
d = "/tmp/c5"; os.makedirs(d, exist_ok=True)
t0 = time.time(); segs, links, walks, reads = synth.linear_panel(22, backbone_bp=bp, n_haps=H)
synth.write_gfa(f"{d}/c5.gfa", segs, links, walks); synth.write_fasta(f"{d}/c5.fa", reads)
print(f"generated: {len(segs)} segments, {len(walks)} walks, {len(reads)} reads in {time.time()-t0:.1f}s", flush=True)
env = dict(os.environ, DG_DEBUG="1")
reps = int(os.environ.get("C5_REPS", "2"))
for rep in range(reps):
    t0 = time.time()
    p = subprocess.run([f"{ROOT}/bin/DipGenie", "-t16", "-p2", f"-R{R}", "-g", f"{d}/c5.gfa", "-r", f"{d}/c5.fa", "-o", f"{d}/o{rep}.fa", "-J", f"{d}/o.json"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(f"run {rep}: rc={p.returncode} wall {time.time()-t0:.2f}s")
    for line in p.stderr.decode().split("\n"):
        if any(k in line for k in ("stage]", "lattice", "dg::dp", "Real time", "[E::")): print("  ", line)
    if p.returncode == 0: print("  ", {k: v for k, v in json.load(open(f"{d}/o.json")).items() if k != "stages"})
if reps > 1: print("outputs identical:", open(f"{d}/o0.fa").read() == open(f"{d}/o1.fa").read())
import resource
print(f"peak RSS of the generator process: {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB, children {resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6:.1f} GB")
if len(sys.argv) > 4:      # also dump the levelized graph and keep a prefix window of it (kernel experiments)
    subprocess.run([f"{ROOT}/bin/DipGenie", "-t16", "-p2", f"-R{R}", "-g", f"{d}/c5.gfa", "-r", f"{d}/c5.fa", "-o", f"{d}/o2.fa", "-D", f"{d}/c5"],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import bench
    from dipgenie_amd import capi
    g = capi.DpGraphArrays.load(f"{d}/c5.dpg")
    gs, P = bench.truncated_graph(g, float(sys.argv[4]))
    os.makedirs(f"{ROOT}/gpurun_out", exist_ok=True)
    gs.save(f"{ROOT}/gpurun_out/win_c5.dpg")
    print(f"window: first {P} of {g.n_levels} levels, {gs.n_vertices} vertices -> gpurun_out/win_c5.dpg ({os.path.getsize(ROOT + '/gpurun_out/win_c5.dpg') / 1e6:.1f} MB)")
