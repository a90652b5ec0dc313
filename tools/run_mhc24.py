#!/usr/bin/env python3
"""Run the drop-in CLI on the synthetic MHC-24 workload (BASELINE configs[2] shape) a few times and print the stage
times.  usage: python tools/run_mhc24.py [reps] [threads]"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = "/tmp/mhc24"
gfa, reads, info = synth.ensure_mhc24(d)
env = dict(os.environ, DG_DEBUG="1")
for rep in range(reps):
    t0 = time.time()
    p = subprocess.run([f"{ROOT}/bin/DipGenie", f"-t{thr}", "-p2", "-R18", "-g", gfa, "-r", reads, "-o", f"{d}/o{rep}.fa", "-J", f"{d}/o.json"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(f"run {rep}: rc={p.returncode} wall {time.time()-t0:.2f}s", flush=True)
    for line in p.stderr.decode().split("\n"):
        if any(k in line for k in ("stage]", "lattice", "dg::", "dipgenie_hip]", "Real time", "[E::")): print("  ", line)
    if p.returncode == 0:
        print("  md5", hashlib.md5(open(f"{d}/o{rep}.fa", "rb").read()).hexdigest(), {k: v for k, v in json.load(open(f"{d}/o.json")).items() if k != "stages"}, flush=True)
