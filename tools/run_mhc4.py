#!/usr/bin/env python3
"""Drop-in CLI on the reference's own test data (BASELINE configs[0] and [1]: MHC_4 + CHM13 reads, -p1 and -p2 -R18, and the
seeded HG002 2x read set) with stage times.  usage: python tools/run_mhc4.py [reps] [threads]"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
gfa = os.path.join(ROOT, "tests", "data", "MHC_4.gfa.gz")
chm = os.path.join(ROOT, "tests", "data", "CHM13_reads.fq.gz")
_, hg = synth.ensure_mhc4_hg002("/tmp/hg002")
env = dict(os.environ, DG_DEBUG="1")
for name, args, reads in (("p1 CHM13", ["-p1"], chm), ("p2 CHM13", ["-p2", "-R18"], chm), ("p2 HG002 2x", ["-p2", "-R18"], hg)):
    for rep in range(reps):
        t0 = time.time()
        p = subprocess.run([f"{ROOT}/bin/DipGenie", f"-t{thr}", *args, "-g", gfa, "-r", reads, "-o", "/tmp/o4.fa", "-J", "/tmp/o4.json"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print(f"{name} run {rep}: rc={p.returncode} wall {time.time()-t0:.2f}s", flush=True)
        if rep == reps - 1:
            for line in p.stderr.decode().split("\n"):
                if any(k in line for k in ("stage]", "dg::dp]", "dg::haploid", "Real time", "[E::")): print("  ", line)
        if p.returncode == 0:
            print("  md5", hashlib.md5(open("/tmp/o4.fa", "rb").read()).hexdigest(), flush=True)
