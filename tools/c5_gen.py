#!/usr/bin/env python3
"""Generate the chr22-style panel of BASELINE configs[4] at a given backbone length: python3 tools/c5_gen.py <backbone_bp> <out_dir> [n_haps]
Writes <out_dir>/c5.gfa and <out_dir>/c5.fa (seed 22, the generator of tests/test_gpu_config5.py and tools/run_c5.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import synth
bp = int(sys.argv[1]); d = sys.argv[2]; H = int(sys.argv[3]) if len(sys.argv) > 3 else 100
os.makedirs(d, exist_ok=True)
t0 = time.time()
segs, links, walks, reads = synth.linear_panel(22, backbone_bp=bp, n_haps=H)
synth.write_gfa(f"{d}/c5.gfa", segs, links, walks); synth.write_fasta(f"{d}/c5.fa", reads)
print(f"generated: {len(segs)} segments, {len(walks)} walks, {len(reads)} reads in {time.time() - t0:.1f} s", flush=True)
