#!/usr/bin/env python3
"""Two (or N) independent DP instances on ONE GPU, one dg_ctx + stream + host thread each: does one instance's
level latency hide behind the other's?  usage: python tools/dp_two.py graph.dpg [n_instances] [reps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import capi
g = capi.DpGraphArrays.load(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctxs = [capi.Context(0) for _ in range(N)]
for c in ctxs:
    for kv in os.environ.get("DG_OPTS", "").split(","):          # e.g. DG_OPTS="rowx=0,coop=0"
        if kv: c.dp_set_option(kv.split("=")[0], int(kv.split("=")[1]))
    c.dp_load_graph(g)
outs = [None] * N
def work(q):
    for _ in range(reps): outs[q] = ctxs[q].dp_run()
work(0)                                                     # warm-up, single
t0 = time.time(); work(0); t1 = time.time() - t0
print(f"1 instance : {reps} passes in {t1:.3f} s -> {reps * outs[0].cells / t1 / 1e9:.2f} G cells/s", flush=True)
for n in range(2, N + 1):
    th = [threading.Thread(target=work, args=(q,)) for q in range(n)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    tn = time.time() - t0
    assert all(outs[q].key() == outs[0].key() for q in range(n))
    print(f"{n} instances: {n * reps} passes in {tn:.3f} s -> {n * reps * outs[0].cells / tn / 1e9:.2f} G cells/s aggregate ({t1 * n / tn:.2f}x)", flush=True)
