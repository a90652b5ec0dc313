#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter_collection CSV per kernel name: python tools/pmc_sum.py <dir> <out.csv>"""
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        tot[key][0] += 1
        tot[key][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as o:
    o.write("kernel,counter,dispatches,sum\n")
    for (k, c), (n, v) in sorted(tot.items()):
        o.write(f'"{k}",{c},{n},{v}\n')
print(open(sys.argv[2]).read())
