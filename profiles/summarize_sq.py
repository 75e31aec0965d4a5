#!/usr/bin/env python3
"""Per-kernel averages of an SQ counter pass (profiles/run_pmc.sh): python profiles/summarize_sq.py gpurun_out/pmc_<tag> [name filter]"""
import collections
import csv
import os
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
names = sorted({c for k in acc for c in acc[k]})
print("%-58s" % "kernel", " ".join("%14s" % n[-14:] for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k][names[0]])[1]):
    if flt in k:
        print("%-58s" % k[:58], " ".join("%14.4g" % (acc[k][n][1] / max(acc[k][n][0], 1)) for n in names))
