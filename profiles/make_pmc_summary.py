#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
into per-kernel HBM bytes per launch and the stage-keyed file bench.py reads for `roofline.traffic`.

  python profiles/make_pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write [round tag, default r1] [commit] [command]

Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced stream, so the read side is doubled (an upper estimate for the narrow,
scattered reads of the tree walk); WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import json
import os
import sys

fetch_dir, write_dir = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r1"
commit = sys.argv[4] if len(sys.argv) > 4 else "?"
command = sys.argv[5] if len(sys.argv) > 5 else "python bench.py --no-pipeline"


def per_kernel(d):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))):
        k = r["Kernel_Name"]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


f, w = per_kernel(fetch_dir), per_kernel(write_dir)
rows = []
for k in sorted(f, key=lambda k: -(f[k] + w.get(k, 0))):
    rows.append(dict(kernel=k, fetch_bytes_per_launch=2 * 1024 * f[k], write_bytes_per_launch=1024 * w.get(k, 0.0),
                     raw_FETCH_SIZE_KiB=f[k], raw_WRITE_SIZE_KiB=w.get(k, 0.0)))
here = os.path.dirname(os.path.abspath(__file__))
json.dump(rows, open(os.path.join(here, "%s_pmc_per_kernel.json" % tag), "w"), indent=1)
stage_of = {"knn_search": "ps::knn_pair_kernel<16>"}
out = {}
for stage, frag in stage_of.items():
    for r in rows:
        if frag in r["kernel"]:
            out[stage] = r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]
if tag != "r1":  # bench.py labels roofline.traffic with where it came from
    out["_commit"] = commit
    out["_command"] = command
    out["_top_kernels"] = [dict(kernel=r["kernel"][:90], hbm_bytes_per_launch=r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]) for r in rows[:3]]
json.dump(out, open(os.path.join(here, "%s_pmc_traffic.json" % tag), "w"), indent=1)
for r in rows[:12]:
    print("%-70s fetch %8.1f MB  write %8.1f MB" % (r["kernel"][:70], r["fetch_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6))
