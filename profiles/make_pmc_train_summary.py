#!/usr/bin/env python3
"""HBM bytes of ONE training step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md
prescribes) of `bench.py --mode train ...` (profiles/run_pmc_train.sh).

  python profiles/make_pmc_train_summary.py <key> <fetch dir> <write dir> <commit> "<command>" [<key> <fetch dir> <write dir> ...]

A step = the dispatches between two consecutive Adam kernels (the LAST complete step of the run is taken).  Units / corrections
(MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced stream, so the read side is doubled; WRITE_SIZE is taken as is.  Writes profiles/<round>_pmc_traffic_train.json (what bench.py reads
for train_*.roofline.traffic) and profiles/<round>_pmc_train_per_kernel_<key>.json (per kernel: launches and bytes of that step)."""
import collections
import csv
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
ROUND = os.environ.get("PS_PROFILE_ROUND", "r5")  # file-name prefix of the round the passes belong to


def last_step(d):
    rows = list(csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    assert len(adam) >= 2, "fewer than two steps in %s" % d
    step = rows[adam[-2] + 1:adam[-1] + 1]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


args = sys.argv[1:]
out_path = os.path.join(here, "%s_pmc_traffic_train.json" % ROUND)
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
while len(args) >= 5:
    key, fdir, wdir, commit, command = args[:5]
    args = args[5:]
    f, w = last_step(fdir), last_step(wdir)
    rows = []
    for k in set(f) | set(w):
        fb, wb = 2 * 1024 * f.get(k, [0, 0.0])[1], 1024 * w.get(k, [0, 0.0])[1]
        rows.append(dict(kernel=k, launches=f.get(k, w.get(k))[0], fetch_bytes=fb, write_bytes=wb, bytes=fb + wb))
    rows.sort(key=lambda r: -r["bytes"])
    total = sum(r["bytes"] for r in rows)
    json.dump(rows, open(os.path.join(here, "%s_pmc_train_per_kernel_%s.json" % (ROUND, key)), "w"), indent=1)
    out[key] = dict(bytes_per_step=total, fetch_bytes_per_step=sum(r["fetch_bytes"] for r in rows), write_bytes_per_step=sum(r["write_bytes"] for r in rows),
                    launches_per_step=sum(r["launches"] for r in rows), command=command,
                    top_kernels=[dict(kernel=r["kernel"][:100], launches=r["launches"], gbytes=round(r["bytes"] / 1e9, 3)) for r in rows[:8]])
    out["_commit"] = commit
    print("%s: %.2f GB per step (fetch %.2f, write %.2f), %d launches" % (key, total / 1e9, out[key]["fetch_bytes_per_step"] / 1e9,
                                                                        out[key]["write_bytes_per_step"] / 1e9, out[key]["launches_per_step"]))
    for r in rows[:10]:
        print("   %-90s x%-3d %7.3f GB" % (r["kernel"][:90], r["launches"], r["bytes"] / 1e9))
json.dump(out, open(out_path, "w"), indent=1)
