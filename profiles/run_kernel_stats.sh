#!/bin/bash
# Kernel-time summary of one bench.py run on the GPU box (rocprofv3 --kernel-trace --stats, CSV output).
# usage (through gpurun): bash profiles/run_kernel_stats.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --no-cpu-baseline --no-stage-timing "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv; files:", glob.glob("gpurun_out/prof_$tag/**/*", recursive=True)[:10])
else:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:int("${TOPN:-16}")]:
        print("%-86s %6s %10.1f us" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
