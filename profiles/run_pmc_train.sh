#!/bin/bash
# rocprofv3 --pmc passes of the training step (counters only: no trace domains, as the pool requires; program directly after `--`).
# usage (through gpurun): bash profiles/run_pmc_train.sh <tag> "<COUNTER ...>" [bench args...]  -> gpurun_out/pmc_<tag>/pmc_counter_collection.csv
# PS_BENCH_NO_PREHEAT: the run is 1 (first) + W + K steps, nothing else (--no-stage-timing: no profiled extra steps)
tag=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PS_BENCH_NO_PREHEAT=1
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -o pmc -- python3 bench.py --mode train --no-cpu-baseline --no-stage-timing --steps 2 --warmup 1 "$@" > gpurun_out/pmc_$tag.log 2>&1
ls gpurun_out/pmc_$tag | head
