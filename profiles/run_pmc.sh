#!/bin/bash
# rocprofv3 --pmc passes of one serial bench.py run on the GPU box (counters only: no trace domains, as the pool requires).
# usage (through gpurun): bash profiles/run_pmc.sh <tag> "<COUNTER ...>" [bench args...]  -> gpurun_out/pmc_<tag>/pmc_counter_collection.csv
tag=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -o pmc -- python3 bench.py --no-pipeline --no-cpu-baseline --no-stage-timing "$@" > gpurun_out/pmc_$tag.log 2>&1
ls gpurun_out/pmc_$tag | head
