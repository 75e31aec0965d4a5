#!/bin/bash
# usage (through gpurun): bash profiles/tools/job_gemm32b.sh <tag> [env assignments...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/g32_$tag -o t -- python3 profiles/tools/exp_gemm32b.py 7 > gpurun_out/g32_$tag.log 2>&1
grep "SHAPE\|rror" gpurun_out/g32_$tag.log
python3 profiles/tools/trace_by_dispatch.py gpurun_out/g32_$tag
