#!/bin/bash
# same-box A/B of library flavours on the training steps: bash profiles/tools/job_train_ab.sh <variant|default> ...
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in "$@"; do
if [ "$v" = default ]; then cmd="python3 bench.py"; else cmd="python3 profiles/tools/with_variant.py $v bench.py"; fi
a=$($cmd --mode train --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
b=$($cmd --mode train --batch 1 --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
echo "$v: b8 fp32 $a ms  b1 $b ms"
done; done
