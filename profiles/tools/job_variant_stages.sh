#!/bin/bash
# serial stage timing of library flavours: bash profiles/tools/job_variant_stages.sh <variant> [<variant> ...]   ("default" = the shipped library)
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do for rep in 1 2; do
if [ "$v" = default ]; then cmd="python3 bench.py"; else cmd="python3 profiles/tools/with_variant.py $v bench.py"; fi
$cmd --no-pipeline --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
st={r['name']: r for r in j['stages']}
dense=sum(v['ms_per_step'] for k,v in st.items() if 'dense' in k); dec=sum(v['ms_per_step'] for k,v in st.items() if k.startswith('dec'))
print('%-10s serial %.4f ms/cloud | dense %.4f decoder %.4f |' % ('$v', j['ms_per_step'], dense, dec), {k: round(v['ms_per_step'],4) for k,v in sorted(st.items()) if 'dense' in k or k.startswith('dec')})"
done; done
