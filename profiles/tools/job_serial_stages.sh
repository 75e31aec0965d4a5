#!/bin/bash
# serial (one stream) forward with hipEvent stage timing: prints ms per cloud and the stage table.  usage: bash profiles/tools/job_serial_stages.sh [reps]
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 ${1:-2}); do
python3 bench.py --no-pipeline --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
st={r['name']: r for r in j['stages']}
dense=sum(v['ms_per_step'] for k,v in st.items() if 'dense' in k); dec=sum(v['ms_per_step'] for k,v in st.items() if k.startswith('dec'))
print('serial %.4f ms/cloud | kdtree_build %.4f (%s launches) knn %.4f | dense %.4f decoder %.4f head %.4f' % (j['ms_per_step'], st['kdtree_build']['ms_per_step'], st['kdtree_build'].get('launches_per_step'), st['knn_search']['ms_per_step'], dense, dec, st.get('head',{}).get('ms_per_step',0)))
print('   ', {k: round(v['ms_per_step'],4) for k,v in st.items() if 'dense' in k or k.startswith('dec')})"
done
