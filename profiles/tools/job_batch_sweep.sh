cd "$GRAFT_REPO_ROOT"
for b in 1 2 4; do for l in 2 4; do
python3 bench.py --batch $b --lanes $l --steps 120 --warmup 20 --no-cpu-baseline --no-sub-results 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('batch $b lanes $l: %.4f ms/step  %.4f ms/cloud  %.1f M pts/s' % (j['ms_per_step'], j['ms_per_step']/$b, j['value']/1e6))"
done; done
