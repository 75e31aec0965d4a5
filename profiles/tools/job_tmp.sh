python3 bench.py --no-cpu-baseline --no-sub-results --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st={s['name']:(s['ms_per_step'],s['launches_per_step']) for s in d['stages']}
print(round(d['ms_per_step'],4), round(d['serial_ms_per_cloud'],4), st['kdtree_build'], st['knn_search'])"
TOPN=40 bash profiles/run_kernel_stats.sh kmid --no-pipeline --steps 20 --warmup 3 --no-sub-results 2>&1 | grep -i "build_\|huge\|fatten\|init_points"
