python -m pytest tests/test_gpu_knn.py tests/test_gpu_network.py tests/test_gpu_prepare.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-sub-results --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st={s['name']:(s['ms_per_step'],s['launches_per_step']) for s in d['stages']}
print(round(d['ms_per_step'],4), round(d['serial_ms_per_cloud'],4), st['kdtree_build'], st['knn_search'], st.get('pyramid_slices'))"
done
